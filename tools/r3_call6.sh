#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c6; mkdir -p $O
cd $ROOT
run() { echo "=== stagger=$1 units=$2 $3" >> $O/ab.txt; CTM_STAGGER_MS=$1 CTM_LARGE_N_UNITS=$2 timeout 300 python tools/probe_sweep_conv.py 8 256 4 $3 >> $O/ab.txt 2>&1; }
run 60 2 "rows_target_wgs=512 lz_jacobi_block=16"
run 80 2 "rows_target_wgs=512 lz_jacobi_block=16"
run 100 2 "rows_target_wgs=512 lz_jacobi_block=16"
run 80 2 "rows_target_wgs=512"
run 80 2 ""
run 40 4 "rows_target_wgs=512 lz_jacobi_block=16"
run 80 2 "rows_target_wgs=256 lz_jacobi_block=16"
grep -v amdgpu $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "D8chi256-signed" > $O/tests2.txt 2>&1; echo "tests2 rc=$?"; tail -3 $O/tests2.txt
