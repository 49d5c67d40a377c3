#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c5; mkdir -p $O
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_iterative.py tests/test_gpu_primitives.py -x -q > $O/tests1.txt 2>&1; echo "tests1 rc=$?"; tail -2 $O/tests1.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "D8chi256-signed or sweep_invariances" > $O/tests2.txt 2>&1; echo "tests2 rc=$?"; tail -2 $O/tests2.txt
run() { echo "=== units=$1 $2" >> $O/ab.txt; CTM_LARGE_N_UNITS=$1 timeout 300 python tools/probe_sweep_conv.py 8 256 4 $2 >> $O/ab.txt 2>&1; }
run 2 ""
run 2 "rows_target_wgs=512"
run 2 "rows_target_wgs=512 lz_jacobi_block=16"
run 4 "rows_target_wgs=512 lz_jacobi_block=16"
run 4 ""
run 3 "rows_target_wgs=512 lz_jacobi_block=16"
run 2 "lz_jacobi_block=16"
grep -v amdgpu $O/ab.txt
