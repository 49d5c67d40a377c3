#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c16; mkdir -p $O
cd $ROOT
run() { echo "=== $1" >> $O/ab.txt; timeout 300 python tools/probe_sweep_conv.py 8 256 4 $1 >> $O/ab.txt 2>&1; }
run ""
run "jacobi_gram_kmin_short=128"
run "jacobi_gram_kmin_short=224"
run "jacobi_gram_kmin_short=448"
run "eig64_bpt=1"
run "eig64_bpt=4"
run "rows_target_wgs=1024"
grep -v amdgpu $O/ab.txt | grep -E "===|sweep  3|sweep  4"
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "chunked or one_part or whole_move" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
