#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c19; mkdir -p $O
cd $ROOT
run() { echo "=== $1" >> $O/ab.txt; timeout 300 python tools/probe_sweep_conv.py 8 256 4 $1 >> $O/ab.txt 2>&1; }
run "rows_target_wgs=512"
run "rows_target_wgs=256"
run "rows_target_wgs=640"
run ""
grep -v amdgpu $O/ab.txt | grep -E "===|sweep  3|sweep  4"
