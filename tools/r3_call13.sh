#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c13; mkdir -p $O
cd $ROOT
t() { for i in 1 2 3 4 5 6; do echo -n "[$1 | $2] "; env $1 timeout 120 python tools/dbg_d6.py $2 2>&1 | grep -v amdgpu | tail -1; done; }
t "A=1" ""
run() { echo "=== $1 | $2" >> $O/ab.txt; CTM_ENGINE_OPTS="$1" timeout 300 python tools/probe_sweep_conv.py 8 256 4 $2 >> $O/ab.txt 2>&1; }
run "" ""
run "" "heavy_serial=0"
run "" "rows_target_wgs=512"
grep -v amdgpu $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_dist.py -x -q > $O/dist.txt 2>&1; echo "dist rc=$?"; tail -3 $O/dist.txt
