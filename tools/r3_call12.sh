#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c12; mkdir -p $O
cd $ROOT
t() { for i in 1 2 3; do echo -n "[$1 | $2] "; env $1 timeout 120 python tools/dbg_d6.py $2 2>&1 | grep -v amdgpu | tail -1; done; }
t "A=1" ""
t "GPU_MAX_HW_QUEUES=4" ""
t "A=1" "rows_fused_reduce=0"
t "A=1" "lz_async=0"
t "CTM_MAX_CONCURRENT_UNITS=1" ""
t "CTM_MAX_CONCURRENT_UNITS=2" ""
t "A=1" "rows_fused_reduce=0 lz_async=0"
