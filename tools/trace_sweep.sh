#!/bin/bash
# kernel timeline of single sweeps of a bench run (on the GPU box):  tools/trace_sweep.sh TAG <bench.py args>
# writes gpurun_out/trace_TAG/{moving.txt (last sweep that ran the Cholesky-QR iteration), later.txt (the third and fourth delimiter windows after it: a sweep launches the delimiter kernel twice)}
export TMPDIR=/tmp
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/trace_$TAG; mkdir -p $O; rm -rf /tmp/kt_$TAG
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$TAG -- python $ROOT/bench.py "$@" > $O/bench.json 2> $O/kt.log
cd $ROOT
python tools/trace_window.py /tmp/kt_$TAG layer2_reg_kernel 0 chol64_scaled > $O/moving.txt 2>/dev/null
python tools/trace_window.py /tmp/kt_$TAG layer2_reg_kernel 0 chol64_scaled 3 > $O/later.txt 2>/dev/null
python tools/trace_window.py /tmp/kt_$TAG layer2_reg_kernel 0 chol64_scaled 4 >> $O/later.txt 2>/dev/null
