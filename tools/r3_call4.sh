#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c4; mkdir -p $O
cd $ROOT
timeout 120 tools/bin/probe_cumask nomask > $O/cumask.txt 2>&1; echo "probe rc=$?"
cat $O/cumask.txt
timeout 300 python tools/probe_sweep_conv.py 8 256 3 > $O/async.txt 2>&1; cat $O/async.txt | grep -v amdgpu
mkdir -p /tmp/kt; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $ROOT/tools/probe_sweep_conv.py 8 256 3 > $O/async_trace.txt 2> $O/kt.log
cd $ROOT
python tools/trace_phases.py /tmp/kt 0.6 > $O/phases.txt 2>&1
python tools/trace_timeline.py /tmp/kt 0.80 40 > $O/timeline.txt 2>&1
cat $O/phases.txt
