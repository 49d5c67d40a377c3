#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c18; mkdir -p $O
mkdir -p /tmp/kt; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $ROOT/tools/probe_sweep_conv.py 8 256 3 > $O/trace_run.txt 2> $O/kt.log
cd $ROOT
python tools/trace_phases.py /tmp/kt 0.62 > $O/phases.txt 2>&1
python tools/trace_timeline.py /tmp/kt 0.70 500 > $O/timeline.txt 2>&1
cat $O/phases.txt; grep -v amdgpu $O/trace_run.txt
awk '{print $3, $4}' $O/timeline.txt | sort | uniq -c
