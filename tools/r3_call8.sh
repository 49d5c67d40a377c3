#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c8; mkdir -p $O
cd $ROOT
echo "=== units=4 GPU_MAX_HW_QUEUES=8" >> $O/ab.txt; GPU_MAX_HW_QUEUES=8 CTM_LARGE_N_UNITS=4 timeout 300 python tools/probe_sweep_conv.py 8 256 3 >> $O/ab.txt 2>&1
echo "=== units=2 GPU_MAX_HW_QUEUES=8" >> $O/ab.txt; GPU_MAX_HW_QUEUES=8 CTM_LARGE_N_UNITS=2 timeout 300 python tools/probe_sweep_conv.py 8 256 3 >> $O/ab.txt 2>&1
grep -v amdgpu $O/ab.txt
mkdir -p /tmp/kt; cd /tmp
CTM_LARGE_N_UNITS=4 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $ROOT/tools/probe_sweep_conv.py 8 256 3 > $O/trace_run.txt 2> $O/kt.log
cd $ROOT
python tools/trace_phases.py /tmp/kt 0.6 > $O/phases.txt 2>&1
python tools/trace_timeline.py /tmp/kt 0.70 300 > $O/timeline.txt 2>&1
cat $O/phases.txt; grep -v amdgpu $O/trace_run.txt
awk '{print $3, $4}' $O/timeline.txt | sort | uniq -c
