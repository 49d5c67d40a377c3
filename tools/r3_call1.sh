#!/bin/bash
# round 3, GPU call 1: scheduling probe, sweep-by-sweep record, kernel timeline of two signed sweeps
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c1; mkdir -p $O
cd $ROOT
timeout 300 tools/bin/probe_cumask > $O/cumask.txt 2>&1
timeout 600 python tools/probe_sweep_conv.py 8 256 14 > $O/sweep_conv.txt 2>&1
mkdir -p /tmp/kt; cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $ROOT/bench.py --signed --steps 2 --warmup 4 --no-cpu-baseline --no-serial-pass > $O/bench_trace.json 2> $O/kt.log
cd $ROOT
python tools/trace_phases.py /tmp/kt 0.66 > $O/phases.txt 2>&1
python tools/trace_gaps.py /tmp/kt 0.66 > $O/gaps.txt 2>&1
python tools/trace_timeline.py /tmp/kt 0.80 150 > $O/timeline.txt 2>&1
tail -3 $O/cumask.txt; cat $O/sweep_conv.txt; cat $O/phases.txt
