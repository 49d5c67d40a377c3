#!/bin/bash
# rocprofv3 evidence for the default bench command (run on the GPU box through gpurun); outputs under gpurun_out/prof/
export TMPDIR=/tmp
O=gpurun_out/prof; rm -rf $O; mkdir -p $O/kt $O/fetch $O/write
ARGS="${@:---no-cpu-baseline}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python bench.py $ARGS > $O/bench_under_rocprof.json 2> $O/kt.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python bench.py $ARGS > /dev/null 2> $O/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python bench.py $ARGS > /dev/null 2> $O/write.log
python tools/pmc_summary.py $O/fetch $O/write > $O/pmc_hbm_traffic.csv
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/kt -name "*domain_stats.csv" -exec cp {} $O/domain_stats.csv \;
rm -rf $O/kt/*/ $O/fetch $O/write 2>/dev/null; find $O/kt -name "*kernel_trace.csv" -delete
ls -la $O; head -12 $O/kernel_stats.csv | cut -c1-160; head -8 $O/pmc_hbm_traffic.csv | cut -c1-200; tail -1 $O/bench_under_rocprof.json | cut -c1-400
