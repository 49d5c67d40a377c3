#!/bin/bash
# rocprofv3 evidence for a bench command (run on the GPU box through gpurun).
#   tools/profile_bench.sh TAG [--pmc] [--mfma] -- <bench.py args>
# writes gpurun_out/prof_TAG/{kernel_stats.csv, domain_stats.csv, bench_under_rocprof.json[, pmc_hbm_traffic.csv]}:
# one --kernel-trace --stats pass and, with --pmc, two separate counter passes (FETCH_SIZE, WRITE_SIZE; counters are never combined
# with tracing), aggregated by tools/pmc_summary.py with the gfx950 2x fetch correction (MI355X_MICROARCH.md, HBM section).
export TMPDIR=/tmp
TAG=$1; shift
PMC=0; if [ "$1" == "--pmc" ]; then PMC=1; shift; fi
MFMA=0; if [ "$1" == "--mfma" ]; then MFMA=1; shift; fi
if [ "$1" == "--" ]; then shift; fi
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O/kt $O/fetch $O/write
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $ROOT/bench.py "$@" > $O/bench_under_rocprof.json 2> $O/kt.log
if [ $PMC == 1 ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $ROOT/bench.py "$@" > /dev/null 2> $O/fetch.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $ROOT/bench.py "$@" > /dev/null 2> $O/write.log
  python $ROOT/tools/pmc_summary.py $O/fetch $O/write > $O/pmc_hbm_traffic.csv
fi
if [ $MFMA == 1 ]; then
  mkdir -p $O/mfma
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python $ROOT/bench.py "$@" > /dev/null 2> $O/mfma.log
  python $ROOT/tools/pmc_mfma_summary.py $O/mfma > $O/pmc_mfma.csv
  rm -rf $O/mfma
fi
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/kt -name "*domain_stats.csv" -exec cp {} $O/domain_stats.csv \;
rm -rf $O/kt $O/fetch $O/write 2>/dev/null
cd $ROOT
ls -la $O; head -8 $O/kernel_stats.csv | cut -c1-170; [ $PMC == 1 ] && head -6 $O/pmc_hbm_traffic.csv | cut -c1-200; tail -1 $O/bench_under_rocprof.json | cut -c1-300
