#!/bin/bash
# Round-end evidence batch (run on the GPU box through gpurun): rocprofv3 passes of the bench commands and the un-profiled bench lines.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_bench.sh default --pmc -- --no-full-rank --no-cpu-baseline --no-serial-pass > gpurun_out/prof_default.log 2>&1
bash tools/profile_bench.sh default_serial -- --no-full-rank --no-cpu-baseline --no-serial-pass --serial-units > gpurun_out/prof_default_serial.log 2>&1
bash tools/profile_bench.sh signed --pmc -- --signed --no-cpu-baseline --no-serial-pass > gpurun_out/prof_signed.log 2>&1
bash tools/profile_bench.sh signed_serial -- --signed --no-cpu-baseline --no-serial-pass --serial-units > gpurun_out/prof_signed_serial.log 2>&1
bash tools/profile_bench.sh c4v -- --config c4v_D4_chi64 --no-cpu-baseline > gpurun_out/prof_c4v.log 2>&1
bash tools/profile_bench.sh literal -- --no-cpu-baseline > gpurun_out/prof_literal.log 2>&1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python bench.py --config c4v_D4_chi64 > gpurun_out/bench_c4v_final.json 2>/dev/null
python bench.py --config c4v_D4_chi64_c128 > gpurun_out/bench_c4v_c128.json 2>/dev/null
python bench.py --config generic_D6_chi128 > gpurun_out/bench_d6_final.json 2>/dev/null
tail -c 400 gpurun_out/bench_final.json
