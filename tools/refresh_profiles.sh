#!/bin/bash
# Round-end evidence batch (run on the GPU box through gpurun): rocprofv3 passes of the bench commands and the un-profiled bench lines.
# Copy what should be judged from gpurun_out/prof_* into profiles/ (see profiles/README.md).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
X="--no-cpu-baseline --no-serial-pass --no-other-configs --no-energy --no-live-traffic --no-stationary"
bash tools/profile_bench.sh default --pmc -- --no-full-rank $X > gpurun_out/prof_default.log 2>&1
bash tools/profile_bench.sh default_serial -- --no-full-rank $X --serial-units > gpurun_out/prof_default_serial.log 2>&1
bash tools/profile_bench.sh signed --pmc -- --signed $X > gpurun_out/prof_signed.log 2>&1
bash tools/profile_bench.sh signed_serial --mfma -- --signed $X --serial-units > gpurun_out/prof_signed_serial.log 2>&1
bash tools/profile_bench.sh stationary -- --signed $X --warm-tol 1e-9 --warmup 16 --steps 3 > gpurun_out/prof_stationary.log 2>&1
bash tools/profile_bench.sh c4v -- --config c4v_D4_chi64 --no-cpu-baseline --no-live-traffic > gpurun_out/prof_c4v.log 2>&1
bash tools/profile_bench.sh c128_signed -- --config generic_D8_chi384_c128 --signed --steps 1 --warmup 2 --no-cpu-baseline --no-serial-pass --no-energy --no-live-traffic > gpurun_out/prof_c128_signed.log 2>&1
SECONDS=0; python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "default bench.py: ${SECONDS} s of wall time" > gpurun_out/bench_final_wall.txt
cp gpurun_out/bench_detail.json gpurun_out/bench_final_detail.json
python bench.py --steps 20 --warmup 5 --no-other-configs --no-energy --no-cpu-baseline > gpurun_out/bench_20steps.json 2>/dev/null
python bench.py --config c4v_D4_chi64 > gpurun_out/bench_c4v_final.json 2>/dev/null
tail -c 600 gpurun_out/bench_final.json
# latency-bound C4v path: kernel timeline of one moving sweep, and the single-workgroup kernels alone (tools/build_bench_small.sh first)
bash tools/trace_sweep.sh c4v --config c4v_D4_chi64 --no-energy --no-other-configs --no-cpu-baseline --steps 12 --warmup 3 > /dev/null 2>&1
[ -x tools/bin/bench_small_kernels ] && timeout 120 tools/bin/bench_small_kernels > gpurun_out/bench_small_kernels.txt 2>&1
