mkdir -p gpurun_out
VERBOSE=1 timeout 600 python tools/probe_stationary.py 4 64 1e-9 24 > gpurun_out/probe_stat_D4.log 2>&1; grep -v "^\[" gpurun_out/probe_stat_D4.log | tail -14 | cut -c1-200; grep "^\[stat\]" gpurun_out/probe_stat_D4.log | tail -6
VERBOSE=1 timeout 600 python tools/probe_stationary.py 6 128 1e-9 22 > gpurun_out/probe_stat_D6.log 2>&1; grep -v "^\[" gpurun_out/probe_stat_D6.log | tail -12 | cut -c1-200; grep "^\[stat\]" gpurun_out/probe_stat_D6.log | tail -6
timeout 1500 python -m pytest tests/test_gpu_generic.py tests/test_gpu_threads.py tests/test_gpu_stationary.py tests/test_gpu_bench_ranks.py -m gpu -q -s > gpurun_out/s5_tests.log 2>&1; echo "tests rc=$?"; grep -v "^\[Gloo\]\|amdgpu.ids" gpurun_out/s5_tests.log | tail -40 | cut -c1-400
