mkdir -p gpurun_out/guard
timeout 600 python tools/probe_stationary.py 4 64 1e-9 28 > gpurun_out/probe_stat_D4.log 2>&1; tail -30 gpurun_out/probe_stat_D4.log
VERBOSE=1 timeout 600 python tools/probe_stationary.py 6 128 1e-9 22 > gpurun_out/probe_stat_D6.log 2>&1; grep -v "^\[" gpurun_out/probe_stat_D6.log | tail -24; grep "^\[stat\]" gpurun_out/probe_stat_D6.log | tail -8
timeout 900 python -m pytest tests/test_gpu_options.py tests/test_gpu_shapes.py tests/test_gpu_iterative.py -m gpu -q > gpurun_out/opts1.log 2>&1; echo "opts rc=$?"; tail -25 gpurun_out/opts1.log
timeout 2400 python tools/guard/run_guarded.py --timeout 900 --sync 0 --align 256 tests/test_gpu_00_reference_linalg_tests.py tests/test_gpu_00_reference_linalg_tests.py tests/test_gpu_iterative.py tests/test_gpu_shapes.py tests/test_gpu_dist.py tests/test_gpu_scripts.py
