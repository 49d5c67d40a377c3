mkdir -p gpurun_out/guard
timeout 1200 python -m pytest tests/test_gpu_stationary.py -m gpu -x -q -s > gpurun_out/stationary1.log 2>&1; echo "stationary rc=$?"; tail -30 gpurun_out/stationary1.log
timeout 900 python -m pytest tests/test_gpu_c4v.py tests/test_gpu_primitives.py -m gpu -x -q > gpurun_out/c4v1.log 2>&1; echo "c4v rc=$?"; tail -15 gpurun_out/c4v1.log
timeout 2700 python tools/guard/run_guarded.py --timeout 600 tests/test_gpu_backward.py tests/test_gpu_00_reference_linalg_tests.py tests/test_gpu_primitives.py tests/test_gpu_complex.py tests/test_gpu_gemm_rows.py tests/test_gpu_ad.py tests/test_gpu_generic.py tests/test_gpu_c4v.py
timeout 1200 python -m pytest tests/test_gpu_bench_ranks.py -m gpu -x -q > gpurun_out/ranks1.log 2>&1; echo "ranks rc=$?"; tail -30 gpurun_out/ranks1.log
