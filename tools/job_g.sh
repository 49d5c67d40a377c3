cd /root/repo
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r4g_tests.log 2>&1; tail -8 gpurun_out/r4g_tests.log
python bench.py > gpurun_out/r4g_bench.json 2> gpurun_out/r4g_bench.err; tail -c 5000 gpurun_out/r4g_bench.json; tail -3 gpurun_out/r4g_bench.err | cut -c1-300
