cd /root/repo
python -m pytest tests/test_gpu_iterative.py tests/test_gpu_complex.py -x -q > gpurun_out/r4j_tests.log 2>&1; tail -4 gpurun_out/r4j_tests.log
python tools/probe_strip.py 4608 32 2>&1 | grep rows768 | tee gpurun_out/r4j_strip_a.log
OPTS=rows_min_klen_hbm=288 python tools/probe_strip.py 4608 32 2>&1 | grep rows768 | tee gpurun_out/r4j_strip_b.log
OPTS=rows_min_klen_hbm=384 python tools/probe_strip.py 4608 32 2>&1 | grep rows768 | tee gpurun_out/r4j_strip_c.log
python tools/probe_sweep_conv.py 6 128 5 rows_min_klen_hbm=288 2>&1 | tail -1 | cut -c1-120
timeout 900 python tools/probe_sweep_conv.py 8 384 3 c128 > gpurun_out/r4j_cfg4_stack.log 2>&1; tail -2 gpurun_out/r4j_cfg4_stack.log | cut -c1-220
timeout 900 python tools/probe_sweep_conv.py 8 384 3 c128 lz_block_c=32 > gpurun_out/r4j_cfg4_b32.log 2>&1; tail -2 gpurun_out/r4j_cfg4_b32.log | cut -c1-220
