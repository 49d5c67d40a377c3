cd /root/repo
python -m pytest tests/test_gpu_iterative.py tests/test_gpu_gemm_rows.py tests/test_gpu_primitives.py -x -q > gpurun_out/r4e_tests.log 2>&1; tail -5 gpurun_out/r4e_tests.log
python tools/probe_strip.py 4608 32,64 2>&1 | grep rows768 > gpurun_out/r4e_strip4608.log; cat gpurun_out/r4e_strip4608.log
python tools/probe_strip.py 16384 32,64 2>&1 | grep rows768 > gpurun_out/r4e_strip16384.log; cat gpurun_out/r4e_strip16384.log
python tools/probe_sweep_conv.py 8 256 6 lz_block=32 > gpurun_out/r4e_d8_a.log 2>&1; tail -3 gpurun_out/r4e_d8_a.log
python tools/probe_sweep_conv.py 8 256 6 lz_block=32 jacobi_rot_apply=1 > gpurun_out/r4e_d8_b.log 2>&1; tail -3 gpurun_out/r4e_d8_b.log
python tools/probe_sweep_conv.py 8 256 6 lz_block=32 jacobi_rot_apply=1 jacobi_cross_only=1 > gpurun_out/r4e_d8_c.log 2>&1; tail -3 gpurun_out/r4e_d8_c.log
python tools/probe_sweep_conv.py 6 128 6 lz_block=32 > gpurun_out/r4e_d6_a.log 2>&1; tail -2 gpurun_out/r4e_d6_a.log
python tools/probe_sweep_conv.py 6 128 6 lz_block=32 jacobi_rot_apply=1 > gpurun_out/r4e_d6_b.log 2>&1; tail -2 gpurun_out/r4e_d6_b.log
python tools/probe_sweep_conv.py 6 128 6 lz_block=32 jacobi_rot_apply=1 jacobi_cross_only=1 > gpurun_out/r4e_d6_c.log 2>&1; tail -2 gpurun_out/r4e_d6_c.log
