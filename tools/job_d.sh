cd /root/repo
python -m pytest tests/test_gpu_iterative.py -x -q -k "krylov_solver_variants" > gpurun_out/r4d_variants.log 2>&1; tail -3 gpurun_out/r4d_variants.log
python tools/probe_sweep_conv.py 8 256 7 lz_block=64 > gpurun_out/r4d_d8_b64.log 2>&1; tail -4 gpurun_out/r4d_d8_b64.log
python tools/probe_sweep_conv.py 8 256 7 lz_block=32 > gpurun_out/r4d_d8_b32.log 2>&1; tail -4 gpurun_out/r4d_d8_b32.log
python tools/probe_sweep_conv.py 8 256 7 lz_block=32 jacobi_cross_only=1 > gpurun_out/r4d_d8_b32x.log 2>&1; tail -4 gpurun_out/r4d_d8_b32x.log
python tools/probe_sweep_conv.py 6 128 7 lz_block=64 > gpurun_out/r4d_d6_b64.log 2>&1; tail -3 gpurun_out/r4d_d6_b64.log
python tools/probe_sweep_conv.py 6 128 7 lz_block=32 > gpurun_out/r4d_d6_b32.log 2>&1; tail -3 gpurun_out/r4d_d6_b32.log
