cd /root/repo
python -m pytest tests/test_gpu_dist.py -x -q -k rccl > gpurun_out/r4b_rccl.log 2>&1; tail -15 gpurun_out/r4b_rccl.log
python tools/probe_sweep_conv.py 8 256 8 > gpurun_out/r4b_d8_base.log 2>&1; tail -9 gpurun_out/r4b_d8_base.log
python tools/probe_sweep_conv.py 8 256 8 jacobi_cross_only=1 > gpurun_out/r4b_d8_cross.log 2>&1; tail -9 gpurun_out/r4b_d8_cross.log
python tools/dump_unit_spectrum.py 6 128 4 gpurun_out/spec_d6.npy > gpurun_out/r4b_spec.log 2>&1; tail -3 gpurun_out/r4b_spec.log
timeout 1500 python tools/probe_sweep_conv.py 8 384 8 c128 > gpurun_out/r4b_cfg4.log 2>&1; tail -10 gpurun_out/r4b_cfg4.log
