cd /root/repo
python -m pytest tests/test_gpu_dist.py -x -q -k rccl > gpurun_out/r4c_rccl.log 2>&1; tail -3 gpurun_out/r4c_rccl.log
python bench.py --config generic_D6_chi128 --steps 3 > gpurun_out/r4c_bench_d6.json 2> gpurun_out/r4c_bench_d6.err; tail -c 3000 gpurun_out/r4c_bench_d6.json
timeout 2400 python tools/probe_sweep_conv.py 8 384 6 c128 > gpurun_out/r4c_cfg4.log 2>&1; tail -10 gpurun_out/r4c_cfg4.log
