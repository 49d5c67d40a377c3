cd /root/repo
python tools/probe_strip.py 4608 32,64 2>&1 | grep rows768 > gpurun_out/r4f_strip4608.log; cat gpurun_out/r4f_strip4608.log
python tools/probe_strip.py 16384 32,64 2>&1 | grep rows768 > gpurun_out/r4f_strip16384.log; cat gpurun_out/r4f_strip16384.log
OPTS=rows_deep_prefetch=0 python tools/probe_strip.py 4608 32,64 2>&1 | grep rows768 > gpurun_out/r4f_strip4608_nodeep.log; cat gpurun_out/r4f_strip4608_nodeep.log
python tools/probe_sweep_conv.py 8 256 6 > gpurun_out/r4f_d8.log 2>&1; tail -3 gpurun_out/r4f_d8.log
python tools/probe_sweep_conv.py 6 128 6 > gpurun_out/r4f_d6.log 2>&1; tail -2 gpurun_out/r4f_d6.log
timeout 1500 python tools/probe_sweep_conv.py 8 384 4 c128 > gpurun_out/r4f_cfg4.log 2>&1; tail -5 gpurun_out/r4f_cfg4.log
