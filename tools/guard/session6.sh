VERBOSE=6 timeout 600 python tools/probe_stationary.py 6 128 1e-9 16 > gpurun_out/probe_stat_D6.log 2>&1; grep "^\[stat\]\|^sweep" gpurun_out/probe_stat_D6.log | cut -c1-170 | tail -60
VERBOSE=6 timeout 600 python tools/probe_stationary.py 4 64 1e-9 16 > gpurun_out/probe_stat_D4.log 2>&1; grep "^\[stat\]\|^sweep" gpurun_out/probe_stat_D4.log | cut -c1-170 | tail -40
