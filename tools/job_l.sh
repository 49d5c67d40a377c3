cd /root/repo
for u in 2 3; do CTM_LARGE_N_UNITS=$u python tools/probe_sweep_conv.py 8 256 5 2>&1 | tail -1 | cut -c1-60; done
python tools/probe_sweep_conv.py 8 256 5 jacobi_cross_only=1 2>&1 | tail -1 | cut -c1-60
python tools/probe_sweep_conv.py 8 256 5 2>&1 | tail -1 | cut -c1-60
python tools/probe_sweep_conv.py 6 128 5 jacobi_cross_only=1 2>&1 | tail -1 | cut -c1-60
python tools/probe_sweep_conv.py 6 128 5 2>&1 | tail -1 | cut -c1-60
python tools/probe_sweep_conv.py 8 256 5 rows_target_wgs=1024 2>&1 | tail -1 | cut -c1-60
python tools/probe_sweep_conv.py 8 256 5 rows_target_wgs=512 2>&1 | tail -1 | cut -c1-60
