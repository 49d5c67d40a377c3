cd /root/repo
timeout 600 python -m pytest tests/test_gpu_iterative.py -x -q -k "many_panel or krylov_solver_variants" 2>&1 | tail -6
timeout 300 python tools/probe_sweep_conv.py 6 128 5 jacobi_persist=1 2>&1 | tail -2 | cut -c1-200
timeout 300 python tools/probe_sweep_conv.py 6 128 5 2>&1 | tail -1 | cut -c1-60
timeout 400 python tools/probe_sweep_conv.py 8 256 5 jacobi_persist=1 2>&1 | tail -2 | cut -c1-200
timeout 400 python tools/probe_sweep_conv.py 8 256 5 2>&1 | tail -1 | cut -c1-60
