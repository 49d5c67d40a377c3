cd /root/repo
python tools/probe_c4v_signed.py new 30 > gpurun_out/r4h_c4v_new.log 2>&1; tail -12 gpurun_out/r4h_c4v_new.log
python tools/probe_c4v_signed.py old 30 > gpurun_out/r4h_c4v_old.log 2>&1; tail -8 gpurun_out/r4h_c4v_old.log
python tools/probe_c4v_signed.py new 8 jacobi_verbose=1 2>&1 | grep -E "eigh|sweep" | tail -40 > gpurun_out/r4h_c4v_new_verbose.log; tail -40 gpurun_out/r4h_c4v_new_verbose.log | cut -c1-200
python tools/probe_c4v_signed.py new 30 lz_block=64 > gpurun_out/r4h_c4v_new_b64.log 2>&1; tail -4 gpurun_out/r4h_c4v_new_b64.log
