cd /root/repo
for x in 0 1 2; do python tools/probe_c4v_signed.py old 30 eigh_orth_extra_blocks=$x > gpurun_out/r4i_c4v_x$x.log 2>&1; python - <<P
import re
t=[float(l.split()[2]) for l in open('gpurun_out/r4i_c4v_x$x.log') if l.startswith('sweep')]
mv=[x for x in t[1:] if x>1.5]
print('extra blocks $x: moving sweeps',len(mv),'mean ms',sum(mv)/max(len(mv),1),'total ms',sum(t))
P
done
python tools/probe_sweep_conv.py 8 256 5 > gpurun_out/r4i_d8.log 2>&1; tail -2 gpurun_out/r4i_d8.log | cut -c1-200
python tools/probe_sweep_conv.py 6 128 5 > gpurun_out/r4i_d6.log 2>&1; tail -2 gpurun_out/r4i_d6.log | cut -c1-200
