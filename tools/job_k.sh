cd /root/repo
tools/bin/bench_small_kernels 2>&1 | grep small_eig64 | head -2
python bench.py --no-other-configs --no-energy --no-cpu-baseline > gpurun_out/r4k_bench.json 2> gpurun_out/r4k_bench.err; python - <<P
import json
l=json.loads(open('gpurun_out/r4k_bench.json').read().strip().splitlines()[-1]); r=l['roofline']
print(l['value'], r['traffic'], r.get('traffic_source','')[:80]); print(r['full_rank_ms_per_step'], r['full_rank_traffic'])
P
