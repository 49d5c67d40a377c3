#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c14; mkdir -p $O
cd $ROOT
for hs in 1 0; do
  timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-energy --steps 4 --opt heavy_serial=$hs > $O/bench_hs$hs.json 2> $O/bench_hs$hs.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r3c14/bench_hs$hs.json').read().strip().splitlines()[-1])
r=d['roofline']; f=d['full_rank']; fr=f['roofline']
print('heavy_serial=$hs value %.3f (%.0f ms) frac %.3f union %.3f serial %s | full_rank %.4f (%.0f ms) kernel %s frac %.3f union %.3f serial %s' % (d['value'], d['ms_per_step'], r['frac'], r.get('frac_union',0), r.get('serial_pass',{}).get('dominant',{}).get('frac'), f['value'], f['ms_per_step'], fr['kernel'][:20], fr['frac'], fr.get('frac_union',0), fr.get('serial_pass',{}).get('dominant',{}).get('frac')))
print(' phase_s', d['phase_s'], ' | full', f['phase_s'])
PY
done
