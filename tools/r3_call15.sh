#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c15; mkdir -p $O
cd $ROOT
timeout 2700 python -m pytest tests -m gpu -x -q --durations=15 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -25 $O/tests.txt
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -4 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3c15/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; f=d['full_rank']; fr=f['roofline']
print('value %.3f (%.0f ms) frac %.3f union %.3f | full_rank %.4f (%.0f ms) frac %.3f union %.3f' % (d['value'], d['ms_per_step'], r['frac'], r.get('frac_union',0), f['value'], f['ms_per_step'], fr['frac'], fr.get('frac_union',0)))
print('energy', f.get('energy'))
for k,v in d.get('other_configs',{}).items(): print(k, {kk:v.get(kk) for kk in ('value','ms_per_step','error','moving_environment','stationary_environment')})
PY
