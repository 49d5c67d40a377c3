#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c9; mkdir -p $O
cd $ROOT
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r3c9/bench.json').read().strip().splitlines()[-1])
    print('value',d['value'],'ms',d['ms_per_step'],'full_rank',d['full_rank']['value'],d['full_rank']['ms_per_step'])
    print('energy',d['full_rank'].get('energy'))
    for k,v in d.get('other_configs',{}).items(): print(k, {kk:v.get(kk) for kk in ('value','ms_per_step','error','wall_s_incl_warmup','moving_environment','stationary_environment')})
    print('roof', d['roofline']['frac'], d['full_rank']['roofline']['frac'])
except Exception as e: print('parse fail',e)
PY
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -5 $O/tests.txt
