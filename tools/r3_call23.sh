#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c23; mkdir -p $O
cd $ROOT
for o in "" "--opt si_quad_exit=0"; do python bench.py --config c4v_D4_chi64 --no-cpu-baseline $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c4v [$o]', 'value %.0f'%d['value'], 'moving', d.get('moving_environment'), 'stationary', d.get('stationary_environment',{}).get('sweeps_per_sec'))"; done
for o in "" "--opt si_quad_exit=0"; do python bench.py --no-cpu-baseline --no-other-configs --no-energy --no-full-rank --no-serial-pass --steps 4 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('default [$o]', 'value %.4f'%d['value'], d['svd'])"; done
timeout 1500 python -m pytest tests/test_gpu_c4v.py tests/test_gpu_iterative.py tests/test_gpu_generic.py tests/test_gpu_complex.py -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
