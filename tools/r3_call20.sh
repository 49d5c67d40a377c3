#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c20; mkdir -p $O
cd $ROOT
run() { python bench.py --config c4v_D4_chi64 --no-cpu-baseline --warmup 1 --steps 60 $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'value %.0f'%d['value'], 'moving', d.get('moving_environment'), 'stationary', d.get('stationary_environment',{}).get('sweeps_per_sec'))"; }
run ""
run "--opt jacobi_inner_sweeps=1"
run "--opt jacobi_inner_sweeps=3"
run "--opt eig64_bpt=1"
run "--opt eig64_bpt=4"
run "--opt jacobi_block=16"
run "--opt jacobi_gram_kmin=128"
timeout 900 python -m pytest "tests/test_gpu_iterative.py::test_krylov_solver_variants_agree" "tests/test_gpu_primitives.py::test_in_launch_split_k_combine_equals_the_separate_reduce_kernel" -x -q > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
