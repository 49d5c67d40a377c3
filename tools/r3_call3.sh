#!/bin/bash
# round 3, GPU call 3: correctness of the fused split-K combine / async recurrence, then A/B timings on the signed D8 chi256 state
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c3; mkdir -p $O
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_iterative.py tests/test_gpu_generic.py -x -q > $O/tests1.txt 2>&1; echo "tests1 rc=$?" 
tail -3 $O/tests1.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "generic_unit_at_full_size or sweep_invariances" > $O/tests2.txt 2>&1; echo "tests2 rc=$?"
tail -3 $O/tests2.txt
for cfg in "" "rows_fused_reduce=0" "lz_async=0" "lz_local_project=0" "rows_fused_reduce=0 lz_async=0 lz_local_project=0"; do
  echo "=== $cfg" >> $O/ab.txt
  timeout 300 python tools/probe_sweep_conv.py 8 256 5 $cfg >> $O/ab.txt 2>&1
done
cat $O/ab.txt
