#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c10; mkdir -p $O
cd $ROOT
# determinism of the low-rank n = 8192 run: same process configuration twice, with and without the fused combine / shared heavy stream
for tag in a b; do timeout 200 python tools/check_dist_gpu.py $O/det_$tag.json > /dev/null 2>&1; done
CTM_ENGINE_OPTS="rows_fused_reduce=0" timeout 200 python tools/check_dist_gpu.py $O/det_nofuse.json > /dev/null 2>&1
CTM_ENGINE_OPTS="heavy_serial=0" timeout 200 python tools/check_dist_gpu.py $O/det_noheavy.json > /dev/null 2>&1
CTM_ENGINE_OPTS="heavy_serial=0,rows_fused_reduce=0" timeout 200 python tools/check_dist_gpu.py $O/det_none.json > /dev/null 2>&1
CTM_LARGE_N_UNITS=2 CTM_ENGINE_OPTS="heavy_serial=0" timeout 200 python tools/check_dist_gpu.py $O/det_2units.json > /dev/null 2>&1
python - <<'PY'
import json,glob
ref=None
for f in sorted(glob.glob('gpurun_out/r3c10/det_*.json')):
    d=json.load(open(f)); 
    k=[x for x in d if x not in('checksum','ncol')][0]
    print(f.split('/')[-1], d['checksum'], d[k][1], d[k][5])
PY
run() { echo "=== $1 | $2" >> $O/ab.txt; CTM_ENGINE_OPTS="$1" timeout 300 python tools/probe_sweep_conv.py 8 256 4 $2 >> $O/ab.txt 2>&1; }
run "" ""
run "" "rows_target_wgs=512"
run "" "heavy_serial=0"
run "" "rows_target_wgs=512 heavy_min_flops=4e9"
grep -v amdgpu $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_iterative.py -x -q -k "signed or iterative or sweep_invariances" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests.txt
