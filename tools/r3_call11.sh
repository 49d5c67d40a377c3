#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c11; mkdir -p $O
cd $ROOT
echo "--- default"; timeout 120 python tools/dbg_d6.py 2>&1 | grep -v amdgpu | tail -3
echo "--- lz_async=0"; timeout 120 python tools/dbg_d6.py lz_async=0 2>&1 | grep -v amdgpu | tail -3
echo "--- heavy_serial=0"; timeout 120 python tools/dbg_d6.py heavy_serial=0 2>&1 | grep -v amdgpu | tail -3
echo "--- verbose"; timeout 120 python tools/dbg_d6.py jacobi_verbose=1 > $O/verbose.txt 2>&1; grep -n -i "nan\|flagged\|breakdown" $O/verbose.txt | head -10; grep -c "\[lz\]" $O/verbose.txt
run() { echo "=== $1 | $2" >> $O/ab.txt; CTM_ENGINE_OPTS="$1" timeout 300 python tools/probe_sweep_conv.py 8 256 4 $2 >> $O/ab.txt 2>&1; }
run "" ""
run "" "rows_target_wgs=512"
run "" "heavy_serial=0"
grep -v amdgpu $O/ab.txt
