#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r3c17; mkdir -p $O
cd $ROOT
timeout 2700 python -m pytest tests -m gpu -q --durations=8 > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -14 $O/tests.txt
