#!/bin/bash
# The GPU suite in the file order of rounds 1-3 (the order in which the rare abort inside ctm_svd_backward was seen):
# tests/test_gpu_00_reference_linalg_tests.py at the position its old name (test_gpu_reference_linalg_tests.py) sorted to, i.e. after
# test_gpu_primitives.py, instead of first.  Usage (GPU box): bash tools/run_suite_original_order.sh [extra pytest args]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
files=""
for f in $(ls tests/test_*.py | sort); do
  case $f in
    tests/test_gpu_00_reference_linalg_tests.py) ;;
    tests/test_gpu_scripts.py) files="$files tests/test_gpu_00_reference_linalg_tests.py $f" ;;
    *) files="$files $f" ;;
  esac
done
# (--soak: with the long repeats, wall-clock reports and optimiser script variants the driver's selection leaves out, tests/conftest.py)
exec python -X faulthandler -m pytest $files -m gpu --soak -q -p no:cacheprovider "$@"
