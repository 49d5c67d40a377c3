#!/bin/bash
# The reference's own linalg unit tests (gradchecks of the native SVD / eigh nodes) N times, uncaptured: hunting the rare abort of a full
# suite run inside ctm_svd_backward (tests/test_gpu_00_reference_linalg_tests.py::test_SVDGESDD_COMPLEX_random).
cd "$(dirname "$0")/.."
N=${1:-12}
for i in $(seq 1 $N); do
  python -X faulthandler -m pytest tests/test_gpu_00_reference_linalg_tests.py tests/test_gpu_ad.py -x -q -s -p no:cacheprovider > gpurun_out/loop_$i.log 2>&1
  rc=$?
  echo "iteration $i rc=$rc $(tail -1 gpurun_out/loop_$i.log | cut -c1-80)"
  if [ $rc -ne 0 ]; then grep -n -i -E "fault|abort|terminate|corrupt|free\(\)|malloc|HSA|error" gpurun_out/loop_$i.log | head -20; fi
done
