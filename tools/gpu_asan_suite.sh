#!/bin/bash
# Full GPU suite with the HOST side of libctm_hip under AddressSanitizer (csrc/build.py --asan), engine options from $1
# (e.g. rows_min_klen=576).  Logs under gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python peps-torch_amd/csrc/build.py --asan > /dev/null || exit 1      # (rebuilt when a source is newer: hipcc is on the GPU box too)
export CTM_LIB=$PWD/peps-torch_amd/libctm_hip_asan.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=1:halt_on_error=1:log_path=$PWD/gpurun_out/asan
export CTM_ENGINE_OPTS="$1"
shift
ulimit -c 0
# (the runtime intercepts dlopen, so the RUNPATH of libtorch no longer finds its siblings: libcaffe2_nvrtc.so)
export LD_LIBRARY_PATH=$(python -c "import os, torch; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))"):$LD_LIBRARY_PATH
# (libstdc++ preloaded as well: the runtime resolves the real __cxa_throw when it initialises, before python has loaded any C++ library)
LD_PRELOAD="$(python peps-torch_amd/csrc/build.py --asan-runtime) $(gcc -print-file-name=libstdc++.so.6)" python -X faulthandler -m pytest "$@" 2>&1 | tail -60
