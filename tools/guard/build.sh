#!/bin/bash
# Builds the guard shim and its self-test into tools/bin/ (both git-ignored; they travel to the GPU box with the snapshot).
cd "$(dirname "$0")/../.."
mkdir -p tools/bin
g++ -O2 -Wno-unused-result -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/guard/guard_malloc.cpp -o tools/bin/libguard_malloc.so -ldl || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/guard/selftest.hip -o tools/bin/guard_selftest || exit 1
echo built
