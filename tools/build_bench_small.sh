#!/bin/bash
# builds tools/bin/bench_small_kernels (see tools/bench_small_kernels.hip)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Ipeps-torch_amd/csrc -Iinclude -c tools/bench_small_kernels.hip -o tools/bin/bench_small_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 tools/bin/bench_small_kernels.o peps-torch_amd/csrc/build/{ctm_runtime,gemm_f64,tensor_ops,contract,layer2,ctm_ops,backward,svd_leading,ctm_move}.o -o tools/bin/bench_small_kernels
