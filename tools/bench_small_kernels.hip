// Micro-benchmark of the single-workgroup kernels of the latency-bound chains (Cholesky-QR step, 64 x 64 LDS eigensolver).
// The kernels live in an anonymous namespace of jacobi_internal.h / jacobi_core.hip / eigh.hip, so this translation unit includes those files and links against the
// other objects of the library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipeps-torch_amd/csrc -Iinclude tools/bench_small_kernels.hip \
//         peps-torch_amd/csrc/build/{ctm_runtime,gemm_f64,tensor_ops,contract,layer2,ctm_ops,backward}.o -o tools/bin/bench_small_kernels
#define CTM_KERNEL_CLOCKS 1      // phase clocks of the kernels (device buffer ctm_dbg_clocks / stat words of the eigensolver)
#include "../peps-torch_amd/csrc/jacobi_core.hip"
#include "../peps-torch_amd/csrc/eigh.hip"
#include <cstdio>
#include <random>

template <class F> static double time_us(F&& launch, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return 1e3 * ms / reps;
}

int main() {
    const int M = 64, n = 1024;
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd;
    std::vector<double> W((size_t)M * n), G(M * M), Gd(M * M);
    for (auto& x : W) x = nd(rng);
    // rows with a decaying scale and some dependence (what a Cholesky-QR pass of the iteration sees)
    for (int i = 0; i < M; ++i) for (int c = 0; c < n; ++c) W[(size_t)i * n + c] = std::pow(0.8, i) * (W[(size_t)i * n + c] + (i ? 0.5 * W[(size_t)(i - 1) * n + c] : 0.0));
    for (int i = 0; i < M; ++i) for (int j = 0; j < M; ++j) { double s = 0; for (int c = 0; c < n; ++c) s += W[(size_t)i * n + c] * W[(size_t)j * n + c]; G[i * M + j] = s; }
    // nearly diagonal Gram (second Jacobi sweep)
    for (int i = 0; i < M; ++i) for (int j = 0; j < M; ++j) Gd[i * M + j] = (i == j) ? std::pow(0.8, i) : 1e-6 * G[i * M + j];
    double *dG, *dGd, *dOut, *dStatus, *dJ; int *dFlag, *dFlags; unsigned long long* dStat;
    hipMalloc(&dG, sizeof(double) * M * M); hipMalloc(&dGd, sizeof(double) * M * M); hipMalloc(&dOut, sizeof(double) * M * M);
    hipMalloc(&dStatus, sizeof(double) * 16); hipMalloc(&dFlag, sizeof(int) * 4); hipMalloc(&dJ, sizeof(double) * M * M);
    hipMalloc(&dFlags, sizeof(int) * 4); hipMalloc(&dStat, sizeof(unsigned long long) * 16);
    hipMemcpy(dG, G.data(), sizeof(double) * M * M, hipMemcpyHostToDevice);
    hipMemcpy(dGd, Gd.data(), sizeof(double) * M * M, hipMemcpyHostToDevice);
    hipMemset(dFlag, 0, sizeof(int) * 4); hipMemset(dStat, 0, sizeof(unsigned long long) * 16);
    for (int mode = 0; mode < 2; ++mode) {
        const double t = time_us([&] { hipLaunchKernelGGL(chol64_scaled_inv_kernel<64>, dim3(1), dim3(64), 0, 0, (const double*)dG, dOut, dStatus, dFlag, mode); }, 200);
        std::vector<double> L(M * M); double st[3];
        hipMemcpy(L.data(), dOut, sizeof(double) * M * M, hipMemcpyDeviceToHost); hipMemcpy(st, dStatus + 3 * 0, sizeof(st), hipMemcpyDeviceToHost);
        // check: out G out^T = I
        double dev = 0;
        for (int i = 0; i < M; ++i) for (int j = 0; j < M; ++j) {
            double s = 0;
            for (int a = 0; a < M; ++a) { double t2 = 0; for (int b = 0; b < M; ++b) t2 += G[a * M + b] * L[j * M + b]; s += L[i * M + a] * t2; }
            dev = std::max(dev, std::fabs(s - (i == j ? 1.0 : 0.0)));
        }
        double ck[2]; hipMemcpyFromSymbol(ck, HIP_SYMBOL(ctm_dbg_clocks), sizeof(ck));
        printf("chol64_scaled_inv_kernel mode %d: %.1f us   |L^-1 G L^-T - I| = %.2e   clocks: factorisation %.0f, inversion %.0f\n", mode, t, dev, ck[0], ck[1]);
    }
    {
        const double t = time_us([&] { hipLaunchKernelGGL(pivchol64_inv_kernel, dim3(1), dim3(64), 0, 0, (const double*)dG, 1e-10, dOut, dStatus); }, 100);
        std::vector<double> Mo(M * M); double st;
        hipMemcpy(Mo.data(), dOut, sizeof(double) * M * M, hipMemcpyDeviceToHost); hipMemcpy(&st, dStatus, sizeof(st), hipMemcpyDeviceToHost);
        const int rank = (int)st;
        double dev = 0;
        for (int i = 0; i < M; ++i) for (int j2 = 0; j2 < M; ++j2) {
            double s2 = 0;
            for (int a = 0; a < M; ++a) { double t2 = 0; for (int b = 0; b < M; ++b) t2 += G[a * M + b] * Mo[j2 * M + b]; s2 += Mo[i * M + a] * t2; }
            dev = std::max(dev, std::fabs(s2 - ((i == j2 && i < rank) ? 1.0 : 0.0)));
        }
        printf("pivchol64_inv_kernel: %.1f us   rank %d   |Mo G Mo^T - I_rank| = %.2e\n", t, rank, dev);
    }
    {
        const double t = time_us([&] { hipLaunchKernelGGL(sym64_lmax_kernel, dim3(1), dim3(64), 0, 0, (const double*)dG, 48, dStatus); }, 100);
        double o[2]; hipMemcpy(o, dStatus, sizeof(o), hipMemcpyDeviceToHost);
        // reference: power iteration on the host
        std::vector<double> x(M, 1.0), y(M); double lam = 0;
        for (int it = 0; it < 2000; ++it) { double nn = 0; for (int i = 0; i < M; ++i) { y[i] = 0; for (int k = 0; k < M; ++k) y[i] += G[i * M + k] * x[k]; nn += y[i] * y[i]; } lam = std::sqrt(nn); for (int i = 0; i < M; ++i) x[i] = y[i] / lam; }
        printf("sym64_lmax_kernel: %.1f us   |G|_F = %.6e  estimate %.12e  (lambda_max %.12e)\n", t, o[0], o[1], lam);
    }
    for (int which = 0; which < 2; ++which)
        for (int sweeps = 1; sweeps <= 3; ++sweeps) {
            SmallEigParams sp;
            sp.G = which ? dGd : dG; sp.nsplit = 1; sp.split_stride = 0; sp.J = dJ; sp.m = 64; sp.tol = 1e-15; sp.max_sweeps = sweeps;
            sp.tau2 = 0.0; sp.tau_both = 0; sp.stat_rel = dStat; sp.stat_abs = dStat + 1; sp.flags = dFlags;
            const double t = time_us([&] { hipLaunchKernelGGL(small_eig64_kernel<2>, dim3(1), dim3(512), 0, 0, sp); }, 100);
            long long ck[4]; hipMemcpy(ck, dStat + 4, sizeof(ck), hipMemcpyDeviceToHost);
            printf("small_eig64_kernel<2> %s Gram, max_sweeps %d: %.1f us   clocks per round (wave 0): loads %.0f, rotation %.0f, updates %.0f, barrier %.0f\n",
                   which ? "nearly diagonal" : "generic", sweeps, t, ck[0] / (63.0 * sweeps), ck[1] / (63.0 * sweeps), ck[2] / (63.0 * sweeps), ck[3] / (63.0 * sweeps));
        }
    return 0;
}
