// Scheduling probe (development tool, not product): does a chain of small latency-bound kernels make progress while chip-filling
// kernels of another stream are resident?  Compares: plain streams, a high-priority stream for the chain, and CU-masked streams
// (hipExtStreamCreateWithCUMask: the chain on a few reserved CUs, the heavy stream on the complement).
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_cumask.hip -o tools/bin/probe_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <set>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));

// chip-filling kernel: `wgs` workgroups of 256 threads holding `lds` bytes of LDS, each issuing `iters` FP64 MFMAs per wave
__global__ __launch_bounds__(256) void heavy_kernel(double* out, int iters) {
    extern __shared__ double sm[];
    v4d acc = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-5;
    sm[threadIdx.x] = a;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc, 0, 0, 0);
    }
    if (acc[0] == 12345.678) out[blockIdx.x] = acc[0] + sm[(threadIdx.x + 1) & 255];
}

// latency-bound kernel: few workgroups, big LDS, spins for `ns` nanoseconds of wall clock; records where it ran
__global__ __launch_bounds__(512) void light_kernel(unsigned* where, long long ns, int record) {
    extern __shared__ double sm[];
    sm[threadIdx.x] = 1.0;
    __syncthreads();
    if (record && threadIdx.x == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID, 4 bits
        unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
        where[blockIdx.x] = (xcc << 24) | (hw & 0xffffff);
    }
    const long long t0 = wall_clock64();            // 100 MHz
    while ((wall_clock64() - t0) * 10 < ns) { }
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now().time_since_epoch()).count(); }

struct Result { double heavy_ms, light_ms; };

// one experiment: `nheavy` heavy streams each launching `hn` heavy kernels, one light stream launching `ln` chained light kernels
static Result run(std::vector<hipStream_t>& hs, hipStream_t ls, int hn, int ln, int heavy_wgs, int heavy_lds, int heavy_iters,
                  int light_wgs, int light_lds, long long light_ns, double* dout, unsigned* dwhere) {
    CK(hipDeviceSynchronize());
    std::vector<hipEvent_t> he0(hs.size()), he1(hs.size());
    hipEvent_t le0, le1;
    CK(hipEventCreate(&le0)); CK(hipEventCreate(&le1));
    for (size_t i = 0; i < hs.size(); ++i) { CK(hipEventCreate(&he0[i])); CK(hipEventCreate(&he1[i])); }
    for (size_t i = 0; i < hs.size(); ++i) CK(hipEventRecord(he0[i], hs[i]));
    if (ls && ln) CK(hipEventRecord(le0, ls));
    // interleave the submissions the way two host threads would
    int li = 0;
    for (int k = 0; k < hn; ++k) {
        for (size_t i = 0; i < hs.size(); ++i)
            hipLaunchKernelGGL(heavy_kernel, dim3(heavy_wgs), dim3(256), heavy_lds, hs[i], dout, heavy_iters);
        if (ls) for (int q = 0; q < (ln + hn - 1) / std::max(hn, 1) && li < ln; ++q, ++li)
            hipLaunchKernelGGL(light_kernel, dim3(light_wgs), dim3(512), light_lds, ls, dwhere, light_ns, 0);
    }
    if (ls) for (; li < ln; ++li) hipLaunchKernelGGL(light_kernel, dim3(light_wgs), dim3(512), light_lds, ls, dwhere, light_ns, 0);
    for (size_t i = 0; i < hs.size(); ++i) CK(hipEventRecord(he1[i], hs[i]));
    if (ls && ln) CK(hipEventRecord(le1, ls));
    CK(hipDeviceSynchronize());
    Result r{0, 0};
    for (size_t i = 0; i < hs.size(); ++i) { float ms; CK(hipEventElapsedTime(&ms, he0[i], he1[i])); r.heavy_ms = std::max(r.heavy_ms, (double)ms); }
    if (ls && ln) { float ms; CK(hipEventElapsedTime(&ms, le0, le1)); r.light_ms = ms; }
    return r;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    int ncu = 0;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, ncu);
    double* dout; unsigned* dwhere;
    CK(hipMalloc(&dout, 1 << 20)); CK(hipMalloc(&dwhere, 4096 * sizeof(unsigned)));
    CK(hipFuncSetAttribute((const void*)heavy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)light_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int heavy_wgs = 768, heavy_lds = 48 * 1024, light_wgs = 14, light_lds = 100 * 1024;
    const long long light_ns = 40000;
    // calibrate heavy iterations to ~600 us per kernel
    hipStream_t s0; CK(hipStreamCreate(&s0));
    int iters = 2000;
    {
        std::vector<hipStream_t> h1{s0};
        for (int rep = 0; rep < 4; ++rep) {
            Result r = run(h1, nullptr, 10, 0, heavy_wgs, heavy_lds, iters, light_wgs, light_lds, light_ns, dout, dwhere);
            const double per = r.heavy_ms / 10;
            printf("calibrate: iters %d -> %.3f ms per heavy kernel\n", iters, per);
            iters = (int)(iters * 0.6 / per);
        }
    }
    const int HN = 100, LN = 600;      // 100 x 0.6 ms heavy per stream; 600 x 40 us chain = 24 ms + launch gaps
    // ---- where do masked kernels run?  mask bit i -> (xcc, se, cu)
    auto mask_stream = [&](const std::vector<uint32_t>& m) { hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data())); return s; };
    const int words = (ncu + 31) / 32;
    auto show_mask = [&](const char* name, const std::vector<uint32_t>& m) {
        hipStream_t s = mask_stream(m);
        std::set<unsigned> seen;
        for (int rep = 0; rep < 8; ++rep) {
            CK(hipMemsetAsync(dwhere, 0xff, 4096 * sizeof(unsigned), s));
            hipLaunchKernelGGL(light_kernel, dim3(512), dim3(512), 64 * 1024, s, dwhere, 20000LL, 1);
            CK(hipStreamSynchronize(s));
            std::vector<unsigned> h(512); CK(hipMemcpy(h.data(), dwhere, 512 * sizeof(unsigned), hipMemcpyDeviceToHost));
            for (unsigned v : h) {
                const unsigned xcc = v >> 24, hw = v & 0xffffff;
                const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
                seen.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
            }
        }
        printf("mask %-28s -> %zu distinct (xcc,se,sh,cu):", name, seen.size());
        int c = 0; for (unsigned v : seen) { if (c++ < 40) printf(" %u.%u.%u.%u", v >> 16, (v >> 8) & 0xff, (v >> 4) & 0xf, v & 0xf); }
        printf("\n");
        CK(hipStreamDestroy(s));
    };
    // ---- scheduling experiments
    struct Cfg { const char* name; int nheavy; int light_mode; int reserve; };   // light_mode 0 plain, 1 high priority, 2 CU mask (reserve CUs), 3 mask + heavy unmasked
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d greatest %d\n", lo, hi);
    std::vector<Cfg> cfgs = {
        {"heavy alone, 1 stream", 1, -1, 0}, {"heavy alone, 2 streams", 2, -1, 0}, {"light alone", 0, 0, 0},
        {"1 heavy + light, plain", 1, 0, 0}, {"2 heavy + light, plain", 2, 0, 0},
        {"1 heavy + light, high priority", 1, 1, 0}, {"2 heavy + light, high priority", 2, 1, 0},
        {"1 heavy(masked) + light on 8 reserved CUs", 1, 2, 8}, {"1 heavy(masked) + light on 16 reserved CUs", 1, 2, 16},
        {"2 heavy(masked) + light on 16 reserved CUs", 2, 2, 16}, {"2 heavy(masked) + light on 32 reserved CUs", 2, 2, 32},
        {"1 heavy(unmasked) + light masked to 16 CUs", 1, 3, 16},
        {"light alone on 16 masked CUs", 0, 2, 16},
    };
    const bool nomask = argc > 1 && std::string(argv[1]) == "nomask";
    for (const Cfg& c : cfgs) {
        if (nomask && c.light_mode >= 2) continue;
        std::vector<hipStream_t> hs;
        hipStream_t ls = nullptr;
        std::vector<uint32_t> lm(words, 0), hm(words, 0xffffffff);
        if (c.reserve) {
            // reserve CUs spread over the XCDs: assume bit i -> XCD i % 8 (checked by the mask report above); take the LAST bits
            for (int i = 0; i < c.reserve; ++i) { const int bit = ncu - 1 - i; lm[bit / 32] |= 1u << (bit % 32); hm[bit / 32] &= ~(1u << (bit % 32)); }
        }
        for (int i = 0; i < c.nheavy; ++i) {
            hipStream_t s;
            if (c.light_mode == 2) s = mask_stream(hm); else CK(hipStreamCreate(&s));
            hs.push_back(s);
        }
        if (c.light_mode == 0) CK(hipStreamCreate(&ls));
        else if (c.light_mode == 1) CK(hipStreamCreateWithPriority(&ls, hipStreamDefault, hi));
        else if (c.light_mode >= 2) ls = mask_stream(lm);
        Result r = run(hs, ls, c.nheavy ? HN : 0, ls ? LN : 0, heavy_wgs, heavy_lds, iters, light_wgs, light_lds, light_ns, dout, dwhere);
        printf("%-46s heavy %8.2f ms (%.3f per kernel per stream)   light chain %8.2f ms (%.1f us per link)\n", c.name, r.heavy_ms,
               c.nheavy ? r.heavy_ms / HN : 0.0, r.light_ms, ls ? 1e3 * r.light_ms / LN : 0.0);
        for (auto s : hs) CK(hipStreamDestroy(s));
        if (ls) CK(hipStreamDestroy(ls));
    }
    // ---- heavy kernels with shorter-lived workgroups (4x the workgroups, 1/4 the work each): does the chain get in sooner?
    {
        std::vector<hipStream_t> hs(2); CK(hipStreamCreate(&hs[0])); CK(hipStreamCreate(&hs[1]));
        hipStream_t ls; CK(hipStreamCreateWithPriority(&ls, hipStreamDefault, hi));
        Result r = run(hs, ls, HN, LN, heavy_wgs * 4, heavy_lds, iters / 4, light_wgs, light_lds, light_ns, dout, dwhere);
        printf("%-46s heavy %8.2f ms (%.3f per kernel per stream)   light chain %8.2f ms (%.1f us per link)\n", "2 heavy (3072 short wgs) + light high prio", r.heavy_ms, r.heavy_ms / HN, r.light_ms, 1e3 * r.light_ms / LN);
        hipStream_t lp; CK(hipStreamCreate(&lp));
        r = run(hs, lp, HN, LN, heavy_wgs * 4, heavy_lds, iters / 4, light_wgs, light_lds, light_ns, dout, dwhere);
        printf("%-46s heavy %8.2f ms (%.3f per kernel per stream)   light chain %8.2f ms (%.1f us per link)\n", "2 heavy (3072 short wgs) + light plain", r.heavy_ms, r.heavy_ms / HN, r.light_ms, 1e3 * r.light_ms / LN);
        // heavy kernels leaving LDS room: 2 per CU x 48 KB
        r = run(hs, ls, HN, LN, 512, heavy_lds, iters * 3 / 2, light_wgs, light_lds, light_ns, dout, dwhere);
        printf("%-46s heavy %8.2f ms (%.3f per kernel per stream)   light chain %8.2f ms (%.1f us per link)\n", "2 heavy (512 wgs: LDS room) + light high prio", r.heavy_ms, r.heavy_ms / HN, r.light_ms, 1e3 * r.light_ms / LN);
        // co-residency: heavy kernels that never fill a CU (2 x 48 KB of LDS, 8 of 32 wave slots) + light kernels that fit beside them (48 KB)
        std::vector<hipStream_t> h1{hs[0]};
        r = run(h1, lp, HN, LN, 512, heavy_lds, iters * 3 / 2, light_wgs, 48 * 1024, light_ns, dout, dwhere);
        printf("%-46s heavy %8.2f ms (%.3f per kernel per stream)   light chain %8.2f ms (%.1f us per link)\n", "1 heavy (512 wgs) + light 48 KB LDS", r.heavy_ms, r.heavy_ms / HN, r.light_ms, 1e3 * r.light_ms / LN);
        r = run(hs, lp, HN, LN, 512, heavy_lds, iters * 3 / 2, light_wgs, 48 * 1024, light_ns, dout, dwhere);
        printf("%-46s heavy %8.2f ms (%.3f per kernel per stream)   light chain %8.2f ms (%.1f us per link)\n", "2 heavy (512 wgs) + light 48 KB LDS", r.heavy_ms, r.heavy_ms / HN, r.light_ms, 1e3 * r.light_ms / LN);
        r = run(h1, lp, HN, LN, 256, heavy_lds, iters * 3, light_wgs, light_lds, light_ns, dout, dwhere);
        printf("%-46s heavy %8.2f ms (%.3f per kernel per stream)   light chain %8.2f ms (%.1f us per link)\n", "1 heavy (256 wgs) + light 100 KB LDS", r.heavy_ms, r.heavy_ms / HN, r.light_ms, 1e3 * r.light_ms / LN);
        r = run(h1, lp, HN, LN, 768, heavy_lds, iters, 196, 32 * 1024, 15000, dout, dwhere);
        printf("%-46s heavy %8.2f ms (%.3f per kernel per stream)   light chain %8.2f ms (%.1f us per link)\n", "1 heavy (768 wgs) + light 196 wgs 32 KB 15us", r.heavy_ms, r.heavy_ms / HN, r.light_ms, 1e3 * r.light_ms / LN);
        r = run(h1, lp, HN, LN, 512, heavy_lds, iters * 3 / 2, 196, 32 * 1024, 15000, dout, dwhere);
        printf("%-46s heavy %8.2f ms (%.3f per kernel per stream)   light chain %8.2f ms (%.1f us per link)\n", "1 heavy (512 wgs) + light 196 wgs 32 KB 15us", r.heavy_ms, r.heavy_ms / HN, r.light_ms, 1e3 * r.light_ms / LN);
    }
    if (argc > 1 && !nomask) {
    { std::vector<uint32_t> m(words, 0); m[0] = 0xffff; show_mask("bits 0-15", m); }
    { std::vector<uint32_t> m(words, 0); m[0] = 0x000000ff; show_mask("bits 0-7", m); }
    { std::vector<uint32_t> m(words, 0); m[0] = 0x01010101; m[1] = 0x01010101; show_mask("bits 0,8,16,..56", m); }
    { std::vector<uint32_t> m(words, 0); m[words - 1] = 0xffff0000; show_mask("last 16 bits", m); }
    { std::vector<uint32_t> m(words, 0xffffffff); show_mask("all", m); }

    }
    return 0;
}
