// LD_PRELOAD shim (diagnostic tool, not product code): every hipMalloc becomes its own virtual-memory mapping whose END is flush with
// the end of the buffer, followed by reserved-but-unmapped address space.  A kernel that reads or writes past the end of ANY device
// buffer (torch tensors with PYTORCH_NO_HIP_MEMORY_CACHING=1, the engine's arena slabs, its small tables) then raises a GPU memory
// fault at the first such access instead of once in nine sessions ("the round-3 core dump").
//
//   g++ -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/guard/guard_malloc.cpp -o tools/bin/libguard_malloc.so -ldl
//   LD_PRELOAD=tools/bin/libguard_malloc.so PYTORCH_NO_HIP_MEMORY_CACHING=1 python -m pytest tests/... -m gpu
//
// GUARD_ALIGN  (default 16): alignment of the returned pointer; over-reads shorter than the padding this leaves are not seen.
// GUARD_MODE   end (default) | start: which side of the buffer touches unmapped space.
// GUARD_LOG    file that receives one line per allocation / free ("A ptr size" / "F ptr"), so that the address of a fault report
//              ("Memory access fault by GPU ... on address 0x...") can be attributed: tools/guard/whose_address.py.
// GUARD_MIN    only allocations of at least this many bytes are guarded (default 0: all).
// The HIP runtime is resolved at run time from the copy the process has already loaded (torch bundles its own): nothing is linked.
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace {

struct Block { void* base; size_t reserved, mapped, map_off; hipMemGenericAllocationHandle_t handle; size_t size; };

std::mutex g_mu;
std::unordered_map<void*, Block> g_blocks;
void* g_rt = nullptr;
FILE* g_log = nullptr;
size_t g_gran = 0, g_align = 16, g_min = 0;
bool g_start_mode = false;
unsigned long long g_count = 0;

template <class F> F sym(const char* name) {
    if (!g_rt) {
        g_rt = dlopen("libamdhip64.so.7", RTLD_NOLOAD | RTLD_LAZY);
        if (!g_rt) g_rt = dlopen("libamdhip64.so", RTLD_NOLOAD | RTLD_LAZY);
        if (!g_rt) g_rt = dlopen("libamdhip64.so.7", RTLD_LAZY);
        if (!g_rt) { fprintf(stderr, "guard_malloc: no HIP runtime: %s\n", dlerror()); abort(); }
    }
    void* p = dlsym(g_rt, name);
    if (!p) { fprintf(stderr, "guard_malloc: %s not found\n", name); abort(); }
    return (F)p;
}

void init_once() {
    static bool done = false;
    if (done) return;
    done = true;
    if (const char* e = getenv("GUARD_ALIGN")) g_align = (size_t)atoll(e);
    if (const char* e = getenv("GUARD_MIN")) g_min = (size_t)atoll(e);
    if (const char* e = getenv("GUARD_MODE")) g_start_mode = !strcmp(e, "start");
    if (const char* e = getenv("GUARD_LOG")) g_log = fopen(e, "w");
    fprintf(stderr, "guard_malloc: active (align %zu, mode %s)\n", g_align, g_start_mode ? "start" : "end");
}

hipError_t guarded(void** out, size_t size) {
    int dev = 0;
    hipError_t e = sym<hipError_t (*)(int*)>("hipGetDevice")(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    if (!g_gran) {
        e = sym<hipError_t (*)(size_t*, const hipMemAllocationProp*, hipMemAllocationGranularity_flags)>("hipMemGetAllocationGranularity")(
            &g_gran, &prop, hipMemAllocationGranularityMinimum);
        if (e != hipSuccess || !g_gran) return e != hipSuccess ? e : hipErrorUnknown;
        fprintf(stderr, "guard_malloc: granularity %zu\n", g_gran);
    }
    Block b;
    b.size = size;
    const size_t padded = (size + g_align - 1) / g_align * g_align;
    b.mapped = (padded + g_gran - 1) / g_gran * g_gran;
    b.map_off = g_gran;                                   // one unmapped granule on each side
    b.reserved = b.mapped + 2 * g_gran;
    e = sym<hipError_t (*)(void**, size_t, size_t, void*, unsigned long long)>("hipMemAddressReserve")(&b.base, b.reserved, 0, nullptr, 0);
    if (e != hipSuccess) return e;
    e = sym<hipError_t (*)(hipMemGenericAllocationHandle_t*, size_t, const hipMemAllocationProp*, unsigned long long)>("hipMemCreate")(
        &b.handle, b.mapped, &prop, 0);
    if (e != hipSuccess) { sym<hipError_t (*)(void*, size_t)>("hipMemAddressFree")(b.base, b.reserved); return e; }
    char* at = (char*)b.base + b.map_off;
    e = sym<hipError_t (*)(void*, size_t, size_t, hipMemGenericAllocationHandle_t, unsigned long long)>("hipMemMap")(at, b.mapped, 0, b.handle, 0);
    if (e == hipSuccess) {
        hipMemAccessDesc acc;
        memset(&acc, 0, sizeof(acc));
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = dev;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = sym<hipError_t (*)(void*, size_t, const hipMemAccessDesc*, size_t)>("hipMemSetAccess")(at, b.mapped, &acc, 1);
        if (e != hipSuccess) sym<hipError_t (*)(void*, size_t)>("hipMemUnmap")(at, b.mapped);
    }
    if (e != hipSuccess) {
        sym<hipError_t (*)(hipMemGenericAllocationHandle_t)>("hipMemRelease")(b.handle);
        sym<hipError_t (*)(void*, size_t)>("hipMemAddressFree")(b.base, b.reserved);
        return e;
    }
    void* user = g_start_mode ? (void*)at : (void*)(at + b.mapped - padded);
    g_blocks[user] = b;
    ++g_count;
    if (g_log) { fprintf(g_log, "A %p %zu\n", user, size); fflush(g_log); }
    *out = user;
    return hipSuccess;
}

}  // namespace

extern "C" hipError_t hipMalloc(void** ptr, size_t size) {
    std::lock_guard<std::mutex> lk(g_mu);
    init_once();
    if (!ptr) return hipErrorInvalidValue;
    if (size == 0) { *ptr = nullptr; return hipSuccess; }
    if (size < g_min) return sym<hipError_t (*)(void**, size_t)>("hipMalloc")(ptr, size);
    return guarded(ptr, size);
}

extern "C" hipError_t hipFree(void* ptr) {
    std::unique_lock<std::mutex> lk(g_mu);
    init_once();
    auto it = g_blocks.find(ptr);
    if (it == g_blocks.end()) { lk.unlock(); return sym<hipError_t (*)(void*)>("hipFree")(ptr); }
    Block b = it->second;
    g_blocks.erase(it);
    if (g_log) { fprintf(g_log, "F %p\n", ptr); fflush(g_log); }
    lk.unlock();
    // hipFree waits for the device (kernels of any stream may still use the block)
    hipError_t e = sym<hipError_t (*)()>("hipDeviceSynchronize")();
    char* at = (char*)b.base + b.map_off;
    sym<hipError_t (*)(void*, size_t)>("hipMemUnmap")(at, b.mapped);
    sym<hipError_t (*)(hipMemGenericAllocationHandle_t)>("hipMemRelease")(b.handle);
    // (the address range is NOT given back: hipMemAddressFree segfaults inside the runtime when called from a torch process -- seen on
    // ROCm 7.0's bundled libamdhip64 -- and a never-reused range also turns every use-after-free into a fault; 2^47 bytes last a while)
    if (getenv("GUARD_ADDRESS_FREE")) sym<hipError_t (*)(void*, size_t)>("hipMemAddressFree")(b.base, b.reserved);
    return e;
}
