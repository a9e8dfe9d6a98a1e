// State arenas by the library (include/ipp_engine.h, "Arena placement"): where a batch's arena lies in PHYSICAL memory decides
// whether large batches run in their fast or their slow mode (profiles/r06_arena_modes.txt), and neither a framework's caching
// allocator nor hipMalloc gives a handle on that.  Here: plain hipMalloc outside any caching allocator, and the virtual-memory
// API (a 1-GiB-aligned reservation backed by physical chunks of a chosen size), plus the bare row-stream probe that tells a fast
// placement from a slow one in a millisecond without an engine on it.  A second translation unit: none of this touches the
// engine's kernels, and the engine takes any device pointer as its arena (ipp_engine_create).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/ipp_engine.h"

namespace ipp { int set_error(int code, const char* msg); }

namespace {

int failf(int code, const char* what, hipError_t e) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s failed: %s", what, hipGetErrorString(e));
    return ipp::set_error(code, buf);
}
#define ARENA_TRY(expr)                                   \
    do {                                                  \
        hipError_t e_ = (expr);                           \
        if (e_ != hipSuccess) return failf(-2, #expr, e_); \
    } while (0)

struct Mapping {
    void* reserved = nullptr;      // VMM: start and size of the address reservation (the arena is an aligned range inside it)
    uint64_t reserved_bytes = 0;
    int kind;
    int device = 0;
    uint64_t bytes;      // reserved = mapped size (VMM), requested size (hipMalloc)
    std::vector<hipMemGenericAllocationHandle_t> chunks;
    std::vector<uint64_t> sizes;
    uint64_t chunk_bytes;
};
std::mutex g_mu;
std::map<void*, Mapping> g_arenas;
// Physical chunks of freed VMM arenas, kept for the next arena of this process (per device and chunk size) instead of going back to the
// driver: a chunk that has served a fast arena serves the next one as well, whereas memory that the driver has just taken back comes out
// again in pieces for a while -- the second large workload of a process ran 5-25 % slower than the same workload in a fresh process
// (profiles/r06_experiments.txt 16).  Bounded by IPP_ARENA_POOL_GIB (default 48, 0: off); ipp_arena_trim hands the memory back.
struct PoolKey { int device; uint64_t size; bool operator<(const PoolKey& o) const { return device != o.device ? device < o.device : size < o.size; } };
std::map<PoolKey, std::vector<hipMemGenericAllocationHandle_t>> g_pool;
uint64_t g_pool_bytes = 0;
uint64_t pool_cap_bytes() {
    static const uint64_t cap = [] {
        const char* e = getenv("IPP_ARENA_POOL_GIB");
        const long long gib = e ? atoll(e) : 48;
        return (uint64_t)(gib > 0 ? gib : 0) << 30;
    }();
    return cap;
}
uint64_t g_retired_bytes = 0;  // address ranges of freed VMM arenas (kept reserved, see ipp_arena_free)

// The row stream of k_step_patch without its arithmetic: item i owns the slot [i, i + 1) x slot_floats of the arena; a wave reads,
// for `rows` stored columns (656-float patches at pseudo-random places of the slot), 512 consecutive bytes per unit, sixteen
// requests in flight, 8 bytes per lane through one buffer resource per column -- the request shape of csrc/k_patch_units.h at
// the kernel's occupancy (two-wave workgroups, 20 KB of LDS each).
constexpr int kPatchFloats = 656, kUnits = 6, kGroup = 16;

__global__ __launch_bounds__(128, 4) void k_arena_probe(const float* __restrict__ base, uint64_t slot_floats, int n_items, int rows,
                                                         int cols, float* out) {
    extern __shared__ float lds[];
    const int item = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (item >= n_items) return;
    const float* env = base + (size_t)item * slot_floats;
    float acc0 = 0.f, acc1 = 0.f;
    for (int u = wave; u < kUnits; u += 2) {
        unsigned off = (unsigned)(u * 128 + 2 * lane) * 4u;
        if (off + 8 > kPatchFloats * 4) off = 0xffffffffu;
        for (int k0 = 0; k0 < rows; k0 += kGroup) {
            float2 v[kGroup];
#pragma unroll
            for (int i = 0; i < kGroup; ++i) {
                const int k = (int)(((unsigned)(k0 + i) * 7u + (unsigned)item * 13u) % (unsigned)cols);
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(env + (size_t)k * kPatchFloats), 0, kPatchFloats * 4, 0x00020000);
                v[i] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, (k0 + i < rows) ? off : 0xffffffffu, 0, 2));
            }
#pragma unroll
            for (int i = 0; i < kGroup; ++i) { acc0 += v[i].x; acc1 += v[i].y; }
        }
    }
    if (acc0 + acc1 == 123.456f) out[0] = acc0 + lds[0];
}

// The other half of a placement: LATENCY.  Every wave follows a chain of `hops` dependent 512-byte requests to pseudo-random patches
// anywhere in the arena (the next address needs the previous value: `zero` is 0 at run time, the compiler cannot know) -- what the
// prologue of a step item is made of (header -> rectangles -> gather -> rows), and what address translation misses lengthen
// while a bandwidth probe with sixteen requests in flight does not notice them.
__global__ __launch_bounds__(64) void k_arena_latency(const float* __restrict__ base, uint64_t n_patches, int hops, unsigned zero, float* out) {
    const int lane = threadIdx.x;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (blockIdx.x + 1);
    float acc = 0.f;
    for (int h = 0; h < hops; ++h) {
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        const uint64_t patch = x % n_patches;
        const float2 v = *reinterpret_cast<const float2*>(base + patch * kPatchFloats + 2 * lane);
        acc += v.x + v.y;
        x += (unsigned long long)(__float_as_uint(v.x) & zero) + 1ull;
    }
    if (acc == 123.456f) out[0] = acc;
}

// scratch of a probe call, released on every return path
struct ProbeScratch {
    float* sink = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    ~ProbeScratch() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
        if (sink) (void)hipFree(sink);
    }
};

}  // namespace

extern "C" {

int ipp_arena_alloc(int device, uint64_t bytes, int32_t kind, uint64_t chunk_bytes, uint64_t align_bytes, void** arena) {
    if (!arena || bytes == 0) return ipp::set_error(-1, "ipp_arena_alloc: null / empty request");
    *arena = nullptr;
    ARENA_TRY(hipSetDevice(device));
    Mapping m;
    m.kind = kind; m.device = device; m.bytes = bytes; m.chunk_bytes = 0;
    void* p = nullptr;
    if (kind == IPP_ARENA_HIPMALLOC) {
        ARENA_TRY(hipMalloc(&p, bytes));
    } else if (kind == IPP_ARENA_VMM) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t gran = 0;
        ARENA_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        if (gran == 0) gran = 2u << 20;
        uint64_t chunk = chunk_bytes ? chunk_bytes : (uint64_t)1 << 30;
        chunk = (chunk + gran - 1) / gran * gran;
        // whole chunks, then ONE tail chunk of the remainder rounded up to the granularity (a 4.1-GiB arena is 4 chunks of 1 GiB
        // and 0.1 GiB, not 5 GiB)
        const uint64_t n_full = bytes / chunk;
        const uint64_t tail = (bytes - n_full * chunk + gran - 1) / gran * gran;
        const uint64_t total = n_full * chunk + tail;
        // the reservation's own alignment argument is not honoured beyond the granularity on this runtime (a 1-GiB request came
        // back 160 MiB off): reserve `align` more and map into the aligned range inside
        uint64_t align = align_bytes ? align_bytes : chunk;
        if (align & (align - 1)) return ipp::set_error(-1, "ipp_arena_alloc: align_bytes must be a power of two");
        if (align < gran) align = gran;
        void* r = nullptr;
        ARENA_TRY(hipMemAddressReserve(&r, total + align, gran, nullptr, 0));
        m.reserved = r;
        m.reserved_bytes = total + align;
        p = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(r) + align - 1) / align * align);
        m.bytes = total;
        m.chunk_bytes = chunk;
        auto undo = [&]() {
            uint64_t o = 0;
            for (size_t i = 0; i < m.chunks.size(); ++i) {
                (void)hipMemUnmap((char*)p + o, m.sizes[i]);
                (void)hipMemRelease(m.chunks[i]);
                o += m.sizes[i];
            }
            (void)hipMemAddressFree(m.reserved, m.reserved_bytes);
        };
        for (uint64_t off = 0; off < total;) {
            const uint64_t sz = (total - off >= chunk) ? chunk : total - off;
            hipMemGenericAllocationHandle_t h;
            bool pooled = false;
            {
                std::lock_guard<std::mutex> g(g_mu);
                auto it = g_pool.find(PoolKey{device, sz});
                if (it != g_pool.end() && !it->second.empty()) {
                    h = it->second.back();
                    it->second.pop_back();
                    g_pool_bytes -= sz;
                    pooled = true;
                }
            }
            hipError_t e = pooled ? hipSuccess : hipMemCreate(&h, sz, &prop, 0);
            if (e == hipSuccess) {
                e = hipMemMap((char*)p + off, sz, 0, h, 0);
                if (e != hipSuccess) (void)hipMemRelease(h);
            }
            if (e != hipSuccess) {
                undo();
                return failf(-2, "hipMemCreate / hipMemMap", e);
            }
            m.chunks.push_back(h);
            m.sizes.push_back(sz);
            off += sz;
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        hipError_t e = hipMemSetAccess(p, total, &acc, 1);
        if (e != hipSuccess) {
            undo();
            return failf(-2, "hipMemSetAccess", e);
        }
    } else {
        return ipp::set_error(-1, "ipp_arena_alloc: unknown kind");
    }
    {
        std::lock_guard<std::mutex> g(g_mu);
        g_arenas[p] = std::move(m);
    }
    *arena = p;
    return 0;
}

int ipp_arena_free(void* arena) {
    if (!arena) return 0;
    Mapping m;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto it = g_arenas.find(arena);
        if (it == g_arenas.end()) return ipp::set_error(-1, "ipp_arena_free: not an arena of ipp_arena_alloc");
        m = std::move(it->second);
        g_arenas.erase(it);
    }
    if (m.kind == IPP_ARENA_HIPMALLOC) {
        ARENA_TRY(hipFree(arena));
        return 0;
    }
    uint64_t off = 0;
    for (size_t i = 0; i < m.chunks.size(); ++i) {
        ARENA_TRY(hipMemUnmap((char*)arena + off, m.sizes[i]));
        bool keep = false;
        if (m.sizes[i] == m.chunk_bytes) {  // (whole chunks only: the tail chunk goes back)
            std::lock_guard<std::mutex> g(g_mu);
            if (g_pool_bytes + m.sizes[i] <= pool_cap_bytes()) {
                g_pool[PoolKey{m.device, m.sizes[i]}].push_back(m.chunks[i]);
                g_pool_bytes += m.sizes[i];
                keep = true;
            }
        }
        if (!keep) ARENA_TRY(hipMemRelease(m.chunks[i]));
        off += m.sizes[i];
    }
    // The ADDRESS range stays reserved for the life of the process.  Measured on this runtime (ROCm 7.2, gfx950; tools/arena_modes.py,
    // profiles/r06_arena_modes.txt): an arena mapped into a range that an earlier, freed reservation had covered faults on first
    // touch now and then ("Memory access fault ... Reason: Unknown" at a freshly mapped address, one run in three of a test sequence
    // that frees and maps arenas of changing sizes) -- a fresh reservation never overlaps a retired one, and 47 bits of address
    // space hold thousands of them.  The PHYSICAL memory went back above.
    {
        std::lock_guard<std::mutex> g(g_mu);
        g_retired_bytes += m.reserved_bytes;
    }
    return 0;
}

int ipp_arena_trim(int device, uint64_t* released) {
    uint64_t n = 0;
    std::vector<std::pair<uint64_t, hipMemGenericAllocationHandle_t>> out;
    {
        std::lock_guard<std::mutex> g(g_mu);
        for (auto& kv : g_pool)
            if (device < 0 || kv.first.device == device) {
                for (auto h : kv.second) out.emplace_back(kv.first.size, h);
                kv.second.clear();
            }
        for (auto& x : out) g_pool_bytes -= x.first;
    }
    for (auto& x : out) {
        ARENA_TRY(hipMemRelease(x.second));
        n += x.first;
    }
    if (released) *released = n;
    return 0;
}

int ipp_arena_retired_bytes(uint64_t* bytes) {
    if (!bytes) return ipp::set_error(-1, "null argument");
    std::lock_guard<std::mutex> g(g_mu);
    *bytes = g_retired_bytes;
    return 0;
}

int ipp_arena_probe(int device, const void* arena, uint64_t bytes, int32_t items, int32_t rows, int32_t launches, void* stream,
                    double* ms) {
    if (!arena || !ms || items < 1 || rows < 1 || launches < 1 || launches > 1000)
        return ipp::set_error(-1, "ipp_arena_probe: bad argument");
    const uint64_t slot_floats = bytes / 4 / (uint64_t)items;
    const uint64_t cols = slot_floats / kPatchFloats;
    if (cols < 1) return ipp::set_error(-1, "ipp_arena_probe: arena smaller than one patch per item");
    ARENA_TRY(hipSetDevice(device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProbeScratch ps;  // (per call: a cached sink would belong to ONE device)
    ARENA_TRY(hipMalloc(&ps.sink, 64));
    ARENA_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_arena_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 20480));
    ARENA_TRY(hipEventCreate(&ps.a));
    ARENA_TRY(hipEventCreate(&ps.b));
    float* sink = ps.sink;
    hipEvent_t a = ps.a, b = ps.b;
    const int c = (int)(cols > 0x7fffffff ? 0x7fffffff : cols);
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL(k_arena_probe, dim3(items), dim3(128), 20480, s, (const float*)arena, slot_floats, items, rows, c, sink);
    ARENA_TRY(hipEventRecord(a, s));
    for (int i = 0; i < launches; ++i)
        hipLaunchKernelGGL(k_arena_probe, dim3(items), dim3(128), 20480, s, (const float*)arena, slot_floats, items, rows, c, sink);
    ARENA_TRY(hipEventRecord(b, s));
    ARENA_TRY(hipEventSynchronize(b));
    float t = 0.f;
    ARENA_TRY(hipEventElapsedTime(&t, a, b));
    *ms = (double)t / launches;
    ARENA_TRY(hipGetLastError());
    return 0;
}

int ipp_arena_latency(int device, const void* arena, uint64_t bytes, int32_t waves, int32_t hops, void* stream, double* ns_per_hop) {
    if (!arena || !ns_per_hop || waves < 1 || hops < 1) return ipp::set_error(-1, "ipp_arena_latency: bad argument");
    const uint64_t n_patches = bytes / 4 / kPatchFloats;
    if (n_patches < 1) return ipp::set_error(-1, "ipp_arena_latency: arena smaller than one patch");
    ARENA_TRY(hipSetDevice(device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProbeScratch ps;
    ARENA_TRY(hipMalloc(&ps.sink, 64));
    ARENA_TRY(hipEventCreate(&ps.a));
    ARENA_TRY(hipEventCreate(&ps.b));
    float* sink = ps.sink;
    hipEvent_t a = ps.a, b = ps.b;
    hipLaunchKernelGGL(k_arena_latency, dim3(waves), dim3(64), 0, s, (const float*)arena, n_patches, hops, 0u, sink);
    ARENA_TRY(hipEventRecord(a, s));
    hipLaunchKernelGGL(k_arena_latency, dim3(waves), dim3(64), 0, s, (const float*)arena, n_patches, hops, 0u, sink);
    ARENA_TRY(hipEventRecord(b, s));
    ARENA_TRY(hipEventSynchronize(b));
    float t = 0.f;
    ARENA_TRY(hipEventElapsedTime(&t, a, b));
    *ns_per_hop = 1e6 * (double)t / hops;
    ARENA_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
