// Shared helpers of the extern "C" translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/fhe_ring.h"
#include "dev_arith.hpp"

namespace fhe {
inline thread_local int g_last_hip = 0;
}
using fhe::g_last_hip;
using fhe::u64;

#define HIP_TRY(expr)                         \
    do {                                      \
        hipError_t e_ = (expr);               \
        if (e_ != hipSuccess) {               \
            fhe::g_last_hip = (int)e_;        \
            return FHE_ERR_HIP;               \
        }                                     \
    } while (0)

namespace {
inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
inline int ilog2(size_t n) { return 63 - __builtin_clzll((unsigned long long)n); }

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) prev = -1;
        if (prev != dev) e = hipSetDevice(dev);
        if (e != hipSuccess) { ok = false; fhe::g_last_hip = (int)e; }
        if (prev == dev) prev = -1;  // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// Entry points that take a modulus instead of a context have no device of their own: they run where their DEVICE operands
// live (hipPointerGetAttributes of the first one), whatever the caller's current device is; host-memory calls use the current
// device.
struct PtrDeviceGuard {
    DeviceGuard *g = nullptr;
    bool ok = true;
    PtrDeviceGuard(const void *p, fhe_mem mem) {
        if (mem != FHE_MEM_DEVICE || !p) return;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return; }  // not a HIP allocation: leave the device alone
        g = new DeviceGuard(at.device);
        ok = g->ok;
    }
    ~PtrDeviceGuard() { delete g; }
    PtrDeviceGuard(const PtrDeviceGuard &) = delete;
    PtrDeviceGuard &operator=(const PtrDeviceGuard &) = delete;
};
}  // namespace

// Per-call device workspace, allocated and released IN STREAM ORDER from a pool of the LIBRARY'S OWN (one per device, created on
// first use): an entry point neither pays a device-wide hipMalloc / hipFree nor has to synchronise the stream before it returns
// to keep its workspace alive, and the device's default pool -- which a co-resident framework may be using -- keeps its
// settings.  The pool holds on to freed blocks up to FHE_POOL_KEEP_BYTES; fhe_trim() hands everything back to the driver.
namespace fhe {
constexpr uint64_t POOL_KEEP_BYTES = uint64_t(4) << 30;
constexpr int MAX_DEVICES = 64;
inline hipMemPool_t g_pools[MAX_DEVICES] = {};
inline std::mutex &pool_mutex() {
    static std::mutex mu;
    return mu;
}
inline hipMemPool_t private_pool(int dev) {
    if (dev < 0 || dev >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(pool_mutex());
    if (!g_pools[dev]) {
        hipMemPoolProps props = {};
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t pool = nullptr;
        if (hipMemPoolCreate(&pool, &props) != hipSuccess) return nullptr;
        uint64_t keep = POOL_KEEP_BYTES;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        g_pools[dev] = pool;
    }
    return g_pools[dev];
}
}  // namespace fhe

// Lab switches: routes that a faster one replaced stay in the library so that tests can compare the two bit for bit and A/B runs
// need no second build.  None changes a result.  Each is read from the environment ONCE (FHE_RING_<NAME>, when the library first
// looks at any of them) and afterwards only changes through fhe_set_option(): no getenv on any call path.
namespace fhe {
enum Opt { OPT_NO_EDGE = 0, OPT_NO_LIMB_MAJOR, OPT_NO_W12, OPT_NO_FUSED_MUL, OPT_SMALL_BATCH, OPT_NO_PACKED_DIGITS, OPT_NO_F64_EXACT, OPT_COUNT };
inline const char *const OPT_NAMES[OPT_COUNT] = {"NO_EDGE", "NO_LIMB_MAJOR", "NO_W12", "NO_FUSED_MUL", "SMALL_BATCH", "NO_PACKED_DIGITS", "NO_F64_EXACT"};
struct Options {
    std::atomic<long> v[OPT_COUNT];
    Options() {
        for (int i = 0; i < OPT_COUNT; ++i) {
            char name[64] = "FHE_RING_";
            std::strncat(name, OPT_NAMES[i], sizeof(name) - 10);
            const char *e = std::getenv(name);
            v[i].store(e ? std::atol(e) : (i == OPT_SMALL_BATCH ? -1L : 0L), std::memory_order_relaxed);
        }
    }
};
inline Options &options() {
    static Options o;
    return o;
}
inline long opt(Opt k) { return options().v[k].load(std::memory_order_relaxed); }

// compute units of a device (cached): dispatch thresholds are stated in workgroup generations of THIS chip, not in literals
inline int cu_count(int dev) {
    static std::atomic<int> cache[MAX_DEVICES] = {};
    if (dev < 0 || dev >= MAX_DEVICES) return 256;
    int c = cache[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) { (void)hipGetLastError(); c = 256; }
    cache[dev].store(c, std::memory_order_relaxed);
    return c;
}
inline int current_cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    return cu_count(dev);
}
}  // namespace fhe

namespace {
struct StreamWs {
    void *p = nullptr;
    hipStream_t st;
    int rc = FHE_OK;
    StreamWs(size_t bytes, hipStream_t s) : st(s) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        hipMemPool_t pool = e == hipSuccess ? fhe::private_pool(dev) : nullptr;
        if (e == hipSuccess) e = pool ? hipMallocFromPoolAsync(&p, bytes ? bytes : 8, pool, s) : hipMallocAsync(&p, bytes ? bytes : 8, s);
        if (e != hipSuccess) { fhe::g_last_hip = (int)e; rc = FHE_ERR_HIP; p = nullptr; }
    }
    ~StreamWs() {
        if (p) (void)hipFreeAsync(p, st);
    }
    StreamWs(const StreamWs &) = delete;
    StreamWs &operator=(const StreamWs &) = delete;
    template <class T>
    T *as() const { return static_cast<T *>(p); }
};

// Host-memory convenience path (FHE_MEM_HOST): a device mirror of a host array.  `in`: copy host -> device on
// construction; sync_out(): copy back.  Device-memory callers (FHE_MEM_DEVICE) get the pointer passed through.
struct Mirror {
    u64 *d = nullptr;
    void *host = nullptr;
    size_t bytes = 0;
    bool owned = false;
    int rc = FHE_OK;
    hipStream_t st_ = nullptr;
    Mirror(const void *p, size_t count, fhe_mem mem, bool copy_in, hipStream_t st) : st_(st) {
        bytes = count * sizeof(u64);
        if (mem == FHE_MEM_DEVICE || count == 0) { d = (u64 *)p; return; }
        host = const_cast<void *>(p);
        owned = true;
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        hipMemPool_t pool = e == hipSuccess ? fhe::private_pool(dev) : nullptr;
        if (e == hipSuccess) e = pool ? hipMallocFromPoolAsync((void **)&d, bytes, pool, st) : hipMallocAsync((void **)&d, bytes, st);
        if (e == hipSuccess && copy_in) e = hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { fhe::g_last_hip = (int)e; rc = FHE_ERR_HIP; }
    }
    int sync_out(hipStream_t st) {
        if (!owned || rc != FHE_OK) return rc;
        hipError_t e = hipMemcpyAsync(host, d, bytes, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { fhe::g_last_hip = (int)e; rc = FHE_ERR_HIP; }
        return rc;
    }
    ~Mirror() {
        if (owned && d) (void)hipFreeAsync(d, st_);  // stream ordered: after the kernels and copies that use it
    }
    Mirror(const Mirror &) = delete;
    Mirror &operator=(const Mirror &) = delete;
};
}  // namespace
