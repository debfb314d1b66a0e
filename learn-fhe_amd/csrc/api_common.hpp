// Shared helpers of the extern "C" translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
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
inline hipMemPool_t private_pool(int dev) {
    static std::mutex mu;
    if (dev < 0 || dev >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
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
