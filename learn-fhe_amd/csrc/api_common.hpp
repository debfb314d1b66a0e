// Shared helpers of the extern "C" translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

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

// Per-call device workspace, allocated and released IN STREAM ORDER (hipMallocAsync / hipFreeAsync on the device's default
// pool, whose release threshold is raised once so freed blocks are kept for the next call): an entry point neither pays a
// device-wide hipMalloc / hipFree nor has to synchronise the stream before it returns to keep its workspace alive.
struct StreamWs {
    void *p = nullptr;
    hipStream_t st;
    int rc = FHE_OK;
    StreamWs(size_t bytes, hipStream_t s) : st(s) {
        static thread_local int pool_ready_for = -1;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && pool_ready_for != dev) {
            hipMemPool_t pool;
            if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
                uint64_t keep = ~uint64_t(0);
                (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            }
            pool_ready_for = dev;
        }
        const hipError_t e = hipMallocAsync(&p, bytes ? bytes : 8, s);
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
    Mirror(const void *p, size_t count, fhe_mem mem, bool copy_in, hipStream_t st) {
        bytes = count * sizeof(u64);
        if (mem == FHE_MEM_DEVICE || count == 0) { d = (u64 *)p; return; }
        host = const_cast<void *>(p);
        owned = true;
        hipError_t e = hipMalloc((void **)&d, bytes);
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
        if (owned && d) (void)hipFree(d);
    }
    Mirror(const Mirror &) = delete;
    Mirror &operator=(const Mirror &) = delete;
};
}  // namespace
