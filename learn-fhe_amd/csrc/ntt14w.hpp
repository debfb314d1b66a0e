// N = 2^14 and N = 2^15 transform kernels, wave-local form (BASELINE configs 2 and 4; the 2^14 form also as the sub-transforms of
// 2^16..2^17 rings).  Written out below for 2^14 (R0 = 3); R0 = 4 adds one layer to pass 0 and doubles the waves.
//
// A polynomial lives in the registers of a 512-thread workgroup (32 coefficients per thread, two workgroups per CU) as in
// the first register-resident kernel, but the index bits are dealt so that after the first three layers every WAVE owns one
// independent sub-transform of 2^11 coefficients (the classic four-step split: layers 0..2 couple the eight blocks of 2^11,
// layers 3..13 stay inside a block).  Only ONE exchange crosses waves (workgroup barriers); the other two are wave-private
// (the wave's own LDS region, ordered by its own lgkmcnt), so the eight waves of a workgroup drift apart and one wave's
// exchange latency hides behind another wave's butterflies.  4 barriers per transform instead of 12.
//
//   wave w0 = i[8:6] , lane = i[5:0]                                            (i = coefficient index, 14 bits)
//   pass 0: regs {13,12,11 | 10,9}   layers 0..2    HBM side: 8-byte coalesced accesses; twiddles wave-uniform (SGPRs)
//   ---- X01: across waves, two rounds split by bit 10, lanes keep i[5:0] ------------------------------------------------
//   wave w = i[13:11] from here on
//   pass 1: regs {10,9,8,7 | 6}      lane = i[5:0]                  layers 3..6     twiddles wave-uniform (SGPRs)
//   ---- X12: wave-private, split by bit 6 -----------------------------------------------------------------------------
//   pass 2: regs {6,5,4,3 | 2}       lane = (i[10:7], i[1:0])       layers 7..10    15 twiddles per thread
//   ---- X23: wave-private, split by bit 2 -----------------------------------------------------------------------------
//   pass 3: regs {2,1,0 | 10,9}      lane = i[8:3]                  layers 11..13   28 twiddles per thread
//                                                                   HBM side: 64 contiguous bytes per lane and (10,9) value
//
// The passive register bits (after the bar) sit BELOW the bits a pass processes in passes 0..2, so its replicas share their
// twiddles.  Forward = passes 0 -> 3 (Cooley-Tukey, util/src/ring/fft.rs:40-54), inverse = 3 -> 0 (Gentleman-Sande, 59-77, n^-1
// folded into the last layer).  Natural-order coefficients <-> bit-reversed evaluations, twiddle tw[2^layer + block], exactly
// as the reference.  LDS layouts are conflict free for ds_write_b64 / ds_read_b64 (tools/lds_bank_sim.py).
#pragma once
#include <type_traits>
#include "ntt_kernels.hpp"

namespace fhe {
namespace w14 {

// R0 = layers of pass 0 = log2 of the waves per workgroup: 1 / 2 -> N = 2^12 / 2^13 (128 / 256 threads, eight / four workgroups per
// CU), 3 -> N = 2^14 (512 threads, two workgroups per CU), 4 -> N = 2^15
// (1024 threads, one workgroup per CU, 136 KiB of LDS: a 2^15 ring in ONE pass over HBM instead of a radix-2 pass + two 2^14
// sub-transforms).  Everything after X01 is the same code: a wave and its 2^11 block.
constexpr int WSLOTS = 1088;                         // wave-private region: 1024 coefficients + padding, in u64 slots
template <int R0> constexpr int threads() { return 64 << R0; }
template <int R0> constexpr size_t lds_bytes() { return (size_t(1) << R0) * WSLOTS * 8; }  // 69 632 B / 139 264 B; X01 uses the first 2^R0 x 8 KiB
constexpr int THREADS = threads<3>();
constexpr size_t LDS_BYTES = lds_bytes<3>();

// A wave's LDS instructions execute in issue order: inside a wave-private exchange a read issued after a write sees it, and a
// write issued after a read cannot overtake it.  All that is needed is that the COMPILER keeps the order (a workgroup-scope
// fence would also drain vmcnt, i.e. make every exchange wait for the twiddle fetches that are deliberately left in flight
// across it).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
// An exchange is a short burst of LDS instructions: latency critical, while the other waves of the SIMD are usually inside a
// butterfly pass (throughput bound).  W14_XCHG_PRIO: exchanging waves take issue priority (lab switch; see DESIGN.md for the A/B).
#ifndef W14_XCHG_PRIO
#define W14_XCHG_PRIO 0
#endif
#if W14_XCHG_PRIO
#define W14_PRIO_UP() __builtin_amdgcn_s_setprio(3)
#define W14_PRIO_DOWN() __builtin_amdgcn_s_setprio(0)
#else
#define W14_PRIO_UP()
#define W14_PRIO_DOWN()
#endif

#ifdef NTT14_STAMPS
__device__ unsigned long long g_stamps[4096][16];
#define STAMP_DECL unsigned long long stamps_[12]
#define STAMP_ENTRY() unsigned long long entry_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(entry_)::"memory")
#define STAMP_ENTRY_ARG , entry_
#define STAMP_ENTRY_PARAM , unsigned long long entry_
#define STAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamps_[i])::"memory")
#define STAMP_REAL(i) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamps_[i])::"memory")  /* 100 MHz */
#define STAMP_FLUSH()                                                                       \
    do {                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < 4096)                                          \
            for (int i_ = 0; i_ < 12; ++i_) g_stamps[blockIdx.x][i_] = stamps_[i_];        \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                        \
            unsigned hw_, xcc_;                                                             \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));               \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));             \
            g_stamps[blockIdx.x][12] = ((unsigned long long)xcc_ << 32) | hw_;              \
            g_stamps[blockIdx.x][13] = entry_;                                              \
        }                                                                                   \
    } while (0)
#else
#define STAMP_DECL
#define STAMP_ENTRY()
#define STAMP_ENTRY_ARG
#define STAMP_ENTRY_PARAM
#define STAMP(i)
#define STAMP_REAL(i)
#define STAMP_FLUSH()
#endif

// ---- register naming -----------------------------------------------------------------------------------------------
//   pass 0: x[(s2 << 3) | n3]    n3 = i[13:11], s2 = i[10:9]
//   pass 1: x[(s << 4) | n4]     n4 = i[10:7],  s = i[6]
//   pass 2: x[(s << 4) | n4]     n4 = i[6:3],   s = i[2]
//   pass 3: x[(ab << 3) | n3]    n3 = i[2:0],   ab = i[10:9]

// pass 0 -> pass 1 (across waves; round h moves the coefficients with i[10] = h).  Pass-0 register (pass, n): n = the top R0 index
// bits, pass = the 5 - R0 bits below them (i[10] first; R0 = 3: (i10, i9), R0 = 4: i10); thread t = the remaining low bits.
// Slot of a coefficient in the half image (bit 10 removed): (n << 10) | i[9:0]; lanes keep i[5:0]: conflict free unpadded.
#ifdef W14_LAB_NO_BARRIER  // developer lab only: what do the workgroup barriers of the cross-wave exchange cost? (results wrong)
#define W14_SYNC() wave_sync()
#else
#define W14_SYNC() __syncthreads()
#endif

template <int R0>
__device__ __forceinline__ void xchg_01(u64 (&x)[32], int t, int w, u64 *lds) {
    constexpr int PB = 5 - R0, NLOW = 1 << (PB - 1);  // register bits below bit 10 (i9 for R0 = 3, none for R0 = 4)
    u64 *wp = lds + t, *rp = lds + (w << 10) + (t & 63);  // constant offsets from here on: immediates of the ds instructions
    u64 y[32];
    W14_PRIO_UP();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int lo = 0; lo < NLOW; ++lo)
#pragma unroll
            for (int n = 0; n < (1 << R0); ++n) wp[(n << 10) | (lo << (10 - (PB - 1)))] = x[((((h << (PB - 1)) | lo)) << R0) | n];
        W14_SYNC();
#pragma unroll
        for (int m = 0; m < 16; ++m) {  // m = (i9 i8 i7 i6)
            const int n4 = (h << 3) | (m >> 1), s = m & 1;
            y[(s << 4) | n4] = rp[m << 6];
        }
        W14_SYNC();
    }
    W14_PRIO_DOWN();
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = y[r];
}

template <int R0>
__device__ __forceinline__ void xchg_10(u64 (&x)[32], int t, int w, u64 *lds) {
    constexpr int PB = 5 - R0, NLOW = 1 << (PB - 1);
    u64 *rp = lds + t, *wp = lds + (w << 10) + (t & 63);
    u64 y[32];
    W14_PRIO_UP();
    W14_SYNC();  // every wave has left its private region
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int n4 = (h << 3) | (m >> 1), s = m & 1;
            wp[m << 6] = x[(s << 4) | n4];
        }
        W14_SYNC();
#pragma unroll
        for (int lo = 0; lo < NLOW; ++lo)
#pragma unroll
            for (int n = 0; n < (1 << R0); ++n) y[((((h << (PB - 1)) | lo)) << R0) | n] = rp[(n << 10) | (lo << (10 - (PB - 1)))];
        if (h == 0) W14_SYNC();
    }
    W14_PRIO_DOWN();
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = y[r];
}

// pass 1 -> pass 2 (wave-private; round h moves i[6] = h).  Slot of (i[10:7] = n4, i[5:0] = l6): 68 n4 + l6.
__device__ __forceinline__ void xchg_12(u64 (&x)[32], int lane, u64 *wl) {
    u64 *wp = wl + lane;
    const u64 *rp = wl + 68 * (lane >> 2) + (lane & 3);  // as pass-2 lane (i[10:7], i[1:0]); + 4 (i[5:2])
    u64 y[32];
    W14_PRIO_UP();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int n4 = 0; n4 < 16; ++n4) wp[68 * n4] = x[(h << 4) | n4];
        wave_sync();
#pragma unroll
        for (int r4 = 0; r4 < 16; ++r4)  // r4 = i[5:2] -> pass-2 register n4 = (h, i5, i4, i3), s = i2
            y[((r4 & 1) << 4) | (h << 3) | (r4 >> 1)] = rp[4 * r4];
        wave_sync();
    }
    W14_PRIO_DOWN();
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = y[r];
}

__device__ __forceinline__ void xchg_21(u64 (&x)[32], int lane, u64 *wl) {
    u64 *wp = wl + 68 * (lane >> 2) + (lane & 3);
    const u64 *rp = wl + lane;
    u64 y[32];
    W14_PRIO_UP();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r4 = 0; r4 < 16; ++r4) wp[4 * r4] = x[((r4 & 1) << 4) | (h << 3) | (r4 >> 1)];
        wave_sync();
#pragma unroll
        for (int n4 = 0; n4 < 16; ++n4) y[(h << 4) | n4] = rp[68 * n4];
        wave_sync();
    }
    W14_PRIO_DOWN();
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = y[r];
}

// pass 2 -> pass 3 (wave-private; round h moves i[2] = h).  Slot of a coefficient: with hi5 = (i10 i9 i1 i0 i8),
// 34 hi5 + 2 i[6:3] + i7  (= (hi5 << 5 | i[6:3] << 1 | i7) + 2 hi5: two pad slots per 32).
__device__ __forceinline__ void xchg_23(u64 (&x)[32], int lane, u64 *wl) {
    // as pass-2 lane (i10 i9 i8 i7 i1 i0)
    const int hi5 = ((lane >> 4) << 3) | ((lane & 3) << 1) | ((lane >> 3) & 1);
    u64 *wp = wl + 34 * hi5 + ((lane >> 2) & 1);  // + 2 i[6:3]
    // as pass-3 lane i[8:3]: 34 i8 + 2 i[6:3] + i7;  + 68 (i10 i9 i1 i0)
    const u64 *rp = wl + 34 * (lane >> 5) + 2 * (lane & 15) + ((lane >> 4) & 1);
    u64 y[32];
    W14_PRIO_UP();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int c = 0; c < 16; ++c) wp[2 * c] = x[(h << 4) | c];
        wave_sync();
#pragma unroll
        for (int R = 0; R < 16; ++R)  // R = (i10 i9 i1 i0) -> pass-3 register ab = R >> 2, n3 = (h, i1, i0)
            y[((R >> 2) << 3) | (h << 2) | (R & 3)] = rp[68 * R];
        wave_sync();
    }
    W14_PRIO_DOWN();
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = y[r];
}

// pass 3 -> pass 2 (inverse direction).  Its own layout (each direction picks the one that is conflict free for ITS writes and
// reads): slot = 17 L + i[6:3] with L = the pass-2 lane number (i10 i9 i8 i7 i1 i0).
__device__ __forceinline__ void xchg_32(u64 (&x)[32], int lane, u64 *wl) {
    const u64 *rp = wl + 17 * lane;                        // as pass-2 lane; + i[6:3]
    u64 *wp = wl + 68 * (lane >> 4) + (lane & 15);         // as pass-3 lane i[8:3]: 17 ((i8 i7) << 2) + i[6:3]; + 17 ((i10 i9) << 4 | (i1 i0))
    u64 y[32];
    W14_PRIO_UP();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int R = 0; R < 16; ++R) wp[17 * (((R >> 2) << 4) | (R & 3))] = x[((R >> 2) << 3) | (h << 2) | (R & 3)];
        wave_sync();
#pragma unroll
        for (int c = 0; c < 16; ++c) y[(h << 4) | c] = rp[c];
        wave_sync();
    }
    W14_PRIO_DOWN();
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = y[r];
}

// units (arith.hpp): passes 0..2 are replicas sharing twiddles; pass 3 is one replica (i[10:9] = AB) at a time with its own
template <int R0, int l> using P0 = Unit<0, R0, l, 0, (32 >> R0), (1 << R0), true>;
template <int R0, int l> using P1 = Unit<R0, 4, l, 0, 2, 16, true>;
template <int R0, int l> using P2 = Unit<R0 + 4, 4, l, 0, 2, 16, true>;
template <int R0, int l, int AB> using P3 = Unit<R0 + 8, 3, l, AB, 1, 8, false, 6>;  // block prefix (w << 8) | (AB << 6) | lane

// pass 3 of the inverse: refill each layer's twiddle slot one replica ahead (true) or fetch whole replicas two ahead (false)
#ifndef W14_P3_REFILL
#define W14_P3_REFILL -1
#endif
template <class A>
__device__ __host__ constexpr bool w14_p3_refill() {
    return W14_P3_REFILL < 0 ? std::is_same<typename A::TwRaw, uint4>::value : W14_P3_REFILL != 0;
}

// one replica's pass-3 twiddles (layers 11, 12, 13)
template <class A>
struct Tw7 {
    typename A::TwRaw l0[1], l1[2], l2[4];
};
template <class A, bool INV, int R0, int AB>
__device__ __forceinline__ void tw7_load(Tw7<A> &b, int t3, const typename A::K &k) {
    tw_load<A, INV, P3<R0, 0, AB>>(b.l0, t3, k);
    tw_load<A, INV, P3<R0, 1, AB>>(b.l1, t3, k);
    tw_load<A, INV, P3<R0, 2, AB>>(b.l2, t3, k);
}

// HBM side of pass 0: register (pass, n) <-> coefficient (n << 11) | (pass << (6 + R0)) | t, consecutive lanes on consecutive words
template <int R0>
__device__ __forceinline__ void load_p0(u64 (&x)[32], const u64 *__restrict__ g, int t) {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const int n = r & ((1 << R0) - 1), pass = r >> R0;
#ifdef W14_ABLATE_NO_GLOBAL  // developer lab only: no HBM traffic
        x[r] = (u64)(t + r) * 0x9E3779B97F4A7C15ull >> 5;
#else
        x[r] = g[(n << 11) | (pass << (6 + R0)) | t];
#endif
    }
}
// HBM side of pass 3 (inverse loads): register (ab, n3) <-> coefficient (w << 11) | (ab << 9) | (lane << 3) | n3
template <int AB>
__device__ __forceinline__ void load_p3(u64 (&x)[32], const u64 *__restrict__ src) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + (AB << 9) + j);
        x[AB * 8 + j] = v.x; x[AB * 8 + j + 1] = v.y;
    }
}
// ... with the pointwise product of util/src/ring/fft/zq.rs:17 fused in: x <- src (.) mul, both canonical evaluations
template <class A, int AB>
__device__ __forceinline__ void load_mul_p3(u64 (&x)[32], const u64 *__restrict__ src, const u64 *__restrict__ mul, const typename A::K &k) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + (AB << 9) + j);
        const ulonglong2 m = *reinterpret_cast<const ulonglong2 *>(mul + (AB << 9) + j);
        x[AB * 8 + j] = A::mulvar(v.x, m.x, k); x[AB * 8 + j + 1] = A::mulvar(v.y, m.y, k);
    }
}

// Forward stores.  A pass-3 lane owns 64 contiguous bytes per (i10 i9) value; stored as they stand, every 16-byte store
// instruction would touch 32 lines (measured: the store side alone then runs at 4.4 TB/s instead of 5.0).  The finished
// replica goes through the wave's LDS region once more (slot e + 2 (e >> 4): ds_write_b128 conflict free, ds_read_b128 two-way)
// and leaves as four 1 KiB-contiguous store instructions.
template <class A, int AB>
__device__ __forceinline__ void store_p3(u64 (&x)[32], u64 *__restrict__ dst_wave, int lane, u64 *wl, const typename A::K &k) {
    ulonglong2 *wr = reinterpret_cast<ulonglong2 *>(wl + 8 * lane + (lane & ~1));
    const ulonglong2 *rd = reinterpret_cast<const ulonglong2 *>(wl + 2 * lane + 2 * (lane >> 3));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ulonglong2 v;
        v.x = A::canon_fwd(x[AB * 8 + 2 * j], k);
        v.y = A::canon_fwd(x[AB * 8 + 2 * j + 1], k);
        wr[j] = v;
    }
    wave_sync();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {  // back into the replica's own registers (a local array here is kept in scratch memory)
        const ulonglong2 v = rd[72 * kk];  // 144 slots
        x[AB * 8 + 2 * kk] = v.x; x[AB * 8 + 2 * kk + 1] = v.y;
    }
    wave_sync();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        ulonglong2 v;
        v.x = x[AB * 8 + 2 * kk]; v.y = x[AB * 8 + 2 * kk + 1];
#ifdef W14_ABLATE_NO_GLOBAL
        if (v.x == 0xdeadbeefcafef00dull)
#endif
        *reinterpret_cast<ulonglong2 *>(dst_wave + (AB << 9) + 128 * kk + 2 * lane) = v;
    }
}

// One forward transform.  x[] arrives loaded (pass-0 layout, the loads possibly still in flight).
// KEEP: nothing is stored -- every finished replica is left canonical in the pass-3 layout, exactly what the inverse transform loads.
template <class A, int R0, bool KEEP = false>
__device__ __forceinline__ void fwd_one(u64 (&x)[32], u64 *__restrict__ g, const typename A::K &k, u64 *lds, u64 *wl,
                                        const int t, const int lane, const int w STAMP_ENTRY_PARAM) {
    typedef typename A::TwRaw Tw;
    STAMP_DECL;
    STAMP_REAL(10);
    STAMP(0);
    {   // pass 0: layers 0..R0-1, twiddles wave-uniform
        Tw a0[1], a1[2], a2[4], a3[8];
        tw_load<A, false, P0<R0, 0>>(a0, 0, k);
        if constexpr (R0 >= 2) tw_load<A, false, P0<R0, (R0 >= 2 ? 1 : 0)>>(a1, 0, k);
        if constexpr (R0 >= 3) tw_load<A, false, P0<R0, (R0 >= 3 ? 2 : 0)>>(a2, 0, k);
        if constexpr (R0 == 4) tw_load<A, false, P0<R0, (R0 == 4 ? 3 : 0)>>(a3, 0, k);
        STAMP(1);
        FHE_SCHED_FENCE();
        ct_apply<A, P0<R0, 0>>(x, a0, k);
        if constexpr (R0 >= 2) {
            FHE_SCHED_FENCE();
            ct_apply<A, P0<R0, (R0 >= 2 ? 1 : 0)>>(x, a1, k);
        }
        if constexpr (R0 >= 3) {
            FHE_SCHED_FENCE();
            ct_apply<A, P0<R0, (R0 >= 3 ? 2 : 0)>>(x, a2, k);
        }
        if constexpr (R0 == 4) {
            FHE_SCHED_FENCE();
            ct_apply<A, P0<R0, (R0 == 4 ? 3 : 0)>>(x, a3, k);
        }
    }
    FHE_SCHED_FENCE();
    Tw b0[1], b1[2], b2[4], b3[8];  // pass 1: wave-uniform as well (block prefix = w): fetched behind the barriers of X01
    tw_load<A, false, P1<R0, 0>>(b0, w, k); tw_load<A, false, P1<R0, 1>>(b1, w, k); tw_load<A, false, P1<R0, 2>>(b2, w, k); tw_load<A, false, P1<R0, 3>>(b3, w, k);
    STAMP(2);
    xchg_01<R0>(x, t, w, lds);
    STAMP(3);
    if constexpr (A::PASS_FOLD) {
#pragma unroll
        for (int r = 0; r < 32; ++r) x[r] = A::fold(x[r], k);
    }
    FHE_SCHED_FENCE();
    ct_apply<A, P1<R0, 0>>(x, b0, k);
    FHE_SCHED_FENCE();
    ct_apply<A, P1<R0, 1>>(x, b1, k);
    FHE_SCHED_FENCE();
    ct_apply<A, P1<R0, 2>>(x, b2, k);
    FHE_SCHED_FENCE();
    ct_apply<A, P1<R0, 3>>(x, b3, k);
    FHE_SCHED_FENCE();
    const int t2 = (w << 4) | (lane >> 2), t3 = (w << 8) | lane;
    Tw c0[1], c1[2], c2[4], c3[8];  // pass 2: per lane; the first seven ride through X12
    tw_load<A, false, P2<R0, 0>>(c0, t2, k); tw_load<A, false, P2<R0, 1>>(c1, t2, k); tw_load<A, false, P2<R0, 2>>(c2, t2, k);
    STAMP(4);
    xchg_12(x, lane, wl);
    STAMP(5);
    if constexpr (A::PASS_FOLD) {
#pragma unroll
        for (int r = 0; r < 32; ++r) x[r] = A::fold(x[r], k);
    }
    FHE_SCHED_FENCE();
    ct_apply<A, P2<R0, 0>>(x, c0, k);
    FHE_SCHED_FENCE();
    ct_apply<A, P2<R0, 1>>(x, c1, k);
    FHE_SCHED_FENCE();
    tw_load<A, false, P2<R0, 3>>(c3, t2, k);
    ct_apply<A, P2<R0, 2>>(x, c2, k);
    FHE_SCHED_FENCE();
    ct_apply<A, P2<R0, 3>>(x, c3, k);
    FHE_SCHED_FENCE();
    Tw7<A> d[2];  // pass 3: one replica ahead
    tw7_load<A, false, R0, 0>(d[0], t3, k);
    STAMP(6);
    xchg_23(x, lane, wl);
    STAMP(7);
    if constexpr (A::PASS_FOLD) {
#pragma unroll
        for (int r = 0; r < 32; ++r) x[r] = A::fold(x[r], k);
    }
    u64 *dst_wave = g + (w << 11);
    static_for<0, 4>([&](auto abc) {
        constexpr int ab = decltype(abc)::value;
        FHE_SCHED_FENCE();
        ct_apply<A, P3<R0, 0, ab>>(x, d[ab & 1].l0, k);
        ct_apply<A, P3<R0, 1, ab>>(x, d[ab & 1].l1, k);
        FHE_SCHED_FENCE();
        if constexpr (ab < 3) tw7_load<A, false, R0, (ab < 3 ? ab + 1 : 3)>(d[(ab + 1) & 1], t3, k);
        ct_apply<A, P3<R0, 2, ab>>(x, d[ab & 1].l2, k);
        FHE_SCHED_FENCE();
        if constexpr (KEEP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[ab * 8 + j] = A::canon_fwd(x[ab * 8 + j], k);
        } else {
            store_p3<A, ab>(x, dst_wave, lane, wl, k);
        }
    });
    STAMP(8);
#ifdef NTT14_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // diagnostic builds only: when have the stores left?
#endif
    STAMP(9);
    STAMP_REAL(11);
    STAMP_FLUSH();
}

// ---- pass 3 of the inverse in diagonal form (two-operand policy) -------------------------------------------------------
// The inverse meets its N/2 + N/4 + N/8 per-lane twiddles FIRST, together with the HBM loads, where nothing can hide their
// latency (the forward meets them last and fetches a pass ahead).  Within one block of 8 contiguous coefficients the three
// layers factor as  diag(1, t, .., t^7) . F8  with t = twi[4 blk] and F8 the same network with t = 1: its only twiddles are
// I^-1 = twi[1], J^-1 = twi[2] and (I J)^-1 = twi[3] (psi^(-N/2), psi^(-N/4), psi^(-3N/4): wave uniform, scalar registers).
// So the butterflies of a replica start as soon as its coefficients arrive, and the seven per-lane multipliers t^p -- a table
// laid out [p][blk], consecutive lanes on consecutive entries -- are first needed 24 additions and 5 products later.
// Same multiplications (12 per block), same additions as three butterfly layers; results equal mod q, so canonical outputs
// are bit-identical.  Bounds (inputs canonical): every sum stays below 8.2 q < 2^64; products come back below q + 9c.
template <class A>
__device__ __host__ constexpr bool w14_p3_diag() {
#ifdef W14_NO_DIAG
    return false;
#else
    return std::is_same<typename A::TwRaw, uint4>::value;
#endif
}

template <class A, int R0, int AB>
__device__ __forceinline__ int p3_blk(int t3, const typename A::K &k) {
    return (1 << (R0 + 8 + k.pb)) + ((k.prefix << (R0 + 8)) | t3 | (AB << 6));
}
template <class A, int R0, int AB>
__device__ __forceinline__ void p3_diag_load(uint4 (&e)[7], int t3, const typename A::K &k) {
    const int blk = p3_blk<A, R0, AB>(t3, k);
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        const FHE_CONST uint4 *q = k.tw3i + size_t(p) * k.tw3_stride + blk;
        e[p] = uint4{q->x, q->y, q->z, q->w};
    }
}
// F8 on x[8 AB .. 8 AB + 7] (position = register index), then the diagonal; NEXT >= 0: every multiplier slot is refilled for
// replica NEXT as soon as it has been used
template <class A, int R0, int AB, int NEXT>
__device__ __forceinline__ void p3_diag_apply(u64 (&x)[32], uint4 (&e)[7], const uint4 &ci, const uint4 &cj, const uint4 &cij, int t3,
                                              const typename A::K &k) {
    const auto &m = k.m;
    u64 *v = x + 8 * AB;
    // layer A: pairs (2j, 2j + 1)
    const u64 s0 = v[0] + v[1], t0 = v[0] + m.q2 - v[1];
    const u64 s1 = v[2] + v[3], t1 = A::mul(v[2] + m.q2 - v[3], ci, m);
    const u64 s2 = v[4] + v[5], t2 = A::mul(v[4] + m.q2 - v[5], cj, m);
    const u64 s3 = v[6] + v[7], t3v = A::mul(v[6] + m.q2 - v[7], cij, m);
    // layer B: pairs (4j + k, 4j + 2 + k)
    const u64 u0 = s0 + s1, e0 = s0 + m.q4 - s1;              // s < 2q
    const u64 u1 = t0 + t1, e1 = t0 + m.q2 - t1;              // t0 < 3q, t1 < q + 9c
    const u64 u2 = s2 + s3, e2 = A::mul(s2 + m.q4 - s3, ci, m);
    const u64 u3 = t2 + t3v, e3 = A::mul(t2 + m.q2 - t3v, ci, m);
    // layer C: pairs (k, 4 + k); then the diagonal
    const int blk = NEXT >= 0 ? p3_blk<A, R0, (NEXT >= 0 ? NEXT : 0)>(t3, k) : 0;
    auto refill = [&](int p) {
        if constexpr (NEXT >= 0) {
            const FHE_CONST uint4 *q = k.tw3i + size_t(p) * k.tw3_stride + blk;
            e[p] = uint4{q->x, q->y, q->z, q->w};
        }
    };
    v[0] = A::fold1(u0 + u2, m);                              // < 8q: the one value no product reduces
    v[1] = A::mul(u1 + u3, e[0], m); refill(0);
    v[2] = A::mul(e0 + e2, e[1], m); refill(1);
    v[3] = A::mul(e1 + e3, e[2], m); refill(2);
    v[4] = A::mul(u0 + m.q4 - u2, e[3], m); refill(3);        // u2 < 4q
    v[5] = A::mul(u1 + m.q4 - u3, e[4], m); refill(4);        // u3 < 2q + 18c
    v[6] = A::mul(e0 + m.q2 - e2, e[5], m); refill(5);        // e2 < q + 9c
    v[7] = A::mul(e1 + m.q2 - e3, e[6], m); refill(6);
}

// One inverse transform; x[] arrives loaded in the pass-3 layout, d[] with the pass-3 twiddles of replicas 0 and 1 (fetched BEFORE
// the coefficients: vmcnt retires in order, and the twiddles are L2 hits).
template <class A, bool PFX, int R0>
__device__ __forceinline__ void inv_one(u64 (&x)[32], Tw7<A> (&d)[2], u64 *__restrict__ g, const typename A::K &k, u64 *lds, u64 *wl,
                                        const int t, const int lane, const int w STAMP_ENTRY_PARAM) {
    typedef typename A::TwRaw Tw;
    const int t2 = (w << 4) | (lane >> 2), t3 = (w << 8) | lane;
    STAMP_DECL;
    STAMP_REAL(10);
    STAMP(0);
    // pass 3: layers 13, 12, 11.  Eight-byte twiddles are fetched two replicas ahead (d[] arrives holding replicas 0 and 1); sixteen-
    // byte ones (ArithDS) refill each layer's slot for the NEXT replica as soon as the layer has used it -- 28 registers instead
    // of 56 beside the 64 of x.  The two schemes measure the same (tools/ntt_lab2.hip: 0.331 / 0.333 ms); the refill is kept for the
    // registers it leaves free.
    constexpr bool REFILL = w14_p3_refill<A>();
    Tw c3[8], c2[4], c1[2], c0[1];
    if constexpr (w14_p3_diag<A>()) {
        uint4(&e)[7] = reinterpret_cast<uint4(&)[7]>(d[0]);   // Tw7 of a 16-byte policy = seven uint4
        const uint4 ci = A::template fetch<true>(k, 1), cj = A::template fetch<true>(k, 2), cij = A::template fetch<true>(k, 3);
        FHE_SCHED_FENCE();
        p3_diag_apply<A, R0, 0, 1>(x, e, ci, cj, cij, t3, k);
        FHE_SCHED_FENCE();
        p3_diag_apply<A, R0, 1, 2>(x, e, ci, cj, cij, t3, k);
        FHE_SCHED_FENCE();
        p3_diag_apply<A, R0, 2, 3>(x, e, ci, cj, cij, t3, k);
        FHE_SCHED_FENCE();
        p3_diag_apply<A, R0, 3, -1>(x, e, ci, cj, cij, t3, k);
    } else
    static_for<0, 4>([&](auto abc) {
        constexpr int ab = decltype(abc)::value, nx = ab < 3 ? ab + 1 : 3;
        Tw7<A> &e = d[REFILL ? 0 : ab & 1];
        FHE_SCHED_FENCE();
        gs_apply<A, P3<R0, 2, ab>, 0>(x, e.l2, k);
        if constexpr (REFILL && ab < 3) tw_load<A, true, P3<R0, 2, nx>>(e.l2, t3, k);
        FHE_SCHED_FENCE();
        gs_apply<A, P3<R0, 1, ab>, 1>(x, e.l1, k);
        if constexpr (REFILL && ab < 3) tw_load<A, true, P3<R0, 1, nx>>(e.l1, t3, k);
        gs_apply<A, P3<R0, 0, ab>, 2>(x, e.l0, k);
        if constexpr (REFILL && ab < 3) tw_load<A, true, P3<R0, 0, nx>>(e.l0, t3, k);
        FHE_SCHED_FENCE();
        if constexpr (!REFILL && ab < 2) tw7_load<A, true, R0, (ab < 2 ? ab + 2 : 3)>(d[ab & 1], t3, k);
    });
    FHE_SCHED_FENCE();
    STAMP(1);
    // pass 2's first twiddle set is requested BEHIND the exchange when it is 32 registers wide: held across the exchange it pushes
    // the allocator over 128 registers, and a spilled twiddle comes back through scratch memory behind every load in flight
    // (vmcnt retires in order): 0.307 -> 0.294 ms per 4096 transforms
    constexpr bool C3_LATE = sizeof(Tw) > 8;
    if constexpr (!C3_LATE) tw_load<A, true, P2<R0, 3>>(c3, t2, k);
    xchg_32(x, lane, wl);
    STAMP(2);
    FHE_SCHED_FENCE();
    if constexpr (C3_LATE) tw_load<A, true, P2<R0, 3>>(c3, t2, k);
    tw_load<A, true, P2<R0, 2>>(c2, t2, k);
    gs_apply<A, P2<R0, 3>, 3>(x, c3, k);
    FHE_SCHED_FENCE();
    tw_load<A, true, P2<R0, 1>>(c1, t2, k); tw_load<A, true, P2<R0, 0>>(c0, t2, k);
    gs_apply<A, P2<R0, 2>, 4>(x, c2, k);
    FHE_SCHED_FENCE();
    gs_apply<A, P2<R0, 1>, 5>(x, c1, k);
    FHE_SCHED_FENCE();
    gs_apply<A, P2<R0, 0>, 6>(x, c0, k);
    FHE_SCHED_FENCE();
    STAMP(3);
    Tw b3[8], b2[4], b1[2], b0[1];  // wave-uniform
    tw_load<A, true, P1<R0, 3>>(b3, w, k); tw_load<A, true, P1<R0, 2>>(b2, w, k); tw_load<A, true, P1<R0, 1>>(b1, w, k); tw_load<A, true, P1<R0, 0>>(b0, w, k);
    xchg_21(x, lane, wl);
    STAMP(4);
    FHE_SCHED_FENCE();
    gs_apply<A, P1<R0, 3>, 7>(x, b3, k);
    FHE_SCHED_FENCE();
    gs_apply<A, P1<R0, 2>, 8>(x, b2, k);
    FHE_SCHED_FENCE();
    gs_apply<A, P1<R0, 1>, 9>(x, b1, k);
    FHE_SCHED_FENCE();
    gs_apply<A, P1<R0, 0>, 10>(x, b0, k);
    FHE_SCHED_FENCE();
    STAMP(5);
    Tw a3[8], a2[4], a1[2], a0[1];
    if constexpr (R0 == 4) tw_load<A, true, P0<R0, (R0 == 4 ? 3 : 0)>>(a3, 0, k);
    if constexpr (R0 >= 3) tw_load<A, true, P0<R0, (R0 >= 3 ? 2 : 0)>>(a2, 0, k);
    if constexpr (R0 >= 2) tw_load<A, true, P0<R0, (R0 >= 2 ? 1 : 0)>>(a1, 0, k);
    if constexpr (PFX) tw_load<A, true, P0<R0, 0>>(a0, 0, k);
    xchg_10<R0>(x, t, w, lds);
    STAMP(6);
    if constexpr (R0 == 4) {
        FHE_SCHED_FENCE();
        gs_apply<A, P0<R0, (R0 == 4 ? 3 : 0)>, 11>(x, a3, k);
    }
    if constexpr (R0 >= 3) {
        FHE_SCHED_FENCE();
        gs_apply<A, P0<R0, (R0 >= 3 ? 2 : 0)>, 8 + R0>(x, a2, k);
    }
    if constexpr (R0 >= 2) {
        FHE_SCHED_FENCE();
        gs_apply<A, P0<R0, (R0 >= 2 ? 1 : 0)>, 9 + R0>(x, a1, k);
    }
    // the last layer leaves canonical values: a whole ring folds n^-1 into it (the difference branch multiplies by twi[1] n^-1),
    // a sub-transform of a larger ring is not scaled here at all
    STAMP(7);
    typename A::TwReg wlast{};
    if constexpr (PFX) wlast = A::prep(a0[0]);
    constexpr int LAST_PH = A::GS_SPAN > 0 ? (10 + R0) % (A::GS_SPAN > 0 ? A::GS_SPAN : 1) : 1;  // layers since the sums were last folded
    constexpr int HALF = 1 << (R0 - 1), REPS = 32 >> R0, CH = HALF < 4 ? HALF : 4;
    static_for<0, REPS * (HALF / CH)>([&](auto cc) {  // (up to) four butterflies at a time, stored as they finish
        constexpr int pass = decltype(cc)::value / (HALF / CH), j0 = (decltype(cc)::value % (HALF / CH)) * CH;
        FHE_SCHED_FENCE();
#pragma unroll
        for (int j = j0; j < j0 + CH; ++j) {
            const int o = (pass << R0) + j;
            if constexpr (PFX) A::template gs_last_plain<LAST_PH>(x[o], x[o + HALF], wlast, k);
            else A::template gs_last_scaled<LAST_PH>(x[o], x[o + HALF], k);
            g[(j << 11) | (pass << (6 + R0)) | t] = x[o];
            g[((j + HALF) << 11) | (pass << (6 + R0)) | t] = x[o + HALF];
        }
    });
    STAMP(8);
#ifdef NTT14_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STAMP(9);
    STAMP_REAL(11);
    STAMP_FLUSH();
}

// Which sub-polynomial a workgroup takes.  A launch over several moduli (the RNS limbs of CKKS: polynomial p uses descs[p % n_desc])
// comes as a 2-D grid, y = the modulus, x = that modulus's sub-polynomials across the batch: workgroups are dispatched x first, so
// the ~512 that are resident together share a few twiddle tables (768 KiB each at 60 bits) instead of all n_desc of them -- 16
// tables are 12 MiB against 4 MiB of L2 per XCD.
__device__ __forceinline__ unsigned sub_of_block(int pb, unsigned n_desc) {
    if (gridDim.y == 1) return blockIdx.x;
    return ((((blockIdx.x >> pb) * n_desc) + blockIdx.y) << pb) | (blockIdx.x & ((1u << pb) - 1));
}

}  // namespace w14

// One workgroup = one (sub-)polynomial.  (A persistent variant -- two workgroups per CU looping over polynomials, the next one's
// HBM loads issued behind the last twiddle fetch of the current one -- measured 8-10 % SLOWER at 4096 polynomials than letting
// the hardware dispatch one-polynomial workgroups: 0.319-0.324 ms against 0.293 ms, tools/ntt_lab2.hip.)
// PFX = false: whole rings of 2^(11 + R0) (pb = 0).  PFX = true: sub s is sub-transform s & (2^pb - 1) of polynomial s >> pb.
template <class A, bool PFX, int R0 = 3>
__global__ __launch_bounds__(w14::threads<R0>(), 4) void ntt14w_fwd_kernel(u64 *__restrict__ data, const ModDesc *__restrict__ descs,
                                                                             unsigned n_desc, unsigned subs, int pb, NttIo io) {
    STAMP_ENTRY();
    constexpr int LOG_N = 11 + R0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned sub = w14::sub_of_block(PFX ? pb : 0, n_desc);
    const unsigned poly = PFX ? sub >> pb : sub;
    const ModDesc &D = descs[n_desc == 1 ? 0 : poly % n_desc];
    const typename A::K k = A::make(D, LOG_N, PFX ? pb : 0, PFX ? int(sub & ((1u << pb) - 1)) : 0);
    u64 *g = data + (size_t(sub) << LOG_N);
    const u64 *gs = ntt_src(io, sub, LOG_N, g);
    u64 x[32];
    w14::load_p0<R0>(x, gs, t);
    w14::fwd_one<A, R0>(x, g, k, lds, lds + w * w14::WSLOTS, t, lane, w STAMP_ENTRY_ARG);
}

// MUL: the launch carries a pointwise multiplier (io.mul), a separate instantiation so that plain transforms do not carry its code
template <class A, bool PFX, bool MUL = false, int R0 = 3>
__global__ __launch_bounds__(w14::threads<R0>(), 4) void ntt14w_inv_kernel(u64 *__restrict__ data, const ModDesc *__restrict__ descs,
                                                                             unsigned n_desc, unsigned subs, int pb, NttIo io) {
    STAMP_ENTRY();
    constexpr int LOG_N = 11 + R0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned sub = w14::sub_of_block(PFX ? pb : 0, n_desc);
    const unsigned poly = PFX ? sub >> pb : sub;
    const ModDesc &D = descs[n_desc == 1 ? 0 : poly % n_desc];
    const typename A::K k = A::make(D, LOG_N, PFX ? pb : 0, PFX ? int(sub & ((1u << pb) - 1)) : 0);
    u64 *g = data + (size_t(sub) << LOG_N);
    if constexpr (MUL) {
        if (io.dst2) {  // split output (NttIo::dst2): the launch's source is io.src, `g` is only where this workgroup stores
            const unsigned grp = poly / io.dst_period, l = poly - grp * io.dst_period, part = PFX ? sub & ((1u << pb) - 1) : 0;
            const unsigned pbs = PFX ? pb : 0;
            g = l < io.dst_first2 ? data + (((size_t(grp) * io.dst_first2 + l) << pbs | part) << LOG_N)
                                  : io.dst2 + (((size_t(grp) * (io.dst_period - io.dst_first2) + (l - io.dst_first2)) << pbs | part) << LOG_N);
        }
    }
    u64 x[32];
    w14::Tw7<A> d[2];
    auto load_first_twiddles = [&]() {
        if constexpr (w14::w14_p3_diag<A>()) w14::p3_diag_load<A, R0, 0>(reinterpret_cast<uint4(&)[7]>(d[0]), (w << 8) | lane, k);
        else w14::tw7_load<A, true, R0, 0>(d[0], (w << 8) | lane, k);
        if constexpr (!w14::w14_p3_diag<A>() && !w14::w14_p3_refill<A>()) w14::tw7_load<A, true, R0, 1>(d[1], (w << 8) | lane, k);
    };
    // plain transform: twiddles BEFORE coefficients (vmcnt retires in order, the twiddles are L2 hits).  With a multiplier the
    // coefficients and their multipliers come first (two 64-register sets in flight) and the first twiddles are requested behind the
    // products: held across them they are the registers that spill (4-5 per lane, 20 bytes of scratch traffic per lane)
    if constexpr (!MUL) load_first_twiddles();
    const int off = (w << 11) | (lane << 3);
    const u64 *src = (io.src ? io.src + (size_t(sub % io.src_mod) << LOG_N) : g) + off;
    if constexpr (MUL) {
        const u64 *mul = io.mul + ((size_t(sub / io.mul_div) * io.mul_period + sub % io.mul_period) << LOG_N) + off;
        w14::load_mul_p3<A, 0>(x, src, mul, k); w14::load_mul_p3<A, 1>(x, src, mul, k);
        w14::load_mul_p3<A, 2>(x, src, mul, k); w14::load_mul_p3<A, 3>(x, src, mul, k);
        FHE_SCHED_FENCE();
        load_first_twiddles();
    } else {
        w14::load_p3<0>(x, src); w14::load_p3<1>(x, src); w14::load_p3<2>(x, src); w14::load_p3<3>(x, src);
    }
    w14::inv_one<A, PFX, R0>(x, d, g, k, lds, lds + w * w14::WSLOTS, t, lane, w STAMP_ENTRY_ARG);
}

// util/src/ring/fft/zq.rs:14-19 with the left operand never leaving the chip: forward transform of polynomial s, pointwise product
// with the evaluations io.mul holds for it (the right operand, transformed by an ordinary forward launch), inverse transform -- one
// workgroup, one load and one store of the polynomial.  As two launches the left operand's evaluations cross HBM twice and the
// inverse that multiplies on its load is the most memory-heavy launch of the three (2048 polynomials: 0.137 + 0.262 ms).
template <class A, int R0 = 3>
__global__ __launch_bounds__(w14::threads<R0>(), 4) void ntt14w_mul_kernel(u64 *__restrict__ data, const ModDesc *__restrict__ descs,
                                                                             unsigned n_desc, unsigned subs, int pb, NttIo io) {
    STAMP_ENTRY();
    constexpr int LOG_N = 11 + R0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned sub = w14::sub_of_block(0, n_desc);
    const ModDesc &D = descs[n_desc == 1 ? 0 : sub % n_desc];
    const typename A::K k = A::make(D, LOG_N, 0, 0);
    u64 *g = data + (size_t(sub) << LOG_N);
    const u64 *mul_poly = io.mul + ((size_t(sub / io.mul_div) * io.mul_period + sub % io.mul_period) << LOG_N);  // wave uniform
    u64 x[32];
    w14::load_p0<R0>(x, g, t);
    w14::fwd_one<A, R0, true>(x, g, k, lds, lds + w * w14::WSLOTS, t, lane, w STAMP_ENTRY_ARG);
    // the inverse half sees the thread index as a fresh value: its ~100 addresses (LDS slots, twiddle entries, stores) would otherwise
    // be computed at the top of the kernel and live -- spilled -- through the whole forward half
    int ti = t;
    asm volatile("" : "+v"(ti));
    const int lane_i = ti & 63, wi = __builtin_amdgcn_readfirstlane(ti >> 6);
    {   // util/src/ring/fft/zq.rs:17 `a[i] *= b[i]`, two replicas at a time (their multipliers: 32 registers beside the 64 of x)
        const u64 *mp = mul_poly + ((wi << 11) | (lane_i << 3));
        static_for<0, 2>([&](auto hc) {
            constexpr int h = decltype(hc)::value;
            ulonglong2 mv[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) mv[r] = *reinterpret_cast<const ulonglong2 *>(mp + ((2 * h + (r >> 2)) << 9) + 2 * (r & 3));
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int o = (2 * h + (r >> 2)) * 8 + 2 * (r & 3);
                x[o] = A::mulvar(x[o], mv[r].x, k);
                x[o + 1] = A::mulvar(x[o + 1], mv[r].y, k);
            }
            FHE_SCHED_FENCE();
        });
    }
    w14::Tw7<A> d[2];
    if constexpr (w14::w14_p3_diag<A>()) w14::p3_diag_load<A, R0, 0>(reinterpret_cast<uint4(&)[7]>(d[0]), (wi << 8) | lane_i, k);
    else w14::tw7_load<A, true, R0, 0>(d[0], (wi << 8) | lane_i, k);
    if constexpr (!w14::w14_p3_diag<A>() && !w14::w14_p3_refill<A>()) w14::tw7_load<A, true, R0, 1>(d[1], (wi << 8) | lane_i, k);
    w14::inv_one<A, false, R0>(x, d, g, k, lds, lds + wi * w14::WSLOTS, ti, lane_i, wi STAMP_ENTRY_ARG);
}

}  // namespace fhe
