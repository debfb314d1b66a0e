// N = 2^14 transform kernels tuned for MI355X (BASELINE config 2, and the sub-transforms of 2^15..2^17 rings).
//
// Why a dedicated kernel: a 2^14-coefficient polynomial of u64 is 128 KiB.  Kept whole in LDS (the generic kernel in
// ntt_kernels.hpp) only ONE workgroup fits a CU, its 16 waves move through load / exchange / butterfly phases in
// lockstep, and the integer pipes idle whenever the LDS or HBM phase runs.  Here a polynomial lives in the REGISTERS
// of a 512-thread workgroup (32 coefficients per thread) and LDS only carries the exchanges between passes, half of
// the polynomial at a time (68 KiB with padding): TWO workgroups are resident per CU and one computes while the other
// exchanges or waits for HBM.
//
// Bit layout (i = coefficient index, 14 bits).  Every pass owns 5 index bits in registers: 4 it processes (a radix-16
// butterfly network = 4 reference layers, util/src/ring/fft.rs:40-77) and one "passive" bit that is shared with the
// neighbouring pass, so an exchange can be split by that bit into two rounds that each move half the polynomial:
//
//   pass 0: regs {13..10 | 9}    thread = bits 8..0           layers 0..3    HBM side: 8-byte coalesced accesses
//   pass 1: regs {9..6   | 5}    thread = bits 13..10, 4..0   layers 4..7
//   pass 2: regs {5..2   | 1}    thread = bits 13..6, 0       layers 8..11
//   pass 3: regs {1,0 | 13..11}  thread = bits 10..2          layers 12..13  HBM side: 32 contiguous bytes per lane
//
// The forward transform runs passes 0 -> 3 (Cooley-Tukey), the inverse 3 -> 0 (Gentleman-Sande; n^-1 is folded into its
// last layer).
// Natural-order coefficients <-> bit-reversed evaluations, twiddle tw[2^layer + block], exactly as the reference.
//
// Arithmetic is a policy: ArithShoup (any prime < 2^62: Shoup multiplication + Harvey lazy reduction) or ArithPM
// (pseudo-Mersenne primes q = 2^b - c, c <= 2^(b-33), b <= 60 -- every prime `two_adic_primes(bits, log_n)` yields for
// the BASELINE configs: 8 v_mad_u64_u32 + 4 other instructions per product on an 8-byte twiddle, no companion table, no
// compare/select anywhere in a butterfly; arith.hpp).  Both end with canonical values in [0, q): bit-identical results.
#pragma once
#include "ntt_kernels.hpp"

namespace fhe {

// ---------------------------------------------------------------------------------------------------------
// half-image exchanges.  The image holds the 8192 coefficients whose split bit equals `round`; j = index with the
// split bit removed; 2 pad slots per 32 keep 16-byte alignment and make every access pattern below conflict free.
// ---------------------------------------------------------------------------------------------------------
// An exchange is a short burst of LDS instructions between barriers: latency critical.  The other resident workgroup is
// usually in a butterfly pass (throughput bound), so the exchanging waves take issue priority while they exchange.
#ifndef NTT14_XCHG_PRIO
#define NTT14_XCHG_PRIO 1
#endif
#if NTT14_XCHG_PRIO
#define XCHG_PRIO_UP() __builtin_amdgcn_s_setprio(3)
#define XCHG_PRIO_DOWN() __builtin_amdgcn_s_setprio(0)
#else
#define XCHG_PRIO_UP()
#define XCHG_PRIO_DOWN()
#endif
constexpr int N14_THREADS = 512;
constexpr int N14_IMG = 8192 + (8192 >> 5) * 2;            // u64 slots
constexpr size_t N14_LDS_BYTES = size_t(N14_IMG) * 8;       // 69 632 B -> two workgroups per CU
__device__ __forceinline__ int img(int j) { return j + ((j >> 5) << 1); }

// register naming: pass p keeps coefficient (processed nibble n4, passive bit s) in x[(s << 4) | n4]
//   pass 0: i = (n4 << 10) | (s << 9) | t
//   pass 1: i = (hi4 << 10) | (n4 << 6) | (s << 5) | lo5      thread u = (hi4 << 5) | lo5
//   pass 2: i = (hi8 << 6) | (n4 << 2) | (s << 1) | lo1       thread v = (hi8 << 1) | lo1
//   pass 3: i = (s3 << 11) | (z << 2) | p2                    thread z;  x[(s3 << 2) | p2]

// pass 0 regs -> pass 1 regs (split bit 9)
__device__ __forceinline__ void xchg_01(u64 (&x)[32], int t, u64 *lds) {
    XCHG_PRIO_UP();
    const int hi4 = t >> 5, lo5 = t & 31;
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
#pragma unroll
        for (int n4 = 0; n4 < 16; ++n4) lds[img((n4 << 9) | t)] = x[(rnd << 4) | n4];
        __syncthreads();
        u64 y[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {  // m = (n4' low 3 bits, s')  ->  pass-1 register (s' << 4) | (rnd << 3) | n3
            const int n3 = m >> 1, s = m & 1;
            y[m] = lds[img((hi4 << 9) | (n3 << 6) | (s << 5) | lo5)];
        }
        __syncthreads();
        // park the received half: round 0 fills pass-1 registers with n4'[3] = 0 -- which x slots are free?  Only the
        // ones already sent.  Round 0 sent x[0..15]; it may overwrite exactly those, so received data is parked there and
        // the final register order is fixed up after round 1 (all compile-time moves).
#pragma unroll
        for (int m = 0; m < 16; ++m) x[(rnd << 4) | m] = y[m];
    }
    // now x[(rnd << 4) | (n3 << 1) | s] holds pass-1 coefficient (n4' = (rnd << 3) | n3, s); rename to (s << 4) | n4'
    u64 z[32];
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd)
#pragma unroll
        for (int n3 = 0; n3 < 8; ++n3)
#pragma unroll
            for (int s = 0; s < 2; ++s) z[(s << 4) | (rnd << 3) | n3] = x[(rnd << 4) | (n3 << 1) | s];
#pragma unroll
    for (int k = 0; k < 32; ++k) x[k] = z[k];
    XCHG_PRIO_DOWN();
}

// pass 1 regs -> pass 2 regs (split bit 5 = pass-1 passive bit = top bit of pass-2 nibble)
__device__ __forceinline__ void xchg_12(u64 (&x)[32], int t, u64 *lds) {
    XCHG_PRIO_UP();
    const int hi4 = t >> 5, lo5 = t & 31;   // as pass-1 thread
    const int hi8 = t >> 1, lo1 = t & 1;    // as pass-2 thread
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
#pragma unroll
        for (int n4 = 0; n4 < 16; ++n4) lds[img((hi4 << 9) | (n4 << 5) | lo5)] = x[(rnd << 4) | n4];
        __syncthreads();
        u64 y[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int n3 = m >> 1, s = m & 1;
            y[m] = lds[img((hi8 << 5) | (n3 << 2) | (s << 1) | lo1)];
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; ++m) x[(rnd << 4) | m] = y[m];
    }
    u64 z[32];
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd)
#pragma unroll
        for (int n3 = 0; n3 < 8; ++n3)
#pragma unroll
            for (int s = 0; s < 2; ++s) z[(s << 4) | (rnd << 3) | n3] = x[(rnd << 4) | (n3 << 1) | s];
#pragma unroll
    for (int k = 0; k < 32; ++k) x[k] = z[k];
    XCHG_PRIO_DOWN();
}

// pass 2 regs -> pass 3 regs (split bit 1 = pass-2 passive bit = top bit of pass-3 pair)
__device__ __forceinline__ void xchg_23(u64 (&x)[32], int t, u64 *lds) {
    XCHG_PRIO_UP();
    const int hi8 = t >> 1, lo1 = t & 1;  // as pass-2 thread; pass-3 thread z = t
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
#pragma unroll
        for (int n4 = 0; n4 < 16; ++n4) lds[img((hi8 << 5) | (n4 << 1) | lo1)] = x[(rnd << 4) | n4];
        __syncthreads();
        ulonglong2 y[8];
#pragma unroll
        for (int s3 = 0; s3 < 8; ++s3) y[s3] = *reinterpret_cast<const ulonglong2 *>(&lds[img((s3 << 10) | (t << 1))]);
        __syncthreads();
#pragma unroll
        for (int s3 = 0; s3 < 8; ++s3) { x[(rnd << 4) | (s3 << 1)] = y[s3].x; x[(rnd << 4) | (s3 << 1) | 1] = y[s3].y; }
    }
    // x[(rnd << 4) | (s3 << 1) | p0] -> pass-3 register (s3 << 2) | (rnd << 1) | p0
    u64 z[32];
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd)
#pragma unroll
        for (int s3 = 0; s3 < 8; ++s3)
#pragma unroll
            for (int p0 = 0; p0 < 2; ++p0) z[(s3 << 2) | (rnd << 1) | p0] = x[(rnd << 4) | (s3 << 1) | p0];
#pragma unroll
    for (int k = 0; k < 32; ++k) x[k] = z[k];
    XCHG_PRIO_DOWN();
}

// the three inverse-direction exchanges are the same moves backwards
__device__ __forceinline__ void xchg_32(u64 (&x)[32], int t, u64 *lds) {
    XCHG_PRIO_UP();
    const int hi8 = t >> 1, lo1 = t & 1;
    u64 z[32];
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
#pragma unroll
        for (int s3 = 0; s3 < 8; ++s3) {
            ulonglong2 v;
            v.x = x[(s3 << 2) | (rnd << 1)];
            v.y = x[(s3 << 2) | (rnd << 1) | 1];
            *reinterpret_cast<ulonglong2 *>(&lds[img((s3 << 10) | (t << 1))]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int n4 = 0; n4 < 16; ++n4) z[(rnd << 4) | n4] = lds[img((hi8 << 5) | (n4 << 1) | lo1)];
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) x[k] = z[k];  // pass-2 register (s = rnd) << 4 | n4
    XCHG_PRIO_DOWN();
}

__device__ __forceinline__ void xchg_21(u64 (&x)[32], int t, u64 *lds) {
    XCHG_PRIO_UP();
    const int hi4 = t >> 5, lo5 = t & 31;
    const int hi8 = t >> 1, lo1 = t & 1;
    u64 z[32];
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {  // pass-2 register (s << 4) | (rnd << 3) | n3
            const int n3 = m >> 1, s = m & 1;
            lds[img((hi8 << 5) | (n3 << 2) | (s << 1) | lo1)] = x[(s << 4) | (rnd << 3) | n3];
        }
        __syncthreads();
#pragma unroll
        for (int n4 = 0; n4 < 16; ++n4) z[(rnd << 4) | n4] = lds[img((hi4 << 9) | (n4 << 5) | lo5)];
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) x[k] = z[k];
    XCHG_PRIO_DOWN();
}

__device__ __forceinline__ void xchg_10(u64 (&x)[32], int t, u64 *lds) {
    XCHG_PRIO_UP();
    const int hi4 = t >> 5, lo5 = t & 31;
    u64 z[32];
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int n3 = m >> 1, s = m & 1;
            lds[img((hi4 << 9) | (n3 << 6) | (s << 5) | lo5)] = x[(s << 4) | (rnd << 3) | n3];
        }
        __syncthreads();
#pragma unroll
        for (int n4 = 0; n4 < 16; ++n4) z[(rnd << 4) | n4] = lds[img((n4 << 9) | t)];
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) x[k] = z[k];
    XCHG_PRIO_DOWN();
}

// Diagnostic build only (tools/ntt_lab2.hip -DNTT14_STAMPS): wave 0 of each workgroup keeps s_memtime values of its phase
// boundaries in SGPRs and writes them out when the transform is done (nothing is stored, and no vector-memory wait is added,
// inside the transform)
#ifdef NTT14_STAMPS
__device__ unsigned long long g_stamps[4096][16];
#define STAMP_DECL unsigned long long stamps_[10]
#define STAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamps_[i])::"memory")
#define STAMP_FLUSH()                                                                       \
    do {                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < 4096)                                          \
            for (int i_ = 0; i_ < 10; ++i_) g_stamps[blockIdx.x][i_] = stamps_[i_];        \
    } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH()
#endif

// ---------------------------------------------------------------------------------------------------------
// kernels: one workgroup = one (sub-)polynomial of 2^14 coefficients; sub s of polynomial s >> pb
// ---------------------------------------------------------------------------------------------------------
// The 14 layers are 16 units (arith.hpp): passes 0..2 are four units of 2 replicas (the passive bit) sharing twiddles, pass 3
// is two layers on 8 replicas with their own twiddles, taken 4 replicas at a time.  Unit u's butterflies run while unit
// u+1's twiddles are in flight (policies with PREFETCH; otherwise each unit fetches its own just in time), across the
// exchanges too: the first unit of a pass is fetched before the exchange that precedes it.
namespace n14 {
template <int l> using P0 = Unit<0, 4, l, 0, 2, 16, true>;
template <int l> using P1 = Unit<4, 4, l, 0, 2, 16, true>;
template <int l> using P2 = Unit<8, 4, l, 0, 2, 16, true>;
template <int l, int H> using P3 = Unit<12, 2, l, 4 * H, 4, 4, false, 9>;  // replica s3: block prefix (s3 << 9) | t

// is unit U's twiddle set fetched one unit ahead?  (policies bound the size of a prefetched set: registers)
template <class A, class U>
__device__ constexpr bool ahead() { return A::PREFETCH > 0 && U::NT <= A::PREFETCH; }

// forward step: (prefetch NEXT's twiddles | fetch CUR's), then CUR's butterflies
template <class A, class CUR, class NEXT>
__device__ __forceinline__ void fstep(u64 (&x)[32], typename A::TwRaw (&cur)[8], typename A::TwRaw (&next)[8], int top_cur, int top_next,
                                      const typename A::K &k) {
    FHE_SCHED_FENCE();
    if constexpr (!ahead<A, CUR>()) tw_load<A, false, CUR>(cur, top_cur, k);
    if constexpr (!std::is_void<NEXT>::value) {
        if constexpr (ahead<A, NEXT>()) tw_load<A, false, NEXT>(next, top_next, k);
    }
    ct_apply<A, CUR>(x, cur, k);
}
template <class A, class CUR, class NEXT>
__device__ __forceinline__ void istep(u64 (&x)[32], typename A::TwRaw (&cur)[8], typename A::TwRaw (&next)[8], int top_cur, int top_next,
                                      const typename A::K &k) {
    FHE_SCHED_FENCE();
    if constexpr (!ahead<A, CUR>()) tw_load<A, true, CUR>(cur, top_cur, k);
    if constexpr (!std::is_void<NEXT>::value) {
        if constexpr (ahead<A, NEXT>()) tw_load<A, true, NEXT>(next, top_next, k);
    }
    gs_apply<A, CUR>(x, cur, k);
}
}  // namespace n14

template <class A>
__device__ __forceinline__ void ntt14_fwd_body(u64 *__restrict__ g, const typename A::K &k, u64 *lds, const int t) {
    using namespace n14;
    u64 x[32];
    typename A::TwRaw ta[8], tb[8];
    const int t1 = t >> 5, t2 = t >> 1;  // block prefixes of passes 1 and 2; pass 3: (s3 << 9) | t
    STAMP_DECL;
    STAMP(0);
    if constexpr (ahead<A, P0<0>>()) tw_load<A, false, P0<0>>(ta, 0, k);
    // pass 0: layers 0..3
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = g[((r & 15) << 10) | ((r >> 4) << 9) | t];
    STAMP(1);
    fstep<A, P0<0>, P0<1>>(x, ta, tb, 0, 0, k);
    fstep<A, P0<1>, P0<2>>(x, tb, ta, 0, 0, k);
    fstep<A, P0<2>, P0<3>>(x, ta, tb, 0, 0, k);
    fstep<A, P0<3>, P1<0>>(x, tb, ta, 0, t1, k);
    STAMP(2);
    xchg_01(x, t, lds);
    STAMP(3);
    // pass 1: layers 4..7, block prefix = bits 13..10
    if constexpr (A::PASS_FOLD) {
#pragma unroll
        for (int r = 0; r < 32; ++r) x[r] = A::fold(x[r], k);
    }
    fstep<A, P1<0>, P1<1>>(x, ta, tb, t1, t1, k);
    fstep<A, P1<1>, P1<2>>(x, tb, ta, t1, t1, k);
    fstep<A, P1<2>, P1<3>>(x, ta, tb, t1, t1, k);
    fstep<A, P1<3>, P2<0>>(x, tb, ta, t1, t2, k);
    STAMP(4);
    xchg_12(x, t, lds);
    STAMP(5);
    // pass 2: layers 8..11, block prefix = bits 13..6
    if constexpr (A::PASS_FOLD) {
#pragma unroll
        for (int r = 0; r < 32; ++r) x[r] = A::fold(x[r], k);
    }
    fstep<A, P2<0>, P2<1>>(x, ta, tb, t2, t2, k);
    fstep<A, P2<1>, P2<2>>(x, tb, ta, t2, t2, k);
    fstep<A, P2<2>, P2<3>>(x, ta, tb, t2, t2, k);
    fstep<A, P2<3>, P3<0, 0>>(x, tb, ta, t2, t, k);
    STAMP(6);
    xchg_23(x, t, lds);
    STAMP(7);
    // pass 3: layers 12..13, block prefix = bits 13..2 = (s3 << 9) | t
    if constexpr (A::PASS_FOLD) {
#pragma unroll
        for (int r = 0; r < 32; ++r) x[r] = A::fold(x[r], k);
    }
    fstep<A, P3<0, 0>, P3<1, 0>>(x, ta, tb, t, t, k);
    fstep<A, P3<1, 0>, P3<0, 1>>(x, tb, ta, t, t, k);
    fstep<A, P3<0, 1>, P3<1, 1>>(x, ta, tb, t, t, k);
    fstep<A, P3<1, 1>, void>(x, tb, ta, t, t, k);
    STAMP(8);
#pragma unroll
    for (int s3 = 0; s3 < 8; ++s3) {
        ulonglong2 lo, hi;
        lo.x = A::canon_fwd(x[4 * s3 + 0], k); lo.y = A::canon_fwd(x[4 * s3 + 1], k);
        hi.x = A::canon_fwd(x[4 * s3 + 2], k); hi.y = A::canon_fwd(x[4 * s3 + 3], k);
        ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(g + ((s3 << 11) | (t << 2)));
        dst[0] = lo;
        dst[1] = hi;
    }
    STAMP(9);
    STAMP_FLUSH();
}

// PFX = false: a whole 2^14 ring (pb = 0): the block prefix and the table offset are compile-time zeros, which takes the
// variable shifts out of every twiddle index.  PFX = true: sub-transform sub & (2^pb - 1) of polynomial sub >> pb.
template <class A, bool PFX>
__global__ __launch_bounds__(N14_THREADS, 4) void ntt14_fwd_kernel(u64 *__restrict__ data, const ModDesc *__restrict__ descs,
                                                                    unsigned n_desc, unsigned subs, int pb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const unsigned sub = blockIdx.x;
    const unsigned poly = PFX ? sub >> pb : sub;
    const ModDesc &D = descs[n_desc == 1 ? 0 : poly % n_desc];
    const typename A::K k = A::make(D, 14, PFX ? pb : 0, PFX ? int(sub & ((1u << pb) - 1)) : 0);
    ntt14_fwd_body<A>(data + (size_t(sub) << 14), k, lds, threadIdx.x);
}

template <class A, bool PFX>
__device__ __forceinline__ void ntt14_inv_body(u64 *__restrict__ g, const typename A::K &k, u64 *lds, const int t) {
    using namespace n14;
    u64 x[32];
    typename A::TwRaw ta[8], tb[8];
    const int t1 = t >> 5, t2 = t >> 1;
    if constexpr (ahead<A, P3<1, 0>>()) tw_load<A, true, P3<1, 0>>(ta, t, k);
    // pass 3: layers 13..12
#pragma unroll
    for (int s3 = 0; s3 < 8; ++s3) {
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(g + ((s3 << 11) | (t << 2)));
        const ulonglong2 lo = src[0], hi = src[1];
        x[4 * s3 + 0] = lo.x; x[4 * s3 + 1] = lo.y; x[4 * s3 + 2] = hi.x; x[4 * s3 + 3] = hi.y;
    }
    istep<A, P3<1, 0>, P3<0, 0>>(x, ta, tb, t, t, k);
    istep<A, P3<0, 0>, P3<1, 1>>(x, tb, ta, t, t, k);
    istep<A, P3<1, 1>, P3<0, 1>>(x, ta, tb, t, t, k);
    istep<A, P3<0, 1>, P2<3>>(x, tb, ta, t, t2, k);
    xchg_32(x, t, lds);
    istep<A, P2<3>, P2<2>>(x, ta, tb, t2, t2, k);
    istep<A, P2<2>, P2<1>>(x, tb, ta, t2, t2, k);
    istep<A, P2<1>, P2<0>>(x, ta, tb, t2, t2, k);
    istep<A, P2<0>, P1<3>>(x, tb, ta, t2, t1, k);
    xchg_21(x, t, lds);
    istep<A, P1<3>, P1<2>>(x, ta, tb, t1, t1, k);
    istep<A, P1<2>, P1<1>>(x, tb, ta, t1, t1, k);
    istep<A, P1<1>, P1<0>>(x, ta, tb, t1, t1, k);
    istep<A, P1<0>, P0<3>>(x, tb, ta, t1, 0, k);
    xchg_10(x, t, lds);
    istep<A, P0<3>, P0<2>>(x, ta, tb, 0, 0, k);
    istep<A, P0<2>, P0<1>>(x, tb, ta, 0, 0, k);
    // the last layer leaves canonical values: a whole ring folds n^-1 into it (the difference branch multiplies by
    // twi[1] n^-1, only the sum branch needs a product of its own: 16 products fewer than layer + scaling), a sub-transform of a
    // larger ring is not scaled here at all
    istep<A, P0<1>, typename std::conditional<PFX, P0<0>, void>::type>(x, ta, tb, 0, 0, k);
    typename A::TwReg w{};
    if constexpr (PFX) {
        FHE_SCHED_FENCE();
        if constexpr (!ahead<A, P0<0>>()) tw_load<A, true, P0<0>>(tb, 0, k);
        w = A::prep(tb[0]);
    }
    // four butterflies at a time, stored as they finish (an unfenced block of 16 independent butterflies is scheduled
    // for maximum overlap and spills 100 registers)
    static_for<0, 4>([&](auto cc) {
        constexpr int base = (decltype(cc)::value & 1) * 4 + (decltype(cc)::value >> 1) * 16;
        FHE_SCHED_FENCE();
#pragma unroll
        for (int o = base; o < base + 4; ++o) {
            if constexpr (PFX) A::gs_last_plain(x[o], x[o + 8], w, k);
            else A::gs_last_scaled(x[o], x[o + 8], k);
            g[((o & 15) << 10) | ((o >> 4) << 9) | t] = x[o];
            g[(((o + 8) & 15) << 10) | ((o >> 4) << 9) | t] = x[o + 8];
        }
    });
}

template <class A, bool PFX>
__global__ __launch_bounds__(N14_THREADS, 4) void ntt14_inv_kernel(u64 *__restrict__ data, const ModDesc *__restrict__ descs,
                                                                    unsigned n_desc, unsigned subs, int pb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const unsigned sub = blockIdx.x;
    const unsigned poly = PFX ? sub >> pb : sub;
    const ModDesc &D = descs[n_desc == 1 ? 0 : poly % n_desc];
    const typename A::K k = A::make(D, 14, PFX ? pb : 0, PFX ? int(sub & ((1u << pb) - 1)) : 0);
    ntt14_inv_body<A, PFX>(data + (size_t(sub) << 14), k, lds, threadIdx.x);
}

}  // namespace fhe
