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
// The forward transform runs passes 0 -> 3 (Cooley-Tukey), the inverse 3 -> 0 (Gentleman-Sande, then * n^-1).
// Natural-order coefficients <-> bit-reversed evaluations, twiddle tw[2^layer + block], exactly as the reference.
//
// Arithmetic is a policy: ArithShoup (any prime < 2^62: Shoup multiplication + Harvey lazy reduction) or ArithPM
// (pseudo-Mersenne primes q = 2^b - c, c <= 2^(b-33), b <= 60 -- every prime `two_adic_primes(bits, log_n)` yields for
// the BASELINE configs: 7 full 32x32->64 multiply-adds per product on half-width twiddle limbs, no companion table, no
// compare/select anywhere in a butterfly).  Both end with canonical values in [0, q): bit-identical results.
#pragma once
#include "ntt_kernels.hpp"

namespace fhe {

// ---------------------------------------------------------------------------------------------------------
// arithmetic policies
// ---------------------------------------------------------------------------------------------------------
struct ArithShoup {
    struct K {
        u64 q, q2;
        const TwPair *tw, *twi;
        u64 ninv, ninv_s;
        int pb, prefix;
    };
    static __device__ __forceinline__ K make(const ModDesc &D, int log_n_total, int pb, int prefix) {
        return K{D.q, 2 * D.q, D.tw, D.twi, pb ? 1 : D.ninv[log_n_total], pb ? D.one_s : D.ninv_s[log_n_total], pb, prefix};
    }
    static __device__ __forceinline__ void ct(u64 &X, u64 &Y, int idx, const K &k) {
        const TwPair p = k.tw[idx];
        ct_bfly(X, Y, p.w, p.ws, k.q, k.q2);
    }
    template <int PH>
    static __device__ __forceinline__ void gs(u64 &X, u64 &Y, int idx, const K &k) {
        const TwPair p = k.twi[idx];
        gs_bfly(X, Y, p.w, p.ws, k.q, k.q2);
    }
    static constexpr bool GS_FOLDS = false;
    static constexpr bool JIT_TWIDDLES = true;
    static __device__ __forceinline__ u64 gs_fold(u64 x, const K &) { return x; }
    static __device__ __forceinline__ u64 fold(u64 x, const K &) { return x; }
    static __device__ __forceinline__ u64 canon_fwd(u64 x, const K &k) { return canon4(x, k.q, k.q2); }
    static __device__ __forceinline__ u64 finish_inv(u64 x, const K &k) { return csub(mul_shoup_lazy(x, k.ninv, k.ninv_s, k.q), k.q); }
};

// Pseudo-Mersenne product, q = 2^B - c (c < 2^(B-33), 34 <= B <= 60).
// On gfx950 every integer VALU instruction costs about the same issue time (~16 lanes/clk/SIMD; v_mul_lo/hi_u32 twice
// that), v_mad_u64_u32 included, so the product is arranged to need as FEW instructions as possible, all word aligned:
// the fixed operand w is split at B-31 bits, w = wl + wh 2^(B-31) (wh < 2^31), and stored with wlp = wl << (63-B) and
// wh2 = 2 wh, so that with y = y0 + y1 2^32 (any y < 2^63)
//     w y = z0 + z1 2^(B-31) + wh y1 2^(B+1),  z0 = wl y0,  z1 = wh y0 + wlp y1 < 2^64 (no carry),  2^(B+1) = 2c,
//     z1 2^(B-31) = (z1 >> 31) c + (z1 mod 2^31) 2^(B-31),
//     w y = v + u c (mod q),   v = z0 + (z1 mod 2^31) 2^(B-31),   u = wh2 y1 + (z1 >> 31) < 2^63 + 2^33,
// followed by one fold of the 96-bit v + u c at bit B.  8 multiply-adds + 5 other instructions, no compare/select, no
// companion table.  Result < 2^B + 2^(63 + 2k - B) <= 1.25 * 2^B   (k = bits of c).
struct PmTw {  // one twiddle, 16 bytes
    unsigned wl, wlp, wh, wh2;
};
struct PmK {
    u64 q, q2, q4;
    unsigned c;
};

template <int B>
__device__ __forceinline__ u64 pm_mul(u64 y, const PmTw w, const PmK &k) {
    constexpr unsigned M31 = 0x7fffffffu, HMASK = (1u << (B - 32)) - 1;
    const unsigned y0 = (unsigned)y, y1 = (unsigned)(y >> 32);
    const u64 z0 = (u64)w.wl * y0;
    const u64 z1 = (u64)w.wh * y0 + (u64)w.wlp * y1;
    const u64 v = (u64)((unsigned)z1 & M31) * (1u << (B - 31)) + z0;
    const u64 u = (u64)w.wh2 * y1 + (z1 >> 31);
    const u64 r1 = (u64)(unsigned)u * k.c + v;
    const u64 r2 = (u64)(unsigned)(u >> 32) * k.c + (r1 >> 32);  // v + u c = r2 * 2^32 + lo32(r1)
    const unsigned h2 = (unsigned)(r2 >> (B - 32));
    const u64 l2 = ((u64)((unsigned)r2 & HMASK) << 32) | (unsigned)r1;
    return (u64)h2 * k.c + l2;
}

// B = bit length of q (compile time: every shift and mask is an immediate)
template <int B>
struct ArithPM {
    static constexpr u64 MASK = (u64(1) << B) - 1;
    struct K {
        PmK m;
        const PmTw *tw, *twi;
        PmTw ninv;  // n^-1 (or 1) in twiddle form
        int pb, prefix;
    };
    static __host__ __device__ __forceinline__ PmTw split(u64 w) {
        PmTw t;
        t.wl = (unsigned)(w & ((u64(1) << (B - 31)) - 1));
        t.wlp = t.wl << (63 - B);
        t.wh = (unsigned)(w >> (B - 31));
        t.wh2 = t.wh << 1;
        return t;
    }
    static __device__ __forceinline__ K make(const ModDesc &D, int log_n_total, int pb, int prefix) {
        K k;
        k.m.q = D.q; k.m.q2 = 2 * D.q; k.m.q4 = 4 * D.q;
        k.m.c = D.pm_c;
        k.tw = reinterpret_cast<const PmTw *>(D.tww); k.twi = reinterpret_cast<const PmTw *>(D.twwi);
        k.ninv = split(pb ? 1 : D.ninv[log_n_total]);
        k.pb = pb; k.prefix = prefix;
        return k;
    }
    // x mod~ q: < 2^B + 2^(64-B) c
    static __device__ __forceinline__ u64 fold1(u64 x, const PmK &m) { return (x & MASK) + (u64)(unsigned)(x >> B) * m.c; }
    // Forward butterfly without any reduction.  Values grow by at most 2q per layer; a multiplicand must stay below 2^63
    // and a sum below 2^64, which holds for 4 layers after a fold (inputs < q + eps -> multiplicands < 7q, outputs < 9q).
    static __device__ __forceinline__ void ct(u64 &X, u64 &Y, int idx, const K &k) {
        const u64 t = pm_mul<B>(Y, k.tw[idx], k.m);
        const u64 x = X;
        X = x + t;
        Y = x - t + k.m.q2;
    }
    // Inverse butterflies come in pairs of layers: PH = 0 takes inputs < q + eps (outputs: sum < 2q + , product < q +),
    // PH = 1 takes those (outputs: sum < 4q +, product < q +); the sums of a PH = 1 layer are folded (gs_fold) before the
    // next pair.
    template <int PH>
    static __device__ __forceinline__ void gs(u64 &X, u64 &Y, int idx, const K &k) {
        const u64 s = X + Y;
        const u64 d = X - Y + (PH ? k.m.q4 : k.m.q2);
        X = s;
        Y = pm_mul<B>(d, k.twi[idx], k.m);
    }
    static constexpr bool GS_FOLDS = true;
    static constexpr bool JIT_TWIDDLES = true;
    static __device__ __forceinline__ u64 gs_fold(u64 x, const K &k) { return fold1(x, k.m); }
    static __device__ __forceinline__ u64 fold(u64 x, const K &k) { return fold1(x, k.m); }  // between forward passes
    static __device__ __forceinline__ u64 canon_fwd(u64 x, const K &k) { return csub(fold1(x, k.m), k.m.q); }
    static __device__ __forceinline__ u64 finish_inv(u64 x, const K &k) { return csub(pm_mul<B>(x, k.ninv, k.m), k.m.q); }
};

// ---------------------------------------------------------------------------------------------------------
// radix-2^R butterfly networks on x[OFF .. OFF + 2^R): layers L0 .. L0+R-1 of a (sub-)transform
// ---------------------------------------------------------------------------------------------------------
template <class A, int L0, int R, int OFF, int E>
__device__ __forceinline__ void ct_net(u64 (&x)[E], int top, const typename A::K &k) {
#pragma unroll
    for (int l = 0; l < R; ++l) {
        const int half = 1 << (R - 1 - l);
        // keep the twiddle loads of a layer inside that layer: hoisting all 2^R - 1 of them to the top of the pass costs
        // more registers than the kernel has (two workgroups per CU leave 128 VGPRs per thread)
        if (A::JIT_TWIDDLES && l > 0) asm volatile("" ::: "memory");
#pragma unroll
        for (int b = 0; b < (1 << l); ++b) {
            const int idx = (1 << (L0 + l + k.pb)) + ((((k.prefix << L0) | top)) << l) + b;
#pragma unroll
            for (int j = 0; j < half; ++j) A::ct(x[OFF + b * 2 * half + j], x[OFF + b * 2 * half + j + half], idx, k);
        }
    }
}

template <class A, int L0, int R, int OFF, int E>
__device__ __forceinline__ void gs_net(u64 (&x)[E], int top, const typename A::K &k) {
    static_for<0, R>([&](auto step_c) {
        constexpr int step = decltype(step_c)::value;  // 0 .. R-1, layer l = R-1-step
        constexpr int l = R - 1 - step;
        constexpr int PH = step & 1;
        constexpr int half = 1 << (R - 1 - l);
        if (A::JIT_TWIDDLES && step > 0) asm volatile("" ::: "memory");
#pragma unroll
        for (int b = 0; b < (1 << l); ++b) {
            const int idx = (1 << (L0 + l + k.pb)) + ((((k.prefix << L0) | top)) << l) + b;
#pragma unroll
            for (int j = 0; j < half; ++j) {
                A::template gs<PH>(x[OFF + b * 2 * half + j], x[OFF + b * 2 * half + j + half], idx, k);
                if constexpr (A::GS_FOLDS && PH == 1) x[OFF + b * 2 * half + j] = A::gs_fold(x[OFF + b * 2 * half + j], k);
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------------------
// half-image exchanges.  The image holds the 8192 coefficients whose split bit equals `round`; j = index with the
// split bit removed; 2 pad slots per 32 keep 16-byte alignment and make every access pattern below conflict free.
// ---------------------------------------------------------------------------------------------------------
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
}

// pass 1 regs -> pass 2 regs (split bit 5 = pass-1 passive bit = top bit of pass-2 nibble)
__device__ __forceinline__ void xchg_12(u64 (&x)[32], int t, u64 *lds) {
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
}

// pass 2 regs -> pass 3 regs (split bit 1 = pass-2 passive bit = top bit of pass-3 pair)
__device__ __forceinline__ void xchg_23(u64 (&x)[32], int t, u64 *lds) {
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
}

// the three inverse-direction exchanges are the same moves backwards
__device__ __forceinline__ void xchg_32(u64 (&x)[32], int t, u64 *lds) {
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
}

__device__ __forceinline__ void xchg_21(u64 (&x)[32], int t, u64 *lds) {
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
}

__device__ __forceinline__ void xchg_10(u64 (&x)[32], int t, u64 *lds) {
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
}

// ---------------------------------------------------------------------------------------------------------
// kernels: one workgroup = one (sub-)polynomial of 2^14 coefficients; sub s of polynomial s >> pb
// ---------------------------------------------------------------------------------------------------------
template <class A>
__global__ __launch_bounds__(N14_THREADS, 4) void ntt14_fwd_kernel(u64 *__restrict__ data, const ModDesc *__restrict__ descs,
                                                                    unsigned n_desc, unsigned subs, int pb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const int t = threadIdx.x;
    const unsigned sub = blockIdx.x;
    const ModDesc &D = descs[(sub >> pb) % n_desc];
    const typename A::K k = A::make(D, 14, pb, int(sub & ((1u << pb) - 1)));
    u64 *g = data + (size_t(sub) << 14);
    u64 x[32];
    // pass 0: layers 0..3
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = g[((r & 15) << 10) | ((r >> 4) << 9) | t];
    ct_net<A, 0, 4, 0, 32>(x, 0, k);
    ct_net<A, 0, 4, 16, 32>(x, 0, k);
    xchg_01(x, t, lds);
    // pass 1: layers 4..7, block prefix = bits 13..10
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = A::fold(x[r], k);
    ct_net<A, 4, 4, 0, 32>(x, t >> 5, k);
    ct_net<A, 4, 4, 16, 32>(x, t >> 5, k);
    xchg_12(x, t, lds);
    // pass 2: layers 8..11, block prefix = bits 13..6
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = A::fold(x[r], k);
    ct_net<A, 8, 4, 0, 32>(x, t >> 1, k);
    ct_net<A, 8, 4, 16, 32>(x, t >> 1, k);
    xchg_23(x, t, lds);
    // pass 3: layers 12..13, block prefix = bits 13..2 = (s3 << 9) | t
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = A::fold(x[r], k);
    static_for<0, 8>([&](auto s3c) {
        constexpr int s3 = decltype(s3c)::value;
        ct_net<A, 12, 2, 4 * s3, 32>(x, (s3 << 9) | t, k);
    });
#pragma unroll
    for (int s3 = 0; s3 < 8; ++s3) {
        ulonglong2 lo, hi;
        lo.x = A::canon_fwd(x[4 * s3 + 0], k); lo.y = A::canon_fwd(x[4 * s3 + 1], k);
        hi.x = A::canon_fwd(x[4 * s3 + 2], k); hi.y = A::canon_fwd(x[4 * s3 + 3], k);
        ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(g + ((s3 << 11) | (t << 2)));
        dst[0] = lo;
        dst[1] = hi;
    }
}

template <class A>
__global__ __launch_bounds__(N14_THREADS, 4) void ntt14_inv_kernel(u64 *__restrict__ data, const ModDesc *__restrict__ descs,
                                                                    unsigned n_desc, unsigned subs, int pb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const int t = threadIdx.x;
    const unsigned sub = blockIdx.x;
    const ModDesc &D = descs[(sub >> pb) % n_desc];
    const typename A::K k = A::make(D, 14, pb, int(sub & ((1u << pb) - 1)));
    u64 *g = data + (size_t(sub) << 14);
    u64 x[32];
    // pass 3: layers 13..12
#pragma unroll
    for (int s3 = 0; s3 < 8; ++s3) {
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(g + ((s3 << 11) | (t << 2)));
        const ulonglong2 lo = src[0], hi = src[1];
        x[4 * s3 + 0] = lo.x; x[4 * s3 + 1] = lo.y; x[4 * s3 + 2] = hi.x; x[4 * s3 + 3] = hi.y;
    }
    static_for<0, 8>([&](auto s3c) {
        constexpr int s3 = decltype(s3c)::value;
        gs_net<A, 12, 2, 4 * s3, 32>(x, (s3 << 9) | t, k);
    });
    xchg_32(x, t, lds);
    gs_net<A, 8, 4, 0, 32>(x, t >> 1, k);
    gs_net<A, 8, 4, 16, 32>(x, t >> 1, k);
    xchg_21(x, t, lds);
    gs_net<A, 4, 4, 0, 32>(x, t >> 5, k);
    gs_net<A, 4, 4, 16, 32>(x, t >> 5, k);
    xchg_10(x, t, lds);
    gs_net<A, 0, 4, 0, 32>(x, 0, k);
    gs_net<A, 0, 4, 16, 32>(x, 0, k);
#pragma unroll
    for (int r = 0; r < 32; ++r) g[((r & 15) << 10) | ((r >> 4) << 9) | t] = A::finish_inv(x[r], k);
}

}  // namespace fhe
