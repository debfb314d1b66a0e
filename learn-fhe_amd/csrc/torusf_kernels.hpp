// Row T the way the reference computes it: the torus products of a CMUX through an f64 complex FFT (util/src/ring/fft/c64.rs:11-108:
// fold N reals into N/2 complex values, transform, multiply pointwise, transform back, round each f64 to u64 mod 2^64).  The default
// path of this backend is EXACT (torus_kernels.hpp, torus30_kernels.hpp); this one is the opt-in `fft64` mode of a prepared key
// (fhe_tggsw_prepare_fft64): the same team-per-ciphertext kernels with ONE pass of half-size complex transforms where the exact path
// runs three passes of full-size 30-bit ones.  Its results are NOT bit-identical to anything -- neither to the exact product nor to
// the reference, whose own low bits depend on libm's `cis` -- and are accepted the way the reference accepts its own product:
// |result - exact| <= 2^(64 + log_b + log_n - 53) per product (c64.rs:186-208), decode-level equality of everything built on it.
//
// The transform.  For a real polynomial a of degree < N and zeta = exp(i pi / N), b_k = a_k + i a_{k+N/2} is the left output of layer
// 0 of the size-N negacyclic transform (twiddle zeta^(N/2) = i), and the rest of that transform restricted to the left half is a
// size-N/2 Cooley-Tukey network over C with the twiddles T[2^l + i] = cis(pi (2 bitrev_l(i) + 1) / 2^(l+1)) of the left sub-tree:
// exactly what the integer kernels call a sub-transform with pb = 1, prefix = 0 (arith.hpp tw_load).  So the generic pass machinery
// (ntt_kernels.hpp fwd_run / inv_run) runs unchanged on the policy below, no separate twist (c64.rs:19-29 `to_c64_twisted` multiplies
// by zeta^k first and then runs a cyclic transform: the same evaluations, one complex product per coefficient more).  The left half
// holds the evaluations at zeta^(4j+1); the other half would hold their conjugates.
#pragma once
#include "torus_kernels.hpp"

namespace fhe {

struct ArithC64 {
    typedef double2 Elem;   // re, im
    typedef double2 TwRaw;
    typedef double2 TwReg;
    static constexpr int PREFETCH = 0;
    static constexpr bool GS_FOLDS = false;
    static constexpr int CT_LAYERS = 64;
    static constexpr bool PASS_FOLD = false;
    static constexpr int GS_SPAN = 0;
    static __device__ constexpr bool ct_fold_at(int) { return false; }
    // T[j], j < 2 M, staged in LDS by the kernel (stage_twiddles): from the third pass on every lane needs its own entries -- nine 16-byte
    // fetches per transform that were vector loads from the table in HBM/L2 (a third of the wave cycles parked); an LDS read has a
    // quarter of their latency and the wave-uniform entries of the first passes come as broadcasts
    typedef const __attribute__((address_space(3))) double2 *LdsTw;
    struct K {
        LdsTw tw;
        double scale;                 // 1 / (N / 2)
        int pb, prefix;               // 1, 0: the left sub-tree
    };
    static __device__ __forceinline__ K make(const double2 *tw_lds, int log_m) { return K{(LdsTw)tw_lds, 1.0 / double(1 << log_m), 1, 0}; }
    template <bool INV>
    static __device__ __forceinline__ TwRaw fetch(const K &k, int idx) {
        const LdsTw p = k.tw + idx;
        TwRaw r;
        r.x = p->x; r.y = p->y;
        return r;
    }
    static __device__ __forceinline__ TwReg prep(const TwRaw &r) { return r; }
    // (X, Y) <- (X + w Y, X - w Y) in six fused multiply-adds: the sum as two chains, the difference as 2 X - (X + w Y)
    static __device__ __forceinline__ void ct(Elem &X, Elem &Y, const TwReg &w, const K &) {
        const double sr = fma(-w.y, Y.y, fma(w.x, Y.x, X.x)), si = fma(w.y, Y.x, fma(w.x, Y.y, X.y));
        Y.x = fma(2.0, X.x, -sr); Y.y = fma(2.0, X.y, -si);
        X.x = sr; X.y = si;
    }
    // (X, Y) <- (X + Y, (X - Y) conj(w)): |w| = 1, the inverse twiddle is the conjugate (c64.rs:117)
    template <int PH>
    static __device__ __forceinline__ void gs(Elem &X, Elem &Y, const TwReg &w, const K &) {
        const double dr = X.x - Y.x, di = X.y - Y.y;
        X.x += Y.x; X.y += Y.y;
        Y.x = fma(di, w.y, dr * w.x);
        Y.y = fma(-dr, w.y, di * w.x);
    }
    static __device__ __forceinline__ Elem gs_fold(Elem x, const K &) { return x; }
    static __device__ __forceinline__ Elem fold(Elem x, const K &) { return x; }
    static __device__ __forceinline__ Elem canon_fwd(Elem x, const K &) { return x; }
    static __device__ __forceinline__ Elem finish_inv(Elem x, const K &k) { return Elem{x.x * k.scale, x.y * k.scale}; }
};

// util/src/ring/fft/c64.rs:69-85 `f64_mod_u64`: the integer nearest to v (ties away from zero), mod 2^64
__device__ __forceinline__ u64 f64_mod_u64(double v) {
    const u64 bits = (u64)__double_as_longlong(v);
    const int exponent = int((bits >> 52) & 0x7ff);
    const u64 mantissa = (bits << 11) | 0x8000000000000000ull;
    const int shift = 1086 - exponent;
    u64 value = 0;
    if (shift >= -63 && shift <= 0) value = mantissa << (-shift);
    else if (shift >= 1 && shift <= 64) value = ((mantissa >> (shift - 1)) + 1) >> 1;
    return (bits >> 63) ? 0 - value : value;
}

// W = WaveRing over the M = N / 2 complex elements of a polynomial.  A lane holds the torus coefficients of its complex slots as
// c[e] = coefficient k_e, c[E + e] = coefficient k_e + M (k_e = coef_index<W>(lane, e)): the fold needs no exchange.
template <class W>
struct TorusF {
    static constexpr int M = W::N, N = 2 * W::N, E = W::E;
    static constexpr int LDS_WORDS = 2 * W::PN;  // 8-byte words per team: the exchange image (PN complex slots) = the rotation image (N u64 + padding)
    static constexpr int TW_WORDS = 2 * N;       // the block's copy of T[0 .. 2 M): N entries of 16 bytes, behind the teams' images
    static constexpr size_t LDS_BYTES = size_t(LDS_WORDS * W::TEAMS + TW_WORDS) * 8;
};

// every thread of the block: copy the twiddle tree into LDS; returns the table (a workgroup barrier inside)
template <class W>
__device__ __forceinline__ const double2 *stage_twiddles(const double2 *__restrict__ tw, unsigned char *smem_raw) {
    double2 *dst = reinterpret_cast<double2 *>(reinterpret_cast<u64 *>(smem_raw) + TorusF<W>::LDS_WORDS * W::TEAMS);
    for (int i = threadIdx.x; i < TorusF<W>::N; i += W::THREADS) dst[i] = tw[i];
    __syncthreads();
    return dst;
}

template <class W>
__device__ __forceinline__ void pairs_load(u64 (&c)[2 * W::E], const u64 *__restrict__ g, int lane) {
#pragma unroll
    for (int e = 0; e < W::E; ++e) {
        const int k = coef_index<W>(lane, e);
        c[e] = g[k]; c[W::E + e] = g[k + W::N];
    }
}
template <class W>
__device__ __forceinline__ void pairs_store(const u64 (&c)[2 * W::E], u64 *__restrict__ g, int lane) {
#pragma unroll
    for (int e = 0; e < W::E; ++e) {
        const int k = coef_index<W>(lane, e);
        g[k] = c[e]; g[k + W::N] = c[W::E + e];
    }
}

// c <- c * X^r (ring.rs:299-313 on T64), r in [0, 2N), through the team's image
template <class W>
__device__ __forceinline__ void pairs_rotate(u64 (&c)[2 * W::E], unsigned r, int lane, u64 *img) {
    constexpr int E = W::E, M = W::N, N = 2 * M;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const unsigned pos = (unsigned(coef_index<W>(lane, e) + h * M) + r) & (2 * N - 1);
            img[lds_phys(pos & (N - 1))] = pos < N ? c[h * E + e] : 0 - c[h * E + e];
        }
    exchange_sync<W::WAVE>();
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < E; ++e) c[h * E + e] = img[lds_phys(coef_index<W>(lane, e) + h * M)];
    exchange_sync<W::WAVE>();
}

// key rows: [row][a | b][M] complex evaluations; evaluation lane * E + r of a row sits at slot r * TEAM + lane: a team's registers
// load as coalesced 16-byte accesses
template <class W>
struct KeyRowF {
    double2 a[W::E], b[W::E];
};
template <class W>
__device__ __forceinline__ void load_rowf(KeyRowF<W> &kr, const double2 *__restrict__ row, int lane) {
#pragma unroll
    for (int r = 0; r < W::E; ++r) { kr.a[r] = row[r * W::TEAM + lane]; kr.b[r] = row[W::N + r * W::TEAM + lane]; }
}
__device__ __forceinline__ void cmac(double2 &s, const double2 &x, const double2 &k) {
    s.x = fma(x.x, k.x, fma(-x.y, k.y, s.x));
    s.y = fma(x.x, k.y, fma(x.y, k.x, s.y));
}

// The decomposition state of a gadget of at most 31 bits (log_b d <= 31: cfg5's 21, the reference's 23) fits one dword once the
// rounding shift is done: the digit recurrence of decompose.rs:124-134 on 32-bit words -- half the instructions of the u64 form, and the
// signed digit converts with one v_cvt_f64_i32.  Wider gadgets keep the u64 state.
template <bool NARROW> struct DigitState;
template <> struct DigitState<true> {
    typedef unsigned T;
    static __device__ __forceinline__ T init(u64 v, const TDecomp &P) { return (unsigned)((v + P.rnd) >> P.rb); }
    static __device__ __forceinline__ double next(T &c, const TDecomp &P) {
        const unsigned limb = c & (unsigned)P.mask;
        c >>= P.log_b;
        const unsigned carry = (((limb - 1) | c) & limb) >> (P.log_b - 1);
        c += carry;
        return (double)(int)(limb - (carry << P.log_b));
    }
};
template <> struct DigitState<false> {
    typedef u64 T;
    static __device__ __forceinline__ T init(u64 v, const TDecomp &P) { return tdecomp_init(v, P); }
    static __device__ __forceinline__ double next(T &c, const TDecomp &P) { return (double)(long long)tdecomp_next(c, P); }  // exact (c64.rs:25 `to_i64() as f64`)
};

// (sa, sb) <- sum over the 2d limbs of (da, db) of FFT(limb) (.) (row.a, row.b), transformed back: the two halves of the external
// product (tggsw.rs:100-112, k = 1) as complex slot values, coefficient layout, before rounding
template <class W, bool NARROW>
__device__ __forceinline__ void teamf_gadget(const u64 (&da)[2 * W::E], const u64 (&db)[2 * W::E], const double2 *__restrict__ rows, const TDecomp &P,
                                             int lane, double2 *lds, const ArithC64::K &k, double2 (&sa)[W::E], double2 (&sb)[W::E]) {
    using A = ArithC64;
    using D = DigitState<NARROW>;
    constexpr int E = W::E;
    typename D::T st[2 * E];
#pragma unroll
    for (int e = 0; e < 2 * E; ++e) st[e] = D::init(da[e], P);
#pragma unroll
    for (int e = 0; e < E; ++e) sa[e] = sb[e] = double2{0.0, 0.0};
    KeyRowF<W> kr;
    load_rowf<W>(kr, rows, lane);
#pragma unroll 1
    for (int j = 0; j < 2 * P.d; ++j) {
        if (j == P.d) {
#pragma unroll
            for (int e = 0; e < 2 * E; ++e) st[e] = D::init(db[e], P);
        }
        double2 x[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            x[e].x = D::next(st[e], P);
            x[e].y = D::next(st[E + e], P);
        }
        fwd_run<A, typename W::C, W::LOG_N, W::LOG_E, 0, true, W::WAVE>(x, lane, nullptr, lds, true, k);
#pragma unroll
        for (int e = 0; e < E; ++e) { cmac(sa[e], x[e], kr.a[e]); cmac(sb[e], x[e], kr.b[e]); }
        if (j + 1 < 2 * P.d) load_rowf<W>(kr, rows + size_t(j + 1) * 2 * W::N, lane);
    }
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
        inv_run<A, typename W::C, W::LOG_N, W::LOG_E, W::LOG_N, true, W::WAVE>(sa, lane, nullptr, lds, true, k);
#pragma unroll
        for (int e = 0; e < E; ++e) { const double2 t = sa[e]; sa[e] = sb[e]; sb[e] = t; }
    }
}

// r = 0xffffffff: plain external product, (ca, cb) <- key (.) (ca, cb); else one CMUX step of the blind rotation (bootstrapping.rs:91-95)
template <class W>
__device__ __forceinline__ void teamf_cmux(u64 (&ca)[2 * W::E], u64 (&cb)[2 * W::E], unsigned r, const double2 *__restrict__ rows, const TDecomp &P,
                                           const ArithC64::K &k, int lane, u64 *lds64) {
    constexpr int E = W::E;
    const bool plain = r == 0xffffffffu;
    if (r == 0) return;  // team-uniform
    u64 da[2 * E], db[2 * E];
#pragma unroll
    for (int e = 0; e < 2 * E; ++e) { da[e] = ca[e]; db[e] = cb[e]; }
    if (!plain) {
        pairs_rotate<W>(da, r, lane, lds64);
        pairs_rotate<W>(db, r, lane, lds64);
#pragma unroll
        for (int e = 0; e < 2 * E; ++e) { da[e] -= ca[e]; db[e] -= cb[e]; }
    }
    double2 sa[E], sb[E];
    if (P.rb >= 33) teamf_gadget<W, true>(da, db, rows, P, lane, reinterpret_cast<double2 *>(lds64), k, sa, sb);  // kernel-uniform
    else teamf_gadget<W, false>(da, db, rows, P, lane, reinterpret_cast<double2 *>(lds64), k, sa, sb);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const u64 a0 = f64_mod_u64(sa[e].x), a1 = f64_mod_u64(sa[e].y), b0 = f64_mod_u64(sb[e].x), b1 = f64_mod_u64(sb[e].y);
        if (plain) { ca[e] = a0; ca[E + e] = a1; cb[e] = b0; cb[E + e] = b1; }
        else { ca[e] += a0; ca[E + e] += a1; cb[e] += b0; cb[E + e] += b1; }
    }
}

template <class W, int MIN_WAVES>
__global__ __launch_bounds__(W::THREADS, MIN_WAVES) void torusf_cmux_kernel(u64 *__restrict__ acc_a, u64 *__restrict__ acc_b, unsigned batch,
                                                                            const double2 *__restrict__ rows, TDecomp P, const u64 *__restrict__ rot,
                                                                            size_t rot_stride, const double2 *__restrict__ tw) {
    constexpr int E = W::E, N = 2 * W::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const unsigned ct = blockIdx.x * W::TEAMS + team;
    const ArithC64::K k = ArithC64::make(stage_twiddles<W>(tw, smem_raw), W::LOG_N);
    if (ct >= batch) return;
    u64 *lds = reinterpret_cast<u64 *>(smem_raw) + team * TorusF<W>::LDS_WORDS;
    u64 *ga = acc_a + size_t(ct) * N, *gb = acc_b + size_t(ct) * N;
    u64 ca[2 * E], cb[2 * E];
    pairs_load<W>(ca, ga, lane);
    pairs_load<W>(cb, gb, lane);
    teamf_cmux<W>(ca, cb, rot ? unsigned(rot[size_t(ct) * rot_stride]) & (2 * N - 1) : 0xffffffffu, rows, P, k, lane, lds);
    pairs_store<W>(ca, ga, lane);
    pairs_store<W>(cb, gb, lane);
}

// bootstrapping.rs:84-96 in one launch, as torus30_blind_rotate_kernel; rows: [n_lwe][2d][2][M] complex
template <class W, int MIN_WAVES>
__global__ __launch_bounds__(W::THREADS, MIN_WAVES) void torusf_blind_rotate_kernel(const u64 *__restrict__ v, const u64 *__restrict__ a_tilde,
                                                                                    const u64 *__restrict__ b_tilde, unsigned n_lwe, unsigned batch,
                                                                                    const double2 *__restrict__ rows, TDecomp P,
                                                                                    const double2 *__restrict__ tw, u64 *__restrict__ out_a,
                                                                                    u64 *__restrict__ out_b) {
    constexpr int E = W::E, N = 2 * W::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const unsigned ct = blockIdx.x * W::TEAMS + team;
    const ArithC64::K k = ArithC64::make(stage_twiddles<W>(tw, smem_raw), W::LOG_N);
    if (ct >= batch) return;
    u64 *lds = reinterpret_cast<u64 *>(smem_raw) + team * TorusF<W>::LDS_WORDS;
    u64 ca[2 * E], cb[2 * E];
    pairs_load<W>(cb, v, lane);
#pragma unroll
    for (int e = 0; e < 2 * E; ++e) ca[e] = 0;
    pairs_rotate<W>(cb, (2 * N - (unsigned(b_tilde[ct]) & (2 * N - 1))) & (2 * N - 1), lane, lds);
    const size_t per = size_t(2 * P.d) * 2 * W::N;
    const u64 *a = a_tilde + size_t(ct) * n_lwe;
#pragma unroll 1
    for (unsigned i = 0; i < n_lwe; ++i) {
        const unsigned r = __builtin_amdgcn_readfirstlane(unsigned(a[i]) & (2 * N - 1));
        teamf_cmux<W>(ca, cb, r, rows + i * per, P, k, lane, lds);
    }
    pairs_store<W>(ca, out_a + size_t(ct) * N, lane);
    pairs_store<W>(cb, out_b + size_t(ct) * N, lane);
}

// Measured and dropped (round 3): PAIRED transforms -- an element type of two complex values (limb p of the a half with limb p of the b half,
// the two outputs on the way back) shares every twiddle fetch, exchange and barrier between two transforms: four trips through the network
// per CMUX instead of eight, 228 registers -- 72.8 k gates/s at cfg5 against 74.2-75.0 k unpaired: barriers are not what the kernel waits for.
// Measured and dropped: THREE waves per SIMD (accumulator parked in LDS during the gadget, 168 registers with 20 spilled, six ciphertexts
// per CU): 76.6 k at batch 1536, 70.7 k at 4096 (77.8 k with two waves), 58.1 k at 1024 (two thirds of a generation).
// Measured and dropped (round 3): one wave per POLYNOMIAL for N <= 1024 -- wave 0 owns the a half of the ciphertext, wave 1 the b half, each
// runs its d forward transforms alone (eight slots per lane, three passes, wave-private exchanges, no barrier), one hand-over of partial
// sums per CMUX, accumulator half parked in LDS (246 registers, no spill): 68.9 k gates/s at cfg5 against 74.2 k for the team form above
// (17.3 k against 26.5 k at batch 256: one ciphertext per block leaves a CU two waves) -- half the LDS traffic and two barriers per CMUX
// instead of fifty, but eight-slot transforms with two waves per SIMD have nothing to hide their dependency chains behind.

// ---- EXACT products through the same transforms: key words cut into three signed pieces --------------------------------------------
// A key word k (signed 64 bits) = k0 + k1 2^22 + k2 2^43 with |k0| <= 2^21, |k1|, |k2| <= 2^20.  The product of a digit polynomial
// (|digit| <= 2^(log_b - 1)) with ONE piece has coefficients below N 2^(log_b + 20): integers that an f64 transform of this size
// reproduces with an error far below 1/2, so rounding recovers them EXACTLY, and the three rounded products recombine to the
// exact product mod 2^64 with shifts and wrapping adds.  Error bound (Percival 2003 for FFT convolutions: |err| <= ||x||_2 ||y||_2
// ((1 + eps)^(3 n) (1 + eps sqrt 5)^(3 n + 1) (1 + beta)^(3 n) - 1), eps = 2^-53, n = log2 of the transform size, beta <= 2 eps the
// twiddle error: <= 160 eps for n <= 10): ||digits||_2 <= sqrt(N) 2^(log_b - 1), ||piece||_2 <= sqrt(N) 2^21, 2d products summed:
// err <= 2d N 2^(log_b + 20) 160 2^-53.  The host takes this path only while 2d N 2^log_b <= 2^21 (err <= 0.04 at the limit: a factor
// 12 inside 1/2; cfg5: 2d N 2^log_b = 2^19.6, err <= 0.015; the six-FMA butterfly forms its difference as 2X - (X + wY), at most twice a
// plain butterfly's rounding error: inside those factors): the result is the exact oracle's, bit for bit, and the whole-gate and
// extreme-operand tests of tests/test_torus_gpu.py hold it to that.  One CMUX = the digits once (bytes, parked in LDS), then per
// output (a, b): 2d forward transforms, 3 x 2d multiply-accumulates per slot, three inverse transforms, cheap roundings (|value| <
// 2^40: the magic-constant trick) and the recombination: 12 + 6 half-size complex transforms where the three-prime path runs 24
// full-size 30-bit ones.  Measured and dropped: both outputs in ONE pass over the limbs (six sums per slot, the accumulator parked in
// LDS, twiddles from HBM for want of LDS: 6 + 6 transforms, 192 registers) -- 32.2 k gates/s at cfg5 against 34.1 k for the two passes
// (25.4 k against 31.2 k at batch 4096): six key streams per slot consumed the moment they are requested cost more than six transforms.
// Also measured and dropped (on the prefetching form, 46.3 k): every limb transformed ONCE and kept in registers (2d x E complex values, 238
// registers, no spill; 2d + 6 transforms), the two output passes then loads and multiply-accumulates only -- 34.6 k: with no transform to fly
// under, the key rows are waited for in full; and the hybrid (the a output as here, under the transforms, their evaluations kept for a b output of
// loads and products only: 256 registers, 14 spilled): 47.6 k at batch 1024 but 38.8 k at 4096 and 13.3 k at 256 (46.1 k / 18.8 k here).
template <class W>
struct TorusX3 {
    static constexpr int M = W::N, N = 2 * W::N, E = W::E;
    static constexpr int IMG_WORDS = 2 * W::PN;
    static constexpr int DIG_WORDS_PER_LIMB = W::TEAM;          // 2 dwords per lane and limb (E <= 4 digits each) = TEAM 8-byte words
    // per team: exchange / rotation image | one half of the accumulator (N words, lane-private slots: whichever half the running output
    // pass does not need) | the digits
    static __host__ __device__ constexpr int lds_words(int limbs) { return IMG_WORDS + N + limbs * DIG_WORDS_PER_LIMB; }
    static __host__ __device__ constexpr size_t lds_bytes(int limbs) { return size_t(lds_words(limbs) * W::TEAMS + 2 * N) * 8; }  // + the twiddle tree
    static_assert(W::E <= 4, "digits of a lane's E slots are packed into one dword");
};

__device__ __forceinline__ void key_pieces(u64 k, double (&p)[3]) {
    const long long v = (long long)k;
    const long long p0 = (long long)((u64)v << 42) >> 42;   // low 22 bits, sign extended
    const long long rem = (v - p0) >> 22;                    // exact
    const long long p1 = (long long)((u64)rem << 43) >> 43;  // low 21 bits, sign extended
    const long long p2 = (rem - p1) >> 21;                   // |p2| <= 2^20
    p[0] = (double)p0; p[1] = (double)p1; p[2] = (double)p2;
}

// |v| < 2^50, v within 1/2 of an integer: that integer, through the mantissa of v + 1.5 2^52
__device__ __forceinline__ long long round_small(double v) {
    const long long bits = __double_as_longlong(v + 6755399441055744.0);
    return (bits & 0xfffffffffffffll) - (1ll << 51);
}

// the 2d x 2E digits of one CMUX, computed once: dig[(limb * 2 + h) * TEAM + lane] packs the E digits of the lane's slots (h = 0:
// coefficients k_e, h = 1: coefficients k_e + M) as signed bytes
template <class W>
__device__ __forceinline__ void park_digits_x3(const u64 (&da)[2 * W::E], const u64 (&db)[2 * W::E], const TDecomp &P, int lane, unsigned *dig) {
    constexpr int E = W::E;
    const unsigned mask = (unsigned)P.mask;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        unsigned st[2 * E];  // log_b d <= 31 on this path: the state fits a dword after the rounding shift
#pragma unroll
        for (int e = 0; e < 2 * E; ++e) st[e] = (unsigned)(((half ? db[e] : da[e]) + P.rnd) >> P.rb);
#pragma unroll 1
        for (int j = 0; j < P.d; ++j) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned w = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    unsigned &c = st[h * E + e];
                    const unsigned limb = c & mask;
                    c >>= P.log_b;
                    const unsigned carry = (((limb - 1) | c) & limb) >> (P.log_b - 1);
                    c += carry;
                    w |= ((limb - (carry << P.log_b)) & 0xffu) << (8 * e);
                }
                dig[((half * P.d + j) * 2 + h) * W::TEAM + lane] = w;
            }
        }
    }
}

// one output (o = 0: a, 1: b) of the external product from the parked digits: out[e], out[E + e] = the exact coefficients mod 2^64
template <class W>
__device__ __forceinline__ void x3_output(const unsigned *dig, const double2 *__restrict__ rows, int d2, int o, int lane, double2 *lds, const ArithC64::K &k,
                                          u64 (&out)[2 * W::E]) {
    using A = ArithC64;
    constexpr int E = W::E, M = W::N;
    double2 s0[E], s1[E], s2[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s0[e] = s1[e] = s2[e] = double2{0.0, 0.0};
    const double2 *row = rows + size_t(o) * 3 * M;  // rows: [limb][a | b][piece][M]
#pragma unroll 1
    for (int j = 0; j < d2; ++j) {
        const unsigned wl = dig[(j * 2 + 0) * W::TEAM + lane], wh = dig[(j * 2 + 1) * W::TEAM + lane];
        double2 x[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            x[e].x = (double)((int)(wl << (24 - 8 * e)) >> 24);
            x[e].y = (double)((int)(wh << (24 - 8 * e)) >> 24);
        }
        // the limb's key rows are requested BEFORE its transform and consumed after it: the fetch (L2 / Infinity Cache) flies under the
        // butterflies instead of stalling the wave in front of the multiply-accumulates
        double2 k0[E], k1[E], k2[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int slot = e * W::TEAM + lane;
            k0[e] = row[slot]; k1[e] = row[M + slot]; k2[e] = row[2 * M + slot];
        }
        fwd_run<A, typename W::C, W::LOG_N, W::LOG_E, 0, true, W::WAVE>(x, lane, nullptr, lds, true, k);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            cmac(s0[e], x[e], k0[e]);
            cmac(s1[e], x[e], k1[e]);
            cmac(s2[e], x[e], k2[e]);
        }
        row += 6 * M;
    }
    inv_run<A, typename W::C, W::LOG_N, W::LOG_E, W::LOG_N, true, W::WAVE>(s0, lane, nullptr, lds, true, k);
    inv_run<A, typename W::C, W::LOG_N, W::LOG_E, W::LOG_N, true, W::WAVE>(s1, lane, nullptr, lds, true, k);
    inv_run<A, typename W::C, W::LOG_N, W::LOG_E, W::LOG_N, true, W::WAVE>(s2, lane, nullptr, lds, true, k);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        out[e] = (u64)round_small(s0[e].x) + ((u64)round_small(s1[e].x) << 22) + ((u64)round_small(s2[e].x) << 43);
        out[E + e] = (u64)round_small(s0[e].y) + ((u64)round_small(s1[e].y) << 22) + ((u64)round_small(s2[e].y) << 43);
    }
}

// r = 0xffffffff: plain external product; else one CMUX step (r in [1, 2N)).  lds64: the team's image, then its digit area
template <class W>
__device__ __forceinline__ void teamx3_cmux(u64 (&ca)[2 * W::E], u64 (&cb)[2 * W::E], unsigned r, const double2 *__restrict__ rows, const TDecomp &P,
                                            const ArithC64::K &k, int lane, u64 *lds64) {
    constexpr int E = W::E;
    const bool plain = r == 0xffffffffu;
    if (r == 0) return;  // team-uniform
    u64 *park = lds64 + TorusX3<W>::IMG_WORDS;
    unsigned *dig = reinterpret_cast<unsigned *>(park + TorusX3<W>::N);
    {
        u64 da[2 * E], db[2 * E];
#pragma unroll
        for (int e = 0; e < 2 * E; ++e) { da[e] = ca[e]; db[e] = cb[e]; }
        if (!plain) {
            pairs_rotate<W>(da, r, lane, lds64);
            pairs_rotate<W>(db, r, lane, lds64);
#pragma unroll
            for (int e = 0; e < 2 * E; ++e) { da[e] -= ca[e]; db[e] -= cb[e]; }
        }
        park_digits_x3<W>(da, db, P, lane, dig);
    }
    // each output pass runs with only the half it adds to in registers: the other half waits in the team's LDS slots (the prefetched
    // key rows need the room)
    u64 x[2 * E];
#pragma unroll
    for (int e = 0; e < 2 * E; ++e) park[e * W::TEAM + lane] = cb[e];
    x3_output<W>(dig, rows, 2 * P.d, 0, lane, reinterpret_cast<double2 *>(lds64), k, x);
#pragma unroll
    for (int e = 0; e < 2 * E; ++e) {
        ca[e] = plain ? x[e] : ca[e] + x[e];
        cb[e] = park[e * W::TEAM + lane];
        park[e * W::TEAM + lane] = ca[e];
    }
    x3_output<W>(dig, rows, 2 * P.d, 1, lane, reinterpret_cast<double2 *>(lds64), k, x);
#pragma unroll
    for (int e = 0; e < 2 * E; ++e) {
        cb[e] = plain ? x[e] : cb[e] + x[e];
        ca[e] = park[e * W::TEAM + lane];
    }
}

template <class W>
__device__ __forceinline__ const double2 *stage_twiddles_x3(const double2 *__restrict__ tw, unsigned char *smem_raw, int limbs) {
    double2 *dst = reinterpret_cast<double2 *>(reinterpret_cast<u64 *>(smem_raw) + TorusX3<W>::lds_words(limbs) * W::TEAMS);
    for (int i = threadIdx.x; i < TorusX3<W>::N; i += W::THREADS) dst[i] = tw[i];
    __syncthreads();
    return dst;
}

template <class W, int MIN_WAVES>
__global__ __launch_bounds__(W::THREADS, MIN_WAVES) void torusx3_cmux_kernel(u64 *__restrict__ acc_a, u64 *__restrict__ acc_b, unsigned batch,
                                                                             const double2 *__restrict__ rows, TDecomp P, const u64 *__restrict__ rot,
                                                                             size_t rot_stride, const double2 *__restrict__ tw) {
    constexpr int E = W::E, N = 2 * W::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const unsigned ct = blockIdx.x * W::TEAMS + team;
    const ArithC64::K k = ArithC64::make(stage_twiddles_x3<W>(tw, smem_raw, 2 * P.d), W::LOG_N);
    if (ct >= batch) return;
    u64 *lds = reinterpret_cast<u64 *>(smem_raw) + team * TorusX3<W>::lds_words(2 * P.d);
    u64 *ga = acc_a + size_t(ct) * N, *gb = acc_b + size_t(ct) * N;
    u64 ca[2 * E], cb[2 * E];
    pairs_load<W>(ca, ga, lane);
    pairs_load<W>(cb, gb, lane);
    teamx3_cmux<W>(ca, cb, rot ? unsigned(rot[size_t(ct) * rot_stride]) & (2 * N - 1) : 0xffffffffu, rows, P, k, lane, lds);
    pairs_store<W>(ca, ga, lane);
    pairs_store<W>(cb, gb, lane);
}

// bootstrapping.rs:84-96 in one launch; rows: [n_lwe][2d][2][3][M] complex
template <class W, int MIN_WAVES>
__global__ __launch_bounds__(W::THREADS, MIN_WAVES) void torusx3_blind_rotate_kernel(const u64 *__restrict__ v, const u64 *__restrict__ a_tilde,
                                                                                     const u64 *__restrict__ b_tilde, unsigned n_lwe, unsigned batch,
                                                                                     const double2 *__restrict__ rows, TDecomp P,
                                                                                     const double2 *__restrict__ tw, u64 *__restrict__ out_a,
                                                                                     u64 *__restrict__ out_b) {
    constexpr int E = W::E, N = 2 * W::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const unsigned ct = blockIdx.x * W::TEAMS + team;
    const ArithC64::K k = ArithC64::make(stage_twiddles_x3<W>(tw, smem_raw, 2 * P.d), W::LOG_N);
    if (ct >= batch) return;
    u64 *lds = reinterpret_cast<u64 *>(smem_raw) + team * TorusX3<W>::lds_words(2 * P.d);
    u64 ca[2 * E], cb[2 * E];
    pairs_load<W>(cb, v, lane);
#pragma unroll
    for (int e = 0; e < 2 * E; ++e) ca[e] = 0;
    pairs_rotate<W>(cb, (2 * N - (unsigned(b_tilde[ct]) & (2 * N - 1))) & (2 * N - 1), lane, lds);
    const size_t per = size_t(2 * P.d) * 6 * W::N;
    const u64 *a = a_tilde + size_t(ct) * n_lwe;
#pragma unroll 1
    for (unsigned i = 0; i < n_lwe; ++i) {
        const unsigned r = __builtin_amdgcn_readfirstlane(unsigned(a[i]) & (2 * N - 1));
        teamx3_cmux<W>(ca, cb, r, rows + i * per, P, k, lane, lds);
    }
    pairs_store<W>(ca, out_a + size_t(ct) * N, lane);
    pairs_store<W>(cb, out_b + size_t(ct) * N, lane);
}

// key preparation for the three-piece form: job = (row, a | b, piece) -> [row][a | b][piece][M] complex evaluations
template <class W>
__global__ __launch_bounds__(W::THREADS) void torusx3_key_prepare_kernel(const u64 *__restrict__ rows_a, const u64 *__restrict__ rows_b, size_t n_rows,
                                                                         const double2 *__restrict__ tw, double2 *__restrict__ out) {
    constexpr int E = W::E, M = W::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const size_t job = size_t(blockIdx.x) * W::TEAMS + team;
    const ArithC64::K k = ArithC64::make(stage_twiddles<W>(tw, smem_raw), W::LOG_N);
    if (job >= 6 * n_rows) return;
    double2 *lds = reinterpret_cast<double2 *>(reinterpret_cast<u64 *>(smem_raw) + team * TorusF<W>::LDS_WORDS);
    const size_t row = job / 6;
    const int o = int((job / 3) & 1), piece = int(job % 3);
    const u64 *src = (o ? rows_b : rows_a) + row * (2 * M);
    double2 x[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int c = coef_index<W>(lane, e);
        double plo[3], phi[3];
        key_pieces(src[c], plo);
        key_pieces(src[c + M], phi);
        x[e].x = piece == 0 ? plo[0] : (piece == 1 ? plo[1] : plo[2]);
        x[e].y = piece == 0 ? phi[0] : (piece == 1 ? phi[1] : phi[2]);
    }
    fwd_run<ArithC64, typename W::C, W::LOG_N, W::LOG_E, 0, true, W::WAVE>(x, lane, nullptr, lds, true, k);
    double2 *dst = out + job * M;
#pragma unroll
    for (int e = 0; e < E; ++e) dst[e * W::TEAM + lane] = x[e];
}

// key preparation: signed torus rows [rows][N] (a | b) -> complex evaluations [row][a | b][M] in the KeyRowF layout
// (c64.rs:19-29: key words converted with `to_i64() as f64`, round to nearest)
template <class W>
__global__ __launch_bounds__(W::THREADS) void torusf_key_prepare_kernel(const u64 *__restrict__ rows_a, const u64 *__restrict__ rows_b, size_t n_rows,
                                                                        const double2 *__restrict__ tw, double2 *__restrict__ out) {
    constexpr int E = W::E, M = W::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const size_t job = size_t(blockIdx.x) * W::TEAMS + team;  // (row, a|b)
    const ArithC64::K k = ArithC64::make(stage_twiddles<W>(tw, smem_raw), W::LOG_N);
    if (job >= 2 * n_rows) return;
    double2 *lds = reinterpret_cast<double2 *>(reinterpret_cast<u64 *>(smem_raw) + team * TorusF<W>::LDS_WORDS);
    const u64 *src = ((job & 1) ? rows_b : rows_a) + (job >> 1) * (2 * M);
    double2 x[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int c = coef_index<W>(lane, e);
        x[e].x = (double)(long long)src[c];
        x[e].y = (double)(long long)src[c + M];
    }
    fwd_run<ArithC64, typename W::C, W::LOG_N, W::LOG_E, 0, true, W::WAVE>(x, lane, nullptr, lds, true, k);
    double2 *dst = out + job * M;
#pragma unroll
    for (int e = 0; e < E; ++e) dst[e * W::TEAM + lane] = x[e];
}

}  // namespace fhe
