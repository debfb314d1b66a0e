// FHEW-side kernels: gadget decomposition, automorphism, monomial
// multiply, and the fused RLWE x RGSW external product / RLWE key switch / LMKCDEY blind rotation.
//
// Fused kernels: ONE wave64 owns one RLWE ciphertext.  The accumulator (a, b) lives in registers across the
// whole chain; every decomposed limb is transformed by a wave-private NTT (registers + a wave-private LDS
// image, no workgroup barrier anywhere), multiplied into evaluation-domain sums against key rows that were
// transformed once at key-preparation time, and only 2 inverse transforms per step bring the result back:
// 2d+2 transforms per external product instead of the reference's 12d (scheme/fhew/src/rgsw.rs:116-128
// over util/src/ring/fft/zq.rs:14-19), d+2 instead of 6d per key switch (rlwe.rs:177-186).  All sums are
// exact mod q, so the coefficient-domain outputs are bit-identical to the reference's.
#pragma once
#include <type_traits>
#include "ntt_kernels.hpp"

namespace fhe {

// ---- gadget decomposition (util/src/misc/decompose.rs:49-64, 91-112) ---------------------------
struct DecompParams {
    u64 q;
    u64 rnd;      // ((1 << rounding_bits) >> 1) % q
    u64 neg_b;    // q - 2^log_b
    u64 mask;     // 2^log_b - 1
    u64 b_by_2;   // 2^(log_b - 1)
    int log_b, d, rb;
};

// rounding_shr + to_center_u64: the running two's-complement state the digits are peeled from
__device__ __forceinline__ u64 decomp_init(u64 v, const DecompParams &P) {
    u64 r = csub(v + P.rnd, P.q) >> P.rb;      // Zq + u64, then >> bits (decompose.rs:92-95)
    return r < (P.q >> 1) ? r : r - P.q;       // zq.rs:83-89 (wrapping)
}

__device__ __forceinline__ u64 decomp_next(u64 &c, const DecompParams &P) {
    const u64 limb = c & P.mask;
    const u64 carry = (limb + (c & 1)) > P.b_by_2;
    c = (c >> P.log_b) + carry;
    return limb + (carry ? P.neg_b : 0);       // < q because limb < 2^log_b
}

// in: [polys][n]   out: [polys][d][n] (digit-major per polynomial, least significant first)
FHE_HEADER_KERNEL void decompose_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t n, size_t polys, DecompParams P) {
    const size_t total = n * polys;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n, i = idx - p * n;
        u64 c = decomp_init(in[idx], P);
        for (int j = 0; j < P.d; ++j) out[(p * P.d + j) * n + i] = decomp_next(c, P);
    }
}

// ---- automorphism X -> X^t (util/src/avec.rs:34-50) and monomial multiply (util/src/ring.rs:299-313) ----
FHE_HEADER_KERNEL void automorphism_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, unsigned n, size_t batch, unsigned t, u64 q) {
    const size_t total = size_t(n) * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n;
        const unsigned i = unsigned(idx - p * n);
        const unsigned it = unsigned((u64(i) * t) & (2 * n - 1));
        const u64 v = in[idx];
        if (it < n) out[p * n + it] = v;
        else out[p * n + it - n] = v ? q - v : 0;
    }
}

FHE_HEADER_KERNEL void monomial_mul_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, unsigned n, size_t batch, unsigned k2n, u64 q) {
    const size_t total = size_t(n) * batch;
    const unsigned r = k2n & (n - 1);
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n;
        const unsigned j = unsigned(idx - p * n);
        const unsigned dst = (j + r) & (n - 1);
        const bool neg = k2n < n ? dst < k2n : dst >= k2n - n;
        const u64 v = in[idx];
        out[p * n + dst] = neg ? (v ? q - v : 0) : v;
    }
}

// ---- LWE side of the FHEW gate: scheme/fhew/src/lwe.rs:90-99, 151-160 ----------
// util/src/zq.rs:128-130 `mod_switch` and 132-140 `mod_switch_odd`: f64 arithmetic, f64::round = half away from zero,
// `as i64` / `as u64` saturating casts, rem_euclid.  IEEE double multiply and divide, no contraction.
__device__ __forceinline__ u64 zq_mod_switch(u64 v, u64 q, u64 q_prime, bool odd) {
    const double x = __ddiv_rn(__dmul_rn((double)v, (double)q_prime), (double)q);
    if (!odd) {
        const double r = round(x);
        const long long i = r >= 9.2233720368547758e18 ? 0x7fffffffffffffffll : (long long)r;  // x >= 0 here
        return (u64)i % q_prime;
    }
    const double u = floor(x);
    if (u == 0.0) return (u64)round(x) % q_prime;
    return ((u64)u | 1ull) % q_prime;
}

FHE_HEADER_KERNEL void lwe_mod_switch_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t count, u64 q, u64 q_prime, int odd) {
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < count; idx += size_t(gridDim.x) * blockDim.x)
        out[idx] = zq_mod_switch(in[idx], q, q_prime, odd != 0);
}

// scheme/fhew/src/lwe.rs:151-160 `Lwe::key_switch`: one thread per (ciphertext, output coefficient); column n_out carries b.
// ksk_a [d * n_in][n_out], ksk_b [d * n_in]; row j * n_in + i = digit j of input coefficient i (`decompose(a).flatten()`).
// q < 2^32 (the FHEW key-switching moduli are 2^16 .. 2^20), so products fit 64 bits.
FHE_HEADER_KERNEL void lwe_key_switch_kernel(const u64 *__restrict__ ct_a, const u64 *__restrict__ ct_b, unsigned n_in, unsigned n_out,
                                             size_t batch, const u64 *__restrict__ ksk_a, const u64 *__restrict__ ksk_b, DecompParams P,
                                             u64 *__restrict__ out_a, u64 *__restrict__ out_b) {
    const size_t total = size_t(n_out + 1) * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / (n_out + 1);
        const unsigned col = unsigned(idx - p * (n_out + 1));
        u64 acc = 0;
        for (unsigned i = 0; i < n_in; ++i) {
            u64 c = decomp_init(ct_a[p * n_in + i], P);
            for (int j = 0; j < P.d; ++j) {
                const u64 dg = decomp_next(c, P);
                const size_t row = size_t(j) * n_in + i;
                const u64 kv = col < n_out ? ksk_a[row * n_out + col] : ksk_b[row];
                acc = (acc + kv * dg % P.q) % P.q;
            }
        }
        if (col < n_out) out_a[p * n_out + col] = acc;
        else out_b[p] = (acc + ct_b[p]) % P.q;
    }
}

// scheme/fhew/src/lwe.rs:22-75 (Add/Sub/Neg/double on LweCiphertext): out = sum_k coef[k] * in[k] + addend (mod q), q < 2^62.
// |coef| is 1 or 2 for the gates of scheme/fhew/src/fhew.rs:61-69, so the scalar product is a short double-and-add.
struct LinComb {
    const u64 *in[4];
    long long coef[4];
    int k;
};
FHE_HEADER_KERNEL void lwe_lincomb_kernel(LinComb lc, u64 q, u64 addend, u64 *__restrict__ out, size_t count) {
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < count; idx += size_t(gridDim.x) * blockDim.x) {
        u64 acc = addend;
        for (int t = 0; t < lc.k; ++t) {
            const long long c = lc.coef[t];
            u64 m = c < 0 ? 0ull - (u64)c : (u64)c;
            u64 x = lc.in[t][idx], term = 0;
            while (m) {
                if (m & 1) term = csub(term + x, q);
                x = csub(x + x, q);
                m >>= 1;
            }
            if (c < 0 && term) term = q - term;
            acc = csub(acc + term, q);
        }
        out[idx] = acc;
    }
}

// scheme/fhew/src/rlwe.rs:193-202 `Rlwe::sample_extract(ct, i)`; `addend` is added to b (Fhew::op's + Q/8, fhew.rs:39)
FHE_HEADER_KERNEL void rlwe_sample_extract_kernel(const u64 *__restrict__ ct_a, const u64 *__restrict__ ct_b, unsigned n, size_t batch, unsigned i,
                                                  u64 q, u64 addend, u64 *__restrict__ out_a, u64 *__restrict__ out_b) {
    const size_t total = size_t(n) * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n;
        const unsigned j = unsigned(idx - p * n);
        u64 v;
        if (j <= i) v = ct_a[p * n + (i - j)];
        else { v = ct_a[p * n + (n + i - j)]; v = v ? q - v : 0; }
        out_a[idx] = v;
        if (j == 0) out_b[p] = csub(ct_b[p * n + i] + addend, q);
    }
}

// ---- team-owned ring element ---------------------------------------------------------------------
// One ciphertext belongs to a TEAM of threads that keeps it in registers across a whole chain of gadget products.
// N <= 512: the team is ONE wave (E = N/64 coefficients per lane, wave-private LDS image, no workgroup barrier anywhere,
// four teams per block).  N = 1024, 2048: a single wave would need 16 / 32 coefficients per lane for each of the accumulator
// pair, the evaluation-domain sums and the transform -- 256 VGPRs plus spills, ONE wave per SIMD, nothing to hide an LDS
// or key-load latency behind (measured at cfg3: the SIMD issues 52 % of the time).  There the team is 2 / 4 waves with 8
// coefficients per lane, one team per block, exchanges behind a workgroup barrier: half the registers, 3-4 waves per SIMD.
// LOG_E_ (coefficients per lane) defaults to the throughput shape; the FHEW entry points also instantiate LOG_E_ = 2 for
// N >= 1024 (4 / 8 waves per ciphertext): more parallelism per ciphertext, +57 % at batch 1 and 64, -2 % at batch 1024.
template <int LOG_N_, int LOG_E_ = (LOG_N_ <= 9 ? LOG_N_ - 6 : 3)>
struct WaveRing {
    static constexpr int LOG_N = LOG_N_;
    static constexpr int LOG_T = LOG_N_ - LOG_E_;  // log2 threads per ciphertext
    static constexpr int TEAM = 1 << LOG_T;
    static constexpr bool WAVE = LOG_T == 6;                   // exchanges need no workgroup barrier
    static constexpr int TEAMS = WAVE ? 4 : 1;                 // ciphertexts per block
    static constexpr int THREADS = TEAM * TEAMS;
    static constexpr int LOG_E = LOG_E_;
    static constexpr int E = 1 << LOG_E;
    static constexpr int N = 1 << LOG_N;
    using C = NttCfg<LOG_N, LOG_E, 1>;  // T = TEAM
    static constexpr int R0 = C::R0;
    static constexpr int PN = C::PN;
    // N = 1024 at 4 coefficients per lane: the FHEW kernels park the accumulator's b half in LDS across the gadget loop (one team per
    // block: N words behind the exchange image) and are compiled for FOUR waves per SIMD -- see wave_gadget_product
    static constexpr bool PARK = LOG_E_ == 2 && LOG_N_ == 10;
    static constexpr size_t LDS_BYTES = size_t(PN + (PARK ? N : 0)) * 8 * TEAMS;
    static constexpr int TORUS_LDS_WORDS = PN + 2 * N;  // torus kernels: exchange image + parking area for one residue pair
    static constexpr size_t TORUS_LDS_BYTES = size_t(TORUS_LDS_WORDS) * 8 * TEAMS;
    // 30-bit torus blind rotation with the digits of a CMUX parked as bytes (torus30_kernels.hpp): + [limbs][E / 4][TEAM] dwords
    static __host__ __device__ constexpr int torus_dig_words(int limbs) { return E >= 4 ? limbs * (E / 4) * TEAM / 2 : 0; }  // in 8-byte words
    static __host__ __device__ constexpr size_t torus_pk_lds_bytes(int limbs) { return size_t(TORUS_LDS_WORDS + torus_dig_words(limbs)) * 8 * TEAMS; }
#ifndef FHE_TEAM_OCC
#define FHE_TEAM_OCC 2
#endif
    // multi-wave teams: waves per SIMD the kernels are compiled for (bounds the VGPRs; HIP's second launch bound).  Measured
    // at cfg3 / cfg5 (tools/scripts/occ_sweep.sh): 2 -> 47.5k blind rotations/s, 13.1k TFHE gates/s; 3 -> 39.4k / 10.0k;
    // 4 -> 37.4k / 7.7k: below ~170 VGPRs the accumulator pair, the sums and the digit state spill.
    static constexpr int MIN_WAVES = WAVE ? 1 : (LOG_E <= 2 ? 3 : FHE_TEAM_OCC);
    static_assert(LOG_N >= 7 && LOG_N <= 11, "fused FHEW kernels cover N = 128 .. 2048");
    static_assert(C::T == TEAM, "team size");
    static __device__ __forceinline__ int lane() { return threadIdx.x & (TEAM - 1); }  // thread within its team
    static __device__ __forceinline__ int team() { return threadIdx.x >> LOG_T; }      // team within the block
};

// launch bound of the two FHEW kernels: four waves per SIMD where the parked form fits them without a spill (the two-operand
// policies; the Shoup products' temporaries do not: 10 registers spilled), W::MIN_WAVES otherwise (and in the torus kernels)
template <class A, class W>
constexpr int fhew_min_waves() { return (W::PARK && !std::is_same<A, ArithShoup>::value) ? 4 : W::MIN_WAVES; }

// polynomial index held by register k of `lane` in the coefficient (first-pass) layout
template <class W>
__device__ __forceinline__ int coef_index(int lane, int k) {
    const int gg = k >> W::R0, r = k & ((1 << W::R0) - 1);
    return pass_index<W::LOG_N, 0, W::R0>(lane + W::TEAM * gg, r);
}

// key rows are stored so that the evaluation-layout registers (evaluation lane*E + r in x[r]) load as
// coalesced 16-byte pairs: word offset of evaluation e = lane*E + r inside a row
template <class W>
__host__ __device__ __forceinline__ int key_perm(int e) {
    constexpr int E = W::E, TEAM = W::TEAM;
    const int lane = e / E, r = e % E;
    return (r >> 1) * (2 * TEAM) + lane * 2 + (r & 1);
}

// what a fused kernel needs besides its arithmetic policy's constants (policy K built from the ModDesc in-kernel)
struct RingConsts {
    const ModDesc *desc;  // the modulus, in HBM
    Barrett B;            // variable x variable products of the multiply-accumulate
};

// One key row (a and b components of one limb) as a lane holds it: E / 2 16-byte loads each.  The row of limb j + 1 is requested
// as soon as the multiply-accumulate of limb j has consumed its registers, so the fetch (the bootstrapping keys exceed the L2:
// rows come from the Infinity Cache) flies under the next limb's decomposition and forward transform instead of stalling the
// wave in front of every multiply-accumulate (44 % of the wave cycles of a blind rotation were parked in s_waitcnt).
template <class W>
struct KeyRow {
    ulonglong2 a[W::E / 2], b[W::E / 2];
};
template <class W>
__device__ __forceinline__ void load_row(KeyRow<W> &kr, const u64 *__restrict__ row, int lane) {
    const ulonglong2 *ka = reinterpret_cast<const ulonglong2 *>(row);
    const ulonglong2 *kb = reinterpret_cast<const ulonglong2 *>(row + (1 << W::LOG_N));
#pragma unroll
    for (int r2 = 0; r2 < W::E / 2; ++r2) { kr.a[r2] = ka[r2 * W::TEAM + lane]; kr.b[r2] = kb[r2 * W::TEAM + lane]; }
}
// sums[0][r] += x[r] * keyA[r], sums[1][r] += x[r] * keyB[r]  for one limb; x arrives lazy in [0, 4q)
template <class A, class W>
__device__ __forceinline__ void mac_row(const u64 (&x)[W::E], typename A::MacAcc (&sa)[W::E], typename A::MacAcc (&sb)[W::E],
                                        const KeyRow<W> &kr, int term, const RingConsts &K, const typename A::K &k) {
#pragma unroll
    for (int r2 = 0; r2 < W::E / 2; ++r2) {
        const ulonglong2 a = kr.a[r2], b = kr.b[r2];
        const u64 x0 = A::mac_in(x[2 * r2], k), x1 = A::mac_in(x[2 * r2 + 1], k);
        sa[2 * r2] = A::mac(sa[2 * r2], x0, a.x, term, k, K.B);
        sa[2 * r2 + 1] = A::mac(sa[2 * r2 + 1], x1, a.y, term, k, K.B);
        sb[2 * r2] = A::mac(sb[2 * r2], x0, b.x, term, k, K.B);
        sb[2 * r2 + 1] = A::mac(sb[2 * r2 + 1], x1, b.y, term, k, K.B);
    }
}

// Gadget product shared by the external product and the key switch (one transform instance each way):
//   both = true : scheme/fhew/src/rgsw.rs:116-128   limbs = decompose(a) ++ decompose(b), rows = 2d,
//                 (a, b) <- (sum_j rows[j].a * limb_j, sum_j rows[j].b * limb_j)
//   both = false: scheme/fhew/src/rlwe.rs:177-186   limbs = decompose(a), rows = d,
//                 (a, b) <- (sum_j rows[j].a * limb_j, sum_j rows[j].b * limb_j + b)
// Each limb is transformed by the wave-private NTT and multiplied into evaluation-domain sums; two inverse
// transforms bring the result back.  (ca, cb): coefficient layout, canonical, in and out.
template <class A, class W>
__device__ __forceinline__ void wave_gadget_product(u64 (&ca)[W::E], u64 (&cb)[W::E],
                                                    const u64 *__restrict__ rows, const DecompParams &P, bool both, int lane,
                                                    u64 *lds, const RingConsts &K, const typename A::K &k) {
    constexpr int E = W::E;
    typename A::MacAcc ma[E], mb[E];
    u64 sa[E], sb[E], st[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { ma[e] = mb[e] = A::mac_zero(); st[e] = decomp_init(ca[e], P); }
    const int total = both ? 2 * P.d : P.d;
    // W::PARK (N = 1024, 4 coefficients per lane): four ciphertexts per CU need 128 registers.  Of the 148 the kernel took, 33 were
    // loop-invariant: the b half of the accumulator (needed again only at limb d and at the end: parked in LDS) and exchange
    // addresses of BOTH transform directions hoisted out of the limb loop.  Recomputing all of them per limb fits 128 registers but
    // costs more instructions than the fourth ciphertext brings (59 k against 66 k blind rotations/s at batch 1024); recomputing
    // only the inverse's (two inverse transforms per gadget product against 2d forward ones: the lane index is made opaque in front
    // of them) fits 128 registers with no spill: 72.5 k -> 75.4 k at batch 4096, 58.5 k -> 64.5 k at 1024.
    constexpr bool PARK = W::PARK;
    u64 *park = lds + W::PN;
    if constexpr (PARK) {
#pragma unroll
        for (int e = 0; e < E; ++e) park[e * W::TEAM + lane] = cb[e];
    }
    KeyRow<W> kr;
    load_row<W>(kr, rows, lane);
#pragma unroll 1
    for (int j = 0; j < total; ++j) {
        if (both && j == P.d) {
#pragma unroll
            for (int e = 0; e < E; ++e) st[e] = decomp_init(PARK ? park[e * W::TEAM + lane] : cb[e], P);
        }
        u64 x[E];
#pragma unroll
        for (int e = 0; e < E; ++e) x[e] = decomp_next(st[e], P);
        fwd_run<A, typename W::C, W::LOG_N, W::LOG_E, 0, true, W::WAVE>(x, lane, nullptr, lds, true, k);
        mac_row<A, W>(x, ma, mb, kr, j, K, k);
        if (j + 1 < total) load_row<W>(kr, rows + size_t(j + 1) * 2 * W::N, lane);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) { sa[e] = A::mac_finish(ma[e], k); sb[e] = A::mac_finish(mb[e], k); }
    // two inverse transforms through ONE instance: transform sa, swap, transform again
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
        int ln = lane;
        if constexpr (PARK) asm volatile("" : "+v"(ln));
        inv_run<A, typename W::C, W::LOG_N, W::LOG_E, W::LOG_N, true, W::WAVE>(sa, ln, nullptr, lds, true, k);
#pragma unroll
        for (int e = 0; e < E; ++e) { const u64 t = sa[e]; sa[e] = sb[e]; sb[e] = t; }
    }
    // after two swaps sa / sb are back in place
#pragma unroll
    for (int e = 0; e < E; ++e) {
        ca[e] = sa[e];
        cb[e] = both ? sb[e] : csub(sb[e] + (PARK ? park[e * W::TEAM + lane] : cb[e]), K.B.q);
    }
}

// util/src/avec.rs:34-50 on a register-resident polynomial: scatter through the wave's LDS image
template <class W>
__device__ __forceinline__ void wave_automorphism(u64 (&c)[W::E], unsigned t, int lane, u64 *lds, u64 q) {
    constexpr int E = W::E, N = W::N;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const unsigned i = coef_index<W>(lane, k);
        const unsigned it = (i * t) & (2 * N - 1);
        const u64 v = c[k];
        lds[lds_phys(it & (N - 1))] = it < N ? v : (v ? q - v : 0);
    }
    exchange_sync<W::WAVE>();
#pragma unroll
    for (int k = 0; k < E; ++k) c[k] = lds[lds_phys(coef_index<W>(lane, k))];
    exchange_sync<W::WAVE>();
}

template <class W>
__device__ __forceinline__ void wave_load(u64 (&c)[W::E], const u64 *__restrict__ g, int lane) {
#pragma unroll
    for (int k = 0; k < (W::E); ++k) c[k] = g[coef_index<W>(lane, k)];
}
template <class W>
__device__ __forceinline__ void wave_store(const u64 (&c)[W::E], u64 *__restrict__ g, int lane) {
#pragma unroll
    for (int k = 0; k < (W::E); ++k) g[coef_index<W>(lane, k)] = c[k];
}


struct FhewKey {      // device view of a prepared gadget key set
    const u64 *rows;  // [count][rows_per_ct][2][N] evaluation domain, key_perm layout
    int rows_per_ct;  // 2d (RGSW) or d (key-switching key)
    DecompParams P;
};

// batched gadget product, every ciphertext against key entry `index`:
//   both = 1: RLWE x RGSW external product; both = 0: RLWE key switch, preceded by X -> X^t2n when t2n != 1
//   (scheme/fhew/src/rlwe.rs:188-191 `Rlwe::automorphism`)
template <class A, class W>
__global__ __launch_bounds__(W::THREADS, (fhew_min_waves<A, W>())) void gadget_product_kernel(
    u64 *__restrict__ ct_a, u64 *__restrict__ ct_b, unsigned batch, FhewKey key, unsigned index, unsigned both, unsigned t2n,
    RingConsts K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const unsigned ct = blockIdx.x * W::TEAMS + team;
    if (ct >= batch) return;  // team-uniform exit (a multi-wave team is a whole block: barriers stay matched)
    u64 *lds = reinterpret_cast<u64 *>(smem_raw) + team * W::PN;
    const typename A::K k = A::make(*K.desc, W::LOG_N, 0, 0);
    u64 ca[W::E], cb[W::E];
    wave_load<W>(ca, ct_a + size_t(ct) * W::N, lane);
    wave_load<W>(cb, ct_b + size_t(ct) * W::N, lane);
    if (t2n != 1) {
        wave_automorphism<W>(ca, t2n, lane, lds, K.B.q);
        wave_automorphism<W>(cb, t2n, lane, lds, K.B.q);
    }
    wave_gadget_product<A, W>(ca, cb, key.rows + size_t(index) * key.rows_per_ct * 2 * W::N, key.P, both != 0, lane, lds, K, k);
    wave_store<W>(ca, ct_a + size_t(ct) * W::N, lane);
    wave_store<W>(cb, ct_b + size_t(ct) * W::N, lane);
}

// evaluation-domain rows [rows][N] (natural evaluation order as the forward kernel leaves them) -> key_perm layout;
// pm_b != 0: values stored in the pseudo-Mersenne policy's packed operand form {w mod 2^(b-31), w >> (b-31)} (arith.hpp), so
// the multiply-accumulate reads both words of its fixed operand straight from the load
template <class W>
FHE_HEADER_KERNEL void key_permute_kernel(const u64 *__restrict__ in_a, const u64 *__restrict__ in_b, u64 *__restrict__ out, size_t rows, int pm_b) {
    constexpr int N = 1 << W::LOG_N;
    const size_t total = rows * N;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t row = idx >> W::LOG_N;
        const int e = int(idx & (N - 1));
        const int p = key_perm<W>(e);
        u64 va = in_a[idx], vb = in_b[idx];
        if (pm_b) {
            const u64 lo_mask = (u64(1) << (pm_b - 31)) - 1;
            va = ((va >> (pm_b - 31)) << 32) | (va & lo_mask);
            vb = ((vb >> (pm_b - 31)) << 32) | (vb & lo_mask);
        }
        out[(row * 2 + 0) * N + p] = va;
        out[(row * 2 + 1) * N + p] = vb;
    }
}

// ---- LMKCDEY blind rotation (scheme/fhew/src/bootstrapping.rs:158-231) --------------------------
// op list entry: bit 31 = 1 -> automorphism with ak[idx], else external product with brk[idx]
constexpr unsigned BR_OP_AK = 0x80000000u;

// One wave per ciphertext restates i_minus_i_plus + the walk of blind_rotate_core (bootstrapping.rs:176-231) as an op list.
// dlog[x] for x in [0, 2N): (l << 1) | sign (sign 1 = "minus" map), 0xffffffff if x is not +-5^l (even x).
//
// The reference walks l = N/2-1 .. 1 for i_minus, then for i_plus: at level l it emits the external products of bucket l,
// counts a step, and emits an automorphism by g^v when bucket l-1 is non-empty, v reached w, or l = 1.  Buckets are sparse
// (n_lwe items in N levels), so the walk is generated from the SORTED items instead of level by level: items are ranked by
// (minus before plus, level descending, index ascending) with a counting rank in LDS; between two occupied levels a run of
// S steps emits floor((S-1)/w) automorphisms by g^w and one by g^(S - w floor((S-1)/w)).  Same list, ~n_lwe + N/w steps of
// one lane instead of a serial chain of ~5 N global-memory accesses per ciphertext (0.94 ms per launch before, 4 % of cfg3).
FHE_HEADER_KERNEL void blind_rotate_schedule_kernel(const u64 *__restrict__ lwe_a, unsigned n_lwe, unsigned batch, unsigned n, unsigned w,
                                                    const unsigned *__restrict__ dlog, unsigned *__restrict__ ops, unsigned *__restrict__ nops,
                                                    unsigned max_ops, int *__restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned *keys = reinterpret_cast<unsigned *>(smem_raw);  // [n_lwe] sort key of item i, 0xffffffff: not rotated by
    unsigned *slev = keys + n_lwe;                             // [n_lwe] sorted: (sign << 16) | level
    unsigned *sidx = slev + n_lwe;                             // [n_lwe] sorted: item index
    __shared__ unsigned m_valid;
    const unsigned ct = blockIdx.x, lane = threadIdx.x, half = n / 2;
    if (ct >= batch) return;
    const u64 *a = lwe_a + size_t(ct) * n_lwe;
    if (lane == 0) m_valid = 0;
    __syncthreads();
    for (unsigned i = lane; i < n_lwe; i += blockDim.x) {
        const u64 ai = a[i];
        unsigned key = 0xffffffffu;
        if (ai >= 2 * n) *err = 1;
        else if (ai != 0) {  // a_i = 0: bootstrapping.rs:220
            const unsigned e = dlog[ai];
            if (e == 0xffffffffu) *err = 2;  // even a_i: `unreachable!()` in the reference
            else {
                const unsigned sgn = e & 1, l = e >> 1;
                key = (((1 - sgn) * half + (half - 1 - l)) * n_lwe) + i;  // minus first, level descending, index ascending
                atomicAdd(&m_valid, 1u);
            }
        }
        keys[i] = key;
    }
    __syncthreads();
    for (unsigned i = lane; i < n_lwe; i += blockDim.x) {
        const unsigned key = keys[i];
        if (key == 0xffffffffu) continue;
        unsigned rank = 0;
        for (unsigned j = 0; j < n_lwe; ++j) rank += keys[j] < key;
        const unsigned g = key / n_lwe;  // (1 - sign) * half + (half - 1 - level)
        const unsigned sgn = g < half ? 1u : 0u, l = half - 1 - (g - (1 - sgn) * half);
        slev[rank] = (sgn << 16) | l;
        sidx[rank] = i;
    }
    __syncthreads();
    if (lane != 0) return;
    unsigned *out = ops + size_t(ct) * max_ops;
    const unsigned m = m_valid;
    unsigned k = 0, pos = 0;
    auto emit = [&](unsigned op) { if (k < max_ops) out[k++] = op; };
    for (unsigned sgn = 2; sgn-- > 0;) {  // i_minus (sign 1), then i_plus (sign 0)
        unsigned cur = half - 1;
        for (;;) {
            while (pos < m && slev[pos] == ((sgn << 16) | cur)) emit(sidx[pos++]);  // bucket `cur` (bootstrapping.rs:182-184 / 191-193)
            if (cur == 0) break;
            const bool more = pos < m && (slev[pos] >> 16) == sgn;
            const unsigned nl = more ? (slev[pos] & 0xffffu) : 0u;                 // next occupied level below cur (0: none that matters)
            const unsigned target = nl ? nl + 1 : 1u;                              // the level whose step emits the automorphism
            const unsigned steps = cur - target + 1, full = (steps - 1) / w;
            for (unsigned f = 0; f < full; ++f) emit(BR_OP_AK | w);
            emit(BR_OP_AK | (steps - full * w));
            cur = target - 1;
        }
        if (sgn == 1) emit(BR_OP_AK | 0);  // bootstrapping.rs:194
    }
    nops[ct] = k;
}

struct BlindRotateParams {
    FhewKey brk;          // n_lwe RGSW ciphertexts (2d rows each)
    FhewKey ak;           // w + 1 automorphism keys (d_ks rows each)
    const unsigned *ak_t; // [w + 1] exponents mod 2N
    const unsigned *ops;  // [batch][max_ops]
    const unsigned *nops; // [batch]
    unsigned max_ops;
    const u64 *lwe_b;     // [batch] mod 2N
    const u64 *f;         // LUT polynomial(s)
    size_t f_stride;      // 0: one f for the whole batch, N: one per ciphertext
};

template <class A, class W>
__global__ __launch_bounds__(W::THREADS, (fhew_min_waves<A, W>())) void blind_rotate_kernel(BlindRotateParams BR, u64 *__restrict__ out_a,
                                                                                  u64 *__restrict__ out_b, unsigned batch, RingConsts K) {
    constexpr int E = W::E, N = W::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const unsigned ct = blockIdx.x * W::TEAMS + team;
    if (ct >= batch) return;
    u64 *lds = reinterpret_cast<u64 *>(smem_raw) + team * W::PN;
    const typename A::K k = A::make(*K.desc, W::LOG_N, 0, 0);
    u64 ca[E], cb[E];
    // acc = (0, f.automorphism(-g) * X^(b*g))   (bootstrapping.rs:165-167); both steps are signed index maps
    {
        const u64 *f = BR.f + size_t(ct) * BR.f_stride;
        const unsigned b = unsigned(BR.lwe_b[ct] & (2 * N - 1));
        const unsigned kmono = (b * 5u) & (2 * N - 1);  // (b * g) mod 2N; X^k with k taken mod 2N
        const unsigned tneg = (2 * N - 5u) & (2 * N - 1);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const unsigned i = coef_index<W>(lane, e);
            const u64 v = f[i];
            unsigned pos = (i * tneg) & (2 * N - 1);   // automorphism(-5): X^i -> X^(i t)
            pos = (pos + kmono) & (2 * N - 1);         // * X^k
            lds[lds_phys(pos & (N - 1))] = pos < N ? v : (v ? K.B.q - v : 0);
        }
        exchange_sync<W::WAVE>();
#pragma unroll
        for (int e = 0; e < E; ++e) { cb[e] = lds[lds_phys(coef_index<W>(lane, e))]; ca[e] = 0; }
        exchange_sync<W::WAVE>();
    }
    const unsigned *ops = BR.ops + size_t(ct) * BR.max_ops;
    const unsigned nops = BR.nops[ct];
    for (unsigned o = 0; o < nops; ++o) {
        const unsigned op = __builtin_amdgcn_readfirstlane(ops[o]);
        const bool is_ak = (op & BR_OP_AK) != 0;
        const unsigned idx = op & 0x7fffffffu;
        if (is_ak) {
            const unsigned t = BR.ak_t[idx];
            wave_automorphism<W>(ca, t, lane, lds, K.B.q);
            wave_automorphism<W>(cb, t, lane, lds, K.B.q);
        }
        const FhewKey &key = is_ak ? BR.ak : BR.brk;
        wave_gadget_product<A, W>(ca, cb, key.rows + size_t(idx) * key.rows_per_ct * 2 * N, key.P, !is_ak, lane, lds, K, k);
    }
    wave_store<W>(ca, out_a + size_t(ct) * N, lane);
    wave_store<W>(cb, out_b + size_t(ct) * N, lane);
}

}  // namespace fhe
