// RNS kernels: fast base conversion with the reference's f64 correction
// (util/src/ring/rns.rs:331-345), rescale_k (rns.rs:103-132) and the limb-wise products of
// Ckks::key_switch (scheme/ckks/src/ckks.rs:284-293).  One thread per coefficient; limb-major polynomials.
#pragma once
#include "dev_arith.hpp"
#include "arith.hpp"
#include "pm_dot.hpp"

namespace fhe {

constexpr int RNS_MAX_LIMBS = 32;

// The conversion tables are written once at context creation and only ever read by kernels: reading them through the CONSTANT
// address space lets hipcc use scalar loads (one per wave, batched into s_load_dwordx8/x16 over the unrolled limb loops) where
// a plain global pointer forced a 64-lane vector load of one address per constant -- 248 dependent vector loads per thread
// were what bounded the rescale kernel (400 us per 64 cfg4 ciphertexts).
template <class T>
__device__ __forceinline__ T ldc(const T *p, int i) {
    return ((const __attribute__((address_space(4))) T *)p)[i];
}

// conversion from base A (la moduli) to base B (lb moduli); all tables in HBM.  The route for ANY primes below 2^62
// (lazy Shoup products); bases of pseudo-Mersenne primes of one bit length take the unreduced route further down.
struct BaseConv {
    int la, lb;
    const u64 *a_mod;       // [la]
    const u64 *ahat_inv;    // [la]  (A / a_i)^-1 mod a_i          (rns.rs:290-293 q_hats_inv_qs)
    const u64 *ahat_inv_s;  // [la]  Shoup companions
    const double *frac;     // [la]  1.0 / a_i as f64               (rns.rs:294 q_fracs)
    const u64 *b_mod;       // [lb]
    const u64 *c;           // [lb][la]  (A / a_i) mod b_j          (rns.rs:305-313 q_hats_ps)
    const u64 *c_s;         // [lb][la]  Shoup companions
    const u64 *ua;          // [lb][la + 1]  (u * A) mod b_j         (rns.rs:315-320 uq_ps)
};

__device__ __forceinline__ uint4 ldc4(const uint4 *p, int i) {
    const __attribute__((address_space(4))) uint4 *q = (const __attribute__((address_space(4))) uint4 *)p + i;
    return uint4{q->x, q->y, q->z, q->w};
}

// The per-coefficient limb vectors live in REGISTERS: every loop over source limbs is unrolled to the compile-time bound MAXA
// (1 / 4 / 8 / 16 / 32, the smallest that holds the base) and predicated on the run-time count -- a run-time trip count would
// put v[] / vs[] in scratch memory (measured: 57 us -> see DESIGN.md 4.5 for the cfg4 rescale).
//
// vs_i = v_i * ahat_inv_i mod a_i; u = round(sum_i frac_i * vs_i) with the reference's sequential f64 sum
template <int MAXA, bool FULL>
__device__ __forceinline__ int base_conv_prepare(const BaseConv &C, const u64 (&v)[MAXA], u64 (&vs)[MAXA]) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < MAXA; ++i) {
        if (FULL || i < C.la) {
            const u64 a = ldc(C.a_mod, i);
            vs[i] = csub(mul_shoup_lazy(v[i], ldc(C.ahat_inv, i), ldc(C.ahat_inv_s, i), a), a);
            acc = __dadd_rn(acc, __dmul_rn(ldc(C.frac, i), (double)vs[i]));  // no FMA contraction: matches `.sum::<f64>()`
        } else {
            vs[i] = 0;
        }
    }
    return (int)round(acc);  // f64::round: half away from zero
}

// sum_i c_ji * vs_i - ua_j[u]  (mod b_j), canonical
template <int MAXA, bool FULL>
__device__ __forceinline__ u64 base_conv_out(const BaseConv &C, int j, const u64 (&vs)[MAXA], int u) {
    const int la = FULL ? MAXA : C.la;  // FULL: the source base has exactly MAXA limbs -- no predicate, constant strides
    const u64 b = ldc(C.b_mod, j), b2 = 2 * b;
    u64 dot = 0;
#pragma unroll
    for (int i = 0; i < MAXA; ++i)
        if (FULL || i < C.la) dot = csub(dot + mul_shoup_lazy(vs[i], ldc(C.c, j * la + i), ldc(C.c_s, j * la + i), b), b2);
    dot = csub(dot, b);
    const u64 sub = C.ua[j * (la + 1) + u];  // (u differs per lane: a vector load)
    return dot >= sub ? dot - sub : dot + b - sub;
}

// util/src/ring/rns.rs:83-91: in [batch][la][n] (batch stride in_bs words) -> out [batch][lb][n] (stride out_bs)
template <int MAXA, bool FULL>
__global__ void rns_extend_kernel(const u64 *__restrict__ in, size_t in_bs, u64 *__restrict__ out, size_t out_bs, size_t n, size_t batch,
                                  BaseConv C, u64 *__restrict__ copy, size_t copy_bs) {
    const size_t total = n * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n, i = idx - p * n;
        u64 v[MAXA], vs[MAXA];
#pragma unroll
        for (int l = 0; l < MAXA; ++l) v[l] = (FULL || l < C.la) ? in[p * in_bs + size_t(l) * n + i] : 0;
        if (copy) {  // the key switch wants the source limbs next to the new ones ([batch][la + lb][n]): they are in registers already
#pragma unroll
            for (int l = 0; l < MAXA; ++l)
                if (FULL || l < C.la) copy[p * copy_bs + size_t(l) * n + i] = v[l];
        }
        const int u = base_conv_prepare<MAXA, FULL>(C, v, vs);
        // output limbs in independent chains of up to MAXA at a time (the bound that serves the source base serves the target
        // base of the BASELINE shapes too): the unrolled bodies give the scheduler eight dot products to interleave
        for (int j0 = 0; j0 < C.lb; j0 += MAXA) {
#pragma unroll
            for (int jj = 0; jj < MAXA; ++jj)
                if (j0 + jj < C.lb) out[p * out_bs + size_t(j0 + jj) * n + i] = base_conv_out<MAXA, FULL>(C, j0 + jj, vs, u);
        }
    }
}

struct RescaleConsts {
    int L, K;
    const u64 *q_mod, *p_mod;          // [L], [K]
    const u64 *half_q, *half_p;        // floor(P/2) mod q_i, mod p_j        (rns.rs:120-125 `round`)
    const u64 *pinv, *pinv_s;          // P^-1 mod q_i + Shoup               (rns.rs:127-132 `div`)
    const u64 *red_mu;                 // [L] floor(2^64 / q_i): 64-bit Barrett for `vp % q_i` in the K == 1 path
    BaseConv p2q;                      // switch_bases P -> Q                (rns.rs:93-97)
};

// One q-limb of `rescale_k` (rns.rs:103-118): x = the limb's value, vp / vs / u = the rounded p-limbs and their conversion state
template <int MAXA, bool FULL>
__device__ __forceinline__ u64 rescale_limb(const RescaleConsts &R, int l, u64 q, u64 x, const u64 (&vp)[MAXA], const u64 (&vs)[MAXA], int u) {
    const u64 vq = csub(x + ldc(R.half_q, l), q);
    u64 sw;
    if (R.K == 1) {  // rns.rs:108-111: `*vq -= vp.to_u64()` -> vp % q_i
        const u64 y = vp[0];
        sw = y - __umul64hi(y, ldc(R.red_mu, l)) * q;
        sw = csub(csub(sw, q), q);
    } else {
        sw = base_conv_out<MAXA, FULL>(R.p2q, l, vs, u);
    }
    const u64 diff = vq >= sw ? vq - sw : vq + q - sw;
    return csub(mul_shoup_lazy(diff, ldc(R.pinv, l), ldc(R.pinv_s, l), q), q);
}

// util/src/ring/rns.rs:103-118 `rescale_k(K)`: in [batch][L+K][n] -> out [batch][L][n] (+ addend [batch][L][n] if non-null)
// `out` may alias `addend` (each thread reads its addend element before it writes the same slot)
template <int MAXA, bool FULL>
__global__ void rns_rescale_kernel(const u64 *__restrict__ in, size_t in_bs, u64 *out, size_t out_bs,
                                   const u64 *addend, size_t add_bs, size_t n, size_t batch, RescaleConsts R) {
    const size_t total = n * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n, i = idx - p * n;
        u64 vp[MAXA], vs[MAXA];
#pragma unroll
        for (int j = 0; j < MAXA; ++j) vp[j] = (FULL || j < R.K) ? csub(in[p * in_bs + size_t(R.L + j) * n + i] + ldc(R.half_p, j), ldc(R.p_mod, j)) : 0;
        int u = 0;
        if (R.K > 1) u = base_conv_prepare<MAXA, FULL>(R.p2q, vp, vs);
        for (int l0 = 0; l0 < R.L; l0 += MAXA) {
#pragma unroll
          for (int ll = 0; ll < MAXA; ++ll) {
            const int l = l0 + ll;
            if (l >= R.L) continue;
            const u64 q = ldc(R.q_mod, l);
            u64 r = rescale_limb<MAXA, FULL>(R, l, q, in[p * in_bs + size_t(l) * n + i], vp, vs, u);
            if (addend) r = csub(r + addend[p * add_bs + size_t(l) * n + i], q);
            out[p * out_bs + size_t(l) * n + i] = r;
          }
        }
    }
}

// ==== bases of pseudo-Mersenne primes of ONE bit length B (34 <= B <= 60; every prime `two_adic_primes` yields for the CKKS
// parameter sets of the reference: scheme/ckks/src/ckks.rs:20-35 at 60 bits, util/src/ring/rns.rs:373-386 at 55) ===================
//
// Both conversions are, per coefficient, a dense la x lb contraction with constant coefficients (pm_dot.hpp).  On this route the
// contraction is accumulated UNREDUCED -- three multiply-adds per term against scalar-register constants, one reduction per output --
// and everything else that is linear in the same inputs is folded into the constants on the host:
//   * extend:  out_j = sum_i (A/a_i mod b_j) vs_i + u (b_j - A mod b_j)            -- the reference's table lookup (u A) mod b_j
//     (rns.rs:315-320, 343) is u times one constant, so no per-lane table load and no conditional subtraction;
//   * rescale: out_l = ((x + half_l) - (sum_i c_li vs_i - u P)) P^-1 = sum_i (-c_li P^-1) vs_i + u + x P^-1 + half_l P^-1  (mod q_l):
//     the limb's own value x enters as one more term (four multiply-adds), `round` and `div` (rns.rs:120-132) cost nothing;
//   * the source side vs_i = (v_i + half_i) inv_i = v_i inv_i + (half_i inv_i) is one two-operand product + a constant;
//   * at N = 2^15 (the edge kernels) the transforms' outermost layer is folded in as well: the twiddle of the forward layer into a
//     second set of row constants, n^-1 and n^-1 twi[1] of the inverse layer into the multipliers of x and of the p-limbs.
// All of it is exact arithmetic mod b_j on canonical outputs: bit-identical to the term-by-term form (tests/test_rns_gpu.py checks
// every route against the oracle).

// Tables.  Everything a source limb or an output row needs sits in ONE contiguous record, fetched by scalar loads off ONE base
// pointer (s_load_dwordx16): with a pointer per table the kernels held ~16 table addresses in scalar registers next to 24-56
// constants per row and spilled 200 scalar registers into vector lanes.
// source side, vs_i = (v_i inv_i + hk_i) mod a_i, canonical: record i = 16 dwords
//   {a_lo, a_hi, c = 2^B - a_i, 0, inv.{a0, a1, b0, b1} (two-operand form, arith.hpp), hk_lo, hk_hi, frac_lo, frac_hi (1.0 / a_i, rns.rs:294),
//    w.{a0, a1, b0, b1} (tw[1] of a_i: the extend kernel's own butterflies at N = 2^15)}
struct PmSrc {
    int la;
    const unsigned *rec;  // [la][16]
};
constexpr int PM_SRC_DW = 16;
// target side: one row of constants per output limb, every constant cut at 30 bits (pm_dot.hpp).  Row j = 3 stride + 16 dwords:
//   k0[stride], k1[stride], kk[stride]   M[j][i] = k0 + k1 2^30, kk = k0 + k1; stride = la rounded up to 8, zero padded
//   {u0, u1}                             the multiplier of u
//   {x0, x1, x0', x1'}                   multipliers of the limb's own value: plain (or of the pair's sum), of the pair's difference
//   {c, c60}                             2^B - q, (2^B - q) 2^(60-B)
//   {q_lo, q_hi, kc_lo, kc_hi}           the modulus, the constant term;   4 dwords of padding
struct PmRows {
    int stride;
    const unsigned *tab;  // [rows][3 stride + 16]
};
constexpr int PM_ROW_TAIL = 16;
struct PmRowK {  // a row's tail as the kernels use it
    unsigned u0, u1, x0, x1, xd0, xd1, c, c60;
    u64 q, kc;
};
__device__ __forceinline__ PmRowK pm_row_tail(const PmRows &R, int j) {
    const unsigned *t = R.tab + size_t(j) * (3 * R.stride + PM_ROW_TAIL) + 3 * R.stride;
    PmRowK k;
    k.u0 = ldc(t, 0); k.u1 = ldc(t, 1); k.x0 = ldc(t, 2); k.x1 = ldc(t, 3); k.xd0 = ldc(t, 4); k.xd1 = ldc(t, 5); k.c = ldc(t, 6); k.c60 = ldc(t, 7);
    k.q = (u64)ldc(t, 8) | ((u64)ldc(t, 9) << 32);
    k.kc = (u64)ldc(t, 10) | ((u64)ldc(t, 11) << 32);
    return k;
}

// one source limb: vs (canonical) cut at 30 bits; acc += frac_i * vs in the reference's order (rns.rs:337-341)
template <bool HALF>
__device__ __forceinline__ pd::Y3 pm_src_one(const unsigned *rec, u64 v, double &acc, const pd::Uni &U) {
    const u64 a = (u64)ldc(rec, 0) | ((u64)ldc(rec, 1) << 32);
    const unsigned c = ldc(rec, 2);
    u64 r = pd::ds_mul_raw(v, ldc(rec, 4), ldc(rec, 5), ldc(rec, 6), ldc(rec, 7), 2 * c, U);  // < 2^(B+3)
    if constexpr (HALF) r += (u64)ldc(rec, 8) | ((u64)ldc(rec, 9) << 32);                      // < 2^63 + 2^60
    const u64 vs = csub(pd::fold(r, c, U), a);                                                 // fold: < 2^B + 9c < 2 a
    const double frac = __longlong_as_double((long long)((u64)ldc(rec, 10) | ((u64)ldc(rec, 11) << 32)));
    acc = __dadd_rn(acc, __dmul_rn(frac, (double)vs));  // no FMA contraction: matches `.sum::<f64>()`
    return pd::split30(vs);
}

// row j: sum_i M[j][i] vs_i + u U_j (+ X x + kc), lazily reduced (< 2^B + 2^31: one conditional subtraction from canonical).
// xm0, xm1: the multiplier of x cut at 30 bits (the caller picks the pair's sum or difference set)
template <int MAXA, bool XTERM>
__device__ __forceinline__ u64 pm_row(const PmRows &R, int j, const PmRowK &k, const unsigned (&y0)[MAXA], const unsigned (&y1)[MAXA],
                                      const unsigned (&yk)[MAXA], unsigned u, const pd::Y3 &x, unsigned xm0, unsigned xm1, const pd::Uni &U) {
    constexpr int G = (MAXA + 7) / 8;
    const unsigned *k0 = R.tab + size_t(j) * (3 * R.stride + PM_ROW_TAIL), *k1 = k0 + R.stride, *kk = k1 + R.stride;
    u64 total = 0;
    static_for<0, G>([&](auto gc) {
        constexpr int g = decltype(gc)::value, I0 = g * 8, I1 = (I0 + 8 < MAXA) ? I0 + 8 : MAXA;
        u64 s00 = 0, s11 = 0, sk = 0;
#pragma unroll
        for (int i = I0; i < I1; ++i) {  // (limbs past the base have zero constants and zero residues)
            s00 += (u64)ldc(k0, i) * y0[i];
            s11 += (u64)ldc(k1, i) * y1[i];
            sk += (u64)ldc(kk, i) * yk[i];
        }
        const u64 s01a = sk - s00 - s11;  // Karatsuba, mod 2^64: the true cross sum is below 2^64
        u64 s01b = 0;
        if constexpr (g == 0) {
            s00 += (u64)k.u0 * u;
            s01b = (u64)k.u1 * u;
            if constexpr (XTERM) {
                s00 += (u64)xm0 * x.y0 + k.kc;
                s11 += (u64)xm1 * x.y1;
                s01b += (u64)xm0 * x.y1 + (u64)xm1 * x.y0;
            }
        }
        total += pd::reduce_lazy(s00, s01a, s01b, s11, k.c, k.c60, U);
    });
    if constexpr (G > 1) total = pd::fold(total, k.c, U);
    return total;
}

// util/src/ring/rns.rs:83-91: in [batch][la][n] (batch stride in_bs words) -> out [batch][rows][n] (stride out_bs)
template <int MAXA, bool FULL>
__global__ void rns_extend_pm_kernel(const u64 *__restrict__ in, size_t in_bs, u64 *__restrict__ out, size_t out_bs, size_t n, size_t batch,
                                     PmSrc S, PmRows R, int rows, pd::Uni U, u64 *__restrict__ copy, size_t copy_bs, int c_lo, int c_hi) {
    const size_t total = n * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n, i = idx - p * n;
        unsigned y0[MAXA], y1[MAXA], yk[MAXA];
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l < MAXA; ++l) {
            if (FULL || l < S.la) {
                const u64 v = in[p * in_bs + size_t(l) * n + i];
                // the key switch wants the source limbs next to the new ones (a limb-sharded one: the limbs [c_lo, c_hi) it owns)
                if (copy && l >= c_lo && l < c_hi) copy[p * copy_bs + size_t(l - c_lo) * n + i] = v;
                const pd::Y3 y = pm_src_one<false>(S.rec + l * PM_SRC_DW, v, acc, U);
                y0[l] = y.y0; y1[l] = y.y1; yk[l] = y.yk;
                FHE_SCHED_FENCE();  // one limb's record (16 scalar registers) at a time
            } else {
                y0[l] = y1[l] = yk[l] = 0;
            }
        }
        const unsigned u = (unsigned)(int)round(acc);  // f64::round: half away from zero
        u64 *dst = out + p * out_bs + i;
#pragma unroll 1
        for (int j = 0; j < rows; ++j) {  // rolled: one row of constants (scalar registers) live at a time
            const PmRowK k = pm_row_tail(R, j);
            dst[size_t(j) * n] = csub(pm_row<MAXA, false>(R, j, k, y0, y1, yk, u, pd::Y3{0, 0, 0}, 0, 0, U), k.q);
        }
    }
}

// where the limbs of a rescale's input live.  The whole key switch keeps them in one [batch][L+K][n] block; a limb-sharded one reads
// its own q-limbs from one buffer and the K p-limbs from the all-gathered contributions of the other devices: p-limb j of
// ciphertext c sits at in_p + off[j] + c * p_bs (off[j] set by the host: contribution j / np, local limb j % np)
struct RescaleIn {
    const u64 *in_q;  // [batch][L][n] (stride q_bs): the q-limbs this launch rescales
    size_t q_bs;
    const u64 *in_p;
    size_t p_bs;
    size_t off[RNS_MAX_LIMBS];
};

// util/src/ring/rns.rs:103-118 `rescale_k(K)`: L q-limbs + K p-limbs (RescaleIn) -> out [batch][L][n] (+ addend [batch][L][n] if non-null)
// `out` may alias `addend` (each thread reads its addend element before it writes the same slot).  S: the p-limbs' side with
// hk = half_j inv_j; R: one row per q-limb.  K == 1 is the reference's shortcut (rns.rs:108-111): no correction term u.
template <int MAXA, bool FULL>
__global__ void rns_rescale_pm_kernel(RescaleIn I, u64 *out, size_t out_bs, const u64 *addend, size_t add_bs,
                                      size_t n, size_t batch, int L, PmSrc S, PmRows R, pd::Uni U) {
    const size_t total = n * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n, i = idx - p * n;
        const u64 *src = I.in_q + p * I.q_bs + i, *psrc = I.in_p + p * I.p_bs + i, *ad = addend ? addend + p * add_bs + i : nullptr;
        u64 *dst = out + p * out_bs + i;
        unsigned y0[MAXA], y1[MAXA], yk[MAXA];
        double acc = 0.0;
        u64 vp[MAXA];
#pragma unroll
        for (int j = 0; j < MAXA; ++j) vp[j] = (FULL || j < S.la) ? psrc[I.off[j]] : 0;  // every load in flight before the first product
        u64 nx = src[0], na = ad ? ad[0] : 0;  // q-limb 0 flies under the p-limbs' work
#pragma unroll
        for (int j = 0; j < MAXA; ++j) {
            if (FULL || j < S.la) {
                const pd::Y3 y = pm_src_one<true>(S.rec + j * PM_SRC_DW, vp[j], acc, U);
                y0[j] = y.y0; y1[j] = y.y1; yk[j] = y.yk;
                FHE_SCHED_FENCE();
            } else {
                y0[j] = y1[j] = yk[j] = 0;
            }
        }
        const unsigned u = S.la > 1 ? (unsigned)(int)round(acc) : 0u;
#pragma unroll 1
        for (int l = 0; l < L; ++l) {  // rolled; the next limb's memory operands are requested before this limb's products start
            const u64 x = nx, a = na;
            if (l + 1 < L) {
                nx = src[size_t(l + 1) * n];
                if (ad) na = ad[size_t(l + 1) * n];
            }
            const PmRowK k = pm_row_tail(R, l);
            u64 r = csub(pm_row<MAXA, true>(R, l, k, y0, y1, yk, u, pd::split30(x), k.x0, k.x1, U), k.q);
            if (ad) r = csub(r + a, k.q);
            dst[size_t(l) * n] = r;
        }
    }
}

// ---- the key switch at N = 2^15: the transforms' OUTERMOST layer runs in their neighbours -------------------------------------
// A 2^15 ring does not fit the two-workgroups-per-CU kernel of ntt14w.hpp; as ONE workgroup per CU (R0 = 4) its memory phases
// and its butterflies do not overlap across workgroups and it runs at two thirds of the 2^14 kernel's rate.  But layer 0 of
// the forward transform (util/src/ring/fft.rs:42-52 with m = 1: the pairs (i, i + n/2), ONE twiddle tw[1]) and layer 0 of the
// inverse (fft.rs:62-76, twi[1], then n^-1) are elementwise over such pairs -- exactly the shape of the per-coefficient kernels
// that produce the forward's input (`extend_bases`) and consume the inverse's output (`rescale_k`).  With a thread of those
// kernels owning the pair (i, i + n/2) the layer costs them next to nothing (it is folded into their constants, see above), and
// what is left of every transform is two INDEPENDENT 2^14 sub-transforms (ntt14w PFX form, pb = 1).  All arithmetic is exact mod
// q_l, so the coefficient-domain results are bit-identical.  Pseudo-Mersenne bases only.

// extend_bases + layer 0 of the forward transform of the source limbs [c_lo, c_hi) and of `rows` new limbs:
// in [batch][la][n] -> out [batch][c_hi - c_lo + rows][n] (the whole key switch: c_lo = 0, c_hi = la).
// RW = R with every constant multiplied by the target modulus' tw[1]: the butterfly (X, Y) <- (X + w Y, X - w Y) (fft.rs:96-101)
// of an output limb takes w Y straight from the second dot product; the source limbs' own butterflies take tw[1] from their record.
template <int MAXA, bool FULL>
__global__ void rns_extend_edge_pm_kernel(const u64 *__restrict__ in, size_t in_bs, u64 *__restrict__ out, size_t out_bs, size_t n, size_t batch,
                                          PmSrc S, PmRows R, PmRows RW, int rows, pd::Uni U, int c_lo, int c_hi) {
    const size_t h = n >> 1, total = h * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / h, i = idx - p * h;
        const u64 *src = in + p * in_bs + i;
        u64 *dst = out + p * out_bs + i;
        unsigned a0[MAXA], a1[MAXA], ak[MAXA], b0[MAXA], b1[MAXA], bk[MAXA];  // the pair's residues, cut at 30 bits
        double acc0 = 0.0, acc1 = 0.0;
        u64 va[MAXA], vb[MAXA];
#pragma unroll
        for (int l = 0; l < MAXA; ++l) {  // every load in flight before the first product
            va[l] = (FULL || l < S.la) ? src[size_t(l) * n] : 0;
            vb[l] = (FULL || l < S.la) ? src[size_t(l) * n + h] : 0;
        }
#pragma unroll
        for (int l = 0; l < MAXA; ++l) {
            if (FULL || l < S.la) {
                const unsigned *rec = S.rec + l * PM_SRC_DW;
                const u64 v0 = va[l], v1 = vb[l];
                const pd::Y3 ya = pm_src_one<false>(rec, v0, acc0, U), yb = pm_src_one<false>(rec, v1, acc1, U);
                a0[l] = ya.y0; a1[l] = ya.y1; ak[l] = ya.yk;
                b0[l] = yb.y0; b1[l] = yb.y1; bk[l] = yb.yk;
                // the source limb's own butterfly
                const u64 a = (u64)ldc(rec, 0) | ((u64)ldc(rec, 1) << 32);
                const unsigned c = ldc(rec, 2);
                const u64 t = csub(pd::fold(pd::ds_mul_raw(v1, ldc(rec, 12), ldc(rec, 13), ldc(rec, 14), ldc(rec, 15), 2 * c, U), c, U), a);
                if (l >= c_lo && l < c_hi) {  // (a limb-sharded key switch keeps only the source limbs it owns)
                    dst[size_t(l - c_lo) * n] = csub(v0 + t, a);
                    dst[size_t(l - c_lo) * n + h] = v0 >= t ? v0 - t : v0 + a - t;
                }
                FHE_SCHED_FENCE();
            } else {
                a0[l] = a1[l] = ak[l] = b0[l] = b1[l] = bk[l] = 0;
            }
        }
        const unsigned u0 = (unsigned)(int)round(acc0), u1 = (unsigned)(int)round(acc1);
        u64 *dp = dst + size_t(c_hi - c_lo) * n;
#pragma unroll 1
        for (int j = 0; j < rows; ++j) {
            const PmRowK k = pm_row_tail(R, j), kw = pm_row_tail(RW, j);
            const u64 b = k.q;
            const u64 x = csub(pm_row<MAXA, false>(R, j, k, a0, a1, ak, u0, pd::Y3{0, 0, 0}, 0, 0, U), b);
            FHE_SCHED_FENCE();  // one row of constants (36 scalar registers) at a time
            const u64 t = csub(pm_row<MAXA, false>(RW, j, kw, b0, b1, bk, u1, pd::Y3{0, 0, 0}, 0, 0, U), b);
            dp[size_t(j) * n] = csub(x + t, b);
            dp[size_t(j) * n + h] = x >= t ? x - t : x + b - t;
        }
    }
}

// layer 0 of the inverse transform (+ n^-1) of all L + K limbs + rescale_k: L q-limbs + K p-limbs (RescaleIn) -> out [batch][L][n] (+ addend).
// A thread owns the pair (i, i + n/2): with (z0, z1) the pair's values of a limb, the layer gives ((z0 + z1) n^-1, (z0 - z1) twi[1] n^-1)
// (fft.rs:108-113 `dif`, then fft.rs:73-76).  S0 / S1: the p-limbs' side for the pair's sum / difference (multipliers n^-1 inv_j and
// n^-1 twi[1] inv_j, hk = half_j inv_j); R.xc[l] = {n^-1 P^-1, n^-1 twi[1] P^-1} of q_l, each cut at 30 bits.
template <int MAXA, bool FULL>
__global__ void rns_rescale_edge_pm_kernel(RescaleIn I, u64 *out, size_t out_bs, const u64 *addend, size_t add_bs,
                                           size_t n, size_t batch, int L, PmSrc S0, PmSrc S1, PmRows R, pd::Uni U) {
    const size_t h = n >> 1, total = h * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / h, i = idx - p * h;
        const u64 *src = I.in_q + p * I.q_bs + i, *psrc = I.in_p + p * I.p_bs + i;
        const u64 *ad = addend ? addend + p * add_bs + i : nullptr;
        u64 *dst = out + p * out_bs + i;
        unsigned a0[MAXA], a1[MAXA], ak[MAXA], b0[MAXA], b1[MAXA], bk[MAXA];
        double acc0 = 0.0, acc1 = 0.0;
        u64 va[MAXA], vb[MAXA];
#pragma unroll
        for (int j = 0; j < MAXA; ++j) {  // every load in flight before the first product
            va[j] = (FULL || j < S0.la) ? psrc[I.off[j]] : 0;
            vb[j] = (FULL || j < S0.la) ? psrc[I.off[j] + h] : 0;
        }
#pragma unroll
        for (int j = 0; j < MAXA; ++j) {
            if (FULL || j < S0.la) {
                const u64 z0 = va[j], z1 = vb[j], pm = (u64)ldc(S0.rec + j * PM_SRC_DW, 0) | ((u64)ldc(S0.rec + j * PM_SRC_DW, 1) << 32);
                const pd::Y3 ya = pm_src_one<true>(S0.rec + j * PM_SRC_DW, csub(z0 + z1, pm), acc0, U);
                const pd::Y3 yb = pm_src_one<true>(S1.rec + j * PM_SRC_DW, z0 >= z1 ? z0 - z1 : z0 + pm - z1, acc1, U);
                a0[j] = ya.y0; a1[j] = ya.y1; ak[j] = ya.yk;
                b0[j] = yb.y0; b1[j] = yb.y1; bk[j] = yb.yk;
                FHE_SCHED_FENCE();
            } else {
                a0[j] = a1[j] = ak[j] = b0[j] = b1[j] = bk[j] = 0;
            }
        }
        const unsigned u0 = S0.la > 1 ? (unsigned)(int)round(acc0) : 0u, u1 = S0.la > 1 ? (unsigned)(int)round(acc1) : 0u;
        u64 nx0 = src[0], nx1 = src[h], na0 = ad ? ad[0] : 0, na1 = ad ? ad[h] : 0;  // (not before the p-limbs: the registers are not there)
#pragma unroll 1
        for (int l = 0; l < L; ++l) {
            const u64 z0 = nx0, z1 = nx1, ad0 = na0, ad1 = na1;
            if (l + 1 < L) {
                nx0 = src[size_t(l + 1) * n]; nx1 = src[size_t(l + 1) * n + h];
                if (ad) { na0 = ad[size_t(l + 1) * n]; na1 = ad[size_t(l + 1) * n + h]; }
            }
            const PmRowK k = pm_row_tail(R, l);
            const u64 q = k.q;
            const pd::Y3 xs = pd::split30(csub(z0 + z1, q)), xd = pd::split30(z0 >= z1 ? z0 - z1 : z0 + q - z1);
            u64 r0 = csub(pm_row<MAXA, true>(R, l, k, a0, a1, ak, u0, xs, k.x0, k.x1, U), q);
            u64 r1 = csub(pm_row<MAXA, true>(R, l, k, b0, b1, bk, u1, xd, k.xd0, k.xd1, U), q);
            if (ad) { r0 = csub(r0 + ad0, q); r1 = csub(r1 + ad1, q); }
            dst[size_t(l) * n] = r0;
            dst[size_t(l) * n + h] = r1;
        }
    }
}

// Measured and dropped (round 3, per 64 cfg4 ciphertexts, pair form above: 112 us, 52 % of its wave cycles parked on loads with 4 loads
// in flight per thread): (i) ONE coefficient per lane, the butterfly partner at lane ^ 32 -- 105 registers (the lane halves select
// their constants), still four waves per SIMD: 119 us; (ii) the pair form with the row loop's operands requested two limbs ahead
// and the Karatsuba sums y0 + y1 formed per use to pay for it -- the allocator spills (164 bytes of scratch per lane): 238 us.

// ob = kb (.) e, oa = ka (.) e limb-wise in the evaluation domain; e, ob, oa: [batch][lk][n]; kb, ka: [lk][n].
// blockIdx.y = (ciphertext, limb): the limb's Barrett constants come from scalar loads and no thread divides anything
FHE_HEADER_KERNEL void rns_pointwise2_kernel(const u64 *__restrict__ e, const u64 *__restrict__ kb, const u64 *__restrict__ ka,
                                             u64 *__restrict__ ob, u64 *__restrict__ oa, unsigned n, unsigned lk, size_t polys,
                                             const Barrett *__restrict__ B) {
    for (size_t y = blockIdx.y; y < polys; y += gridDim.y) {
        const unsigned limb = unsigned(y % lk);
        const Barrett b{ldc(&B[limb].q, 0), ldc(&B[limb].mu, 0), ldc(&B[limb].sh1, 0), ldc(&B[limb].sh2, 0)};
        const size_t base = y * n, kbase = size_t(limb) * n;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const u64 x = e[base + i];
            ob[base + i] = mulmod_barrett(x, kb[kbase + i], b);
            oa[base + i] = mulmod_barrett(x, ka[kbase + i], b);
        }
    }
}

// the same for a launch whose limbs are a SUBSET of a context's (limb-sharded key switch at n = 1): moduli from the shard's descriptors
FHE_HEADER_KERNEL void rns_pointwise2_desc_kernel(const u64 *__restrict__ e, const u64 *__restrict__ kb, const u64 *__restrict__ ka,
                                                  u64 *__restrict__ ob, u64 *__restrict__ oa, unsigned n, unsigned lk, size_t polys,
                                                  const ModDesc *__restrict__ D) {
    for (size_t y = blockIdx.y; y < polys; y += gridDim.y) {
        const unsigned limb = unsigned(y % lk);
        const Barrett b{D[limb].q, D[limb].bar_mu, D[limb].bar_sh1, D[limb].bar_sh2};
        const size_t base = y * n, kbase = size_t(limb) * n;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const u64 x = e[base + i];
            ob[base + i] = mulmod_barrett(x, kb[kbase + i], b);
            oa[base + i] = mulmod_barrett(x, ka[kbase + i], b);
        }
    }
}
// in [groups][nl][n] -> out_q [groups][nq][n], out_p [groups][nl - nq][n] (ring sizes whose inverse transform has no split store)
FHE_HEADER_KERNEL void rns_split_limbs_kernel(const u64 *__restrict__ in, u64 *__restrict__ out_q, u64 *__restrict__ out_p, size_t n, size_t groups,
                                              int nl, int nq) {
    const size_t total = groups * size_t(nl) * n;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t poly = idx / n, i = idx - poly * n, g = poly / nl;
        const int l = int(poly - g * nl);
        if (l < nq) out_q[(g * nq + l) * n + i] = in[idx];
        else out_p[(g * (nl - nq) + (l - nq)) * n + i] = in[idx];
    }
}

// util/src/ring/rns.rs:148-158 `RnsRq *= &RnsRq` (evaluation basis): a[b][l][i] <- a[b][l][i] * b[b][l][i] mod m_l, limbs-major
FHE_HEADER_KERNEL void rns_pointwise_kernel(u64 *__restrict__ a, const u64 *__restrict__ b, unsigned n, unsigned limbs, size_t polys,
                                            const Barrett *__restrict__ B) {
    for (size_t y = blockIdx.y; y < polys; y += gridDim.y) {
        const unsigned limb = unsigned(y % limbs);
        const Barrett m{ldc(&B[limb].q, 0), ldc(&B[limb].mu, 0), ldc(&B[limb].sh1, 0), ldc(&B[limb].sh2, 0)};
        const size_t base = y * n;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
            a[base + i] = mulmod_barrett(a[base + i], b[base + i], m);
    }
}

// util/src/ring/rns.rs:254-270 `RnsRq` +=, -=, unary - (either basis): a <- a + b (op 0), a - b (op 1), -a (op 2: b unused), limb by limb
FHE_HEADER_KERNEL void rns_addsub_kernel(u64 *__restrict__ a, const u64 *__restrict__ b, unsigned n, unsigned limbs, size_t polys,
                                         const Barrett *__restrict__ B, int op) {
    for (size_t y = blockIdx.y; y < polys; y += gridDim.y) {
        const u64 q = ldc(&B[unsigned(y % limbs)].q, 0);
        const size_t base = y * n;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const u64 x = a[base + i];
            u64 r;
            if (op == 0) r = csub(x + b[base + i], q);
            else if (op == 1) { const u64 y2 = b[base + i]; r = x >= y2 ? x - y2 : x + q - y2; }
            else r = x ? q - x : 0;
            a[base + i] = r;
        }
    }
}

// scheme/ckks/src/ckks.rs:256-260, the tensor of `Ckks::mul` in the evaluation domain, limb by limb:
// d0 = b0 (.) b1, d1 = b0 (.) a1 + a0 (.) b1, d2 = a0 (.) a1.  e: [4][polys][n] in the order b0, a0, b1, a1 (polys = batch * limbs);
// d: [3][polys][n].  The products of the reference are coefficient-domain `Rq * Rq` (three transforms each): mathematically the
// same polynomials, so the inverse transforms of d0, d1, d2 are bit-identical to them.
FHE_HEADER_KERNEL void rns_tensor_kernel(const u64 *__restrict__ e, u64 *__restrict__ d, unsigned n, unsigned limbs, size_t polys,
                                         const Barrett *__restrict__ B) {
    const size_t plane = polys * n;
    for (size_t y = blockIdx.y; y < polys; y += gridDim.y) {
        const unsigned limb = unsigned(y % limbs);
        const Barrett m{ldc(&B[limb].q, 0), ldc(&B[limb].mu, 0), ldc(&B[limb].sh1, 0), ldc(&B[limb].sh2, 0)};
        const size_t base = y * n;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const u64 b0 = e[base + i], a0 = e[plane + base + i], b1 = e[2 * plane + base + i], a1 = e[3 * plane + base + i];
            d[base + i] = mulmod_barrett(b0, b1, m);
            d[plane + base + i] = csub(mulmod_barrett(b0, a1, m) + mulmod_barrett(a0, b1, m), m.q);
            d[2 * plane + base + i] = mulmod_barrett(a0, a1, m);
        }
    }
}

// scheme/ckks/src/ckks.rs:127-129 `CkksCiphertext::automorphism` -> util/src/avec.rs:34-50 on every limb of an RnsRq:
// in, out [polys][n] (polys = batch * limbs), odd t (a bijection: every output slot is written exactly once)
FHE_HEADER_KERNEL void rns_automorphism_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, unsigned n, unsigned limbs, size_t polys,
                                               unsigned t, const Barrett *__restrict__ B) {
    for (size_t y = blockIdx.y; y < polys; y += gridDim.y) {
        const u64 q = ldc(&B[y % limbs].q, 0);
        const size_t base = y * n;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const unsigned it = unsigned((u64(i) * t) & (2 * n - 1));
            const u64 v = in[base + i];
            if (it < n) out[base + it] = v;
            else out[base + it - n] = v ? q - v : 0;
        }
    }
}

}  // namespace fhe
