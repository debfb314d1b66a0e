// RNS kernels: fast base conversion with the reference's f64 correction
// (util/src/ring/rns.rs:331-345), rescale_k (rns.rs:103-132) and the limb-wise products of
// Ckks::key_switch (scheme/ckks/src/ckks.rs:284-293).  One thread per coefficient; limb-major polynomials.
#pragma once
#include "dev_arith.hpp"
#include "arith.hpp"

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

// conversion from base A (la moduli) to base B (lb moduli); all tables in HBM
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
    // the same constants in the two-operand form of ArithDS<60> (arith.hpp), present when every modulus of both bases is a 60-bit
    // pseudo-Mersenne prime (all of CkksParam's): a product is 6 multiply-adds + a 3-instruction fold instead of the ~14 of a lazy
    // Shoup product -- the conversions are bound by exactly these products (80 per coefficient at cfg4)
    const uint4 *ahat_inv_ds;  // [la]
    const uint4 *c_ds;         // [lb][la]
    const unsigned *a_c, *b_c; // [la], [lb]  2^60 - modulus
    unsigned pw;               // 2^29, handed over as a VALUE the compiler cannot see: as a literal it turns the third product of
                               // ArithDS::mul_raw into a 64-bit shift + a 64-bit add (two instructions for one multiply-add)
};

__device__ __forceinline__ DsK rns_dsk(u64 q, unsigned c, unsigned pw) { return DsK{q, 2 * q, 4 * q, c, 2 * c, pw}; }
__device__ __forceinline__ uint4 ldc4(const uint4 *p, int i) {
    const __attribute__((address_space(4))) uint4 *q = (const __attribute__((address_space(4))) uint4 *)p + i;
    return uint4{q->x, q->y, q->z, q->w};
}

// The per-coefficient limb vectors live in REGISTERS: every loop over source limbs is unrolled to the compile-time bound MAXA
// (4 / 8 / 16 / 32, the smallest that holds the base) and predicated on the run-time count -- a run-time trip count would
// put v[] / vs[] in scratch memory (measured: 57 us -> see DESIGN.md 4.5 for the cfg4 rescale).
//
// vs_i = v_i * ahat_inv_i mod a_i; u = round(sum_i frac_i * vs_i) with the reference's sequential f64 sum
template <int MAXA, bool FULL, bool DS = false>
__device__ __forceinline__ int base_conv_prepare(const BaseConv &C, const u64 (&v)[MAXA], u64 (&vs)[MAXA]) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < MAXA; ++i) {
        if (FULL || i < C.la) {
            const u64 a = ldc(C.a_mod, i);
            if constexpr (DS) vs[i] = csub(ArithDS<60>::mul(v[i], ldc4(C.ahat_inv_ds, i), rns_dsk(a, ldc(C.a_c, i), C.pw)), a);  // < q + 9c -> canonical
            else vs[i] = csub(mul_shoup_lazy(v[i], ldc(C.ahat_inv, i), ldc(C.ahat_inv_s, i), a), a);
            acc = __dadd_rn(acc, __dmul_rn(ldc(C.frac, i), (double)vs[i]));  // no FMA contraction: matches `.sum::<f64>()`
        } else {
            vs[i] = 0;
        }
    }
    return (int)round(acc);  // f64::round: half away from zero
}

// sum_i c_ji * vs_i (mod b_j), canonical
// LAZY (two-operand form only): the last conditional subtraction is left out, the result is < b_j + 9c
template <int MAXA, bool FULL, bool DS = false, bool LAZY = false>
__device__ __forceinline__ u64 base_conv_dot(const BaseConv &C, int j, const u64 (&vs)[MAXA]) {
    const int la = FULL ? MAXA : C.la;  // FULL: the source base has exactly MAXA limbs -- no predicate, constant strides
    const u64 b = ldc(C.b_mod, j), b2 = 2 * b;
    u64 dot = 0;
    if constexpr (DS) {
        const DsK m = rns_dsk(b, ldc(C.b_c, j), C.pw);
#pragma unroll
        for (int i = 0; i < MAXA; ++i) {
            if (FULL || i < C.la) dot += ArithDS<60>::mul(vs[i], ldc4(C.c_ds, j * la + i), m);  // each < q + 9c: eight of them fit 2^63
            if ((i & 7) == 7 && i + 1 < MAXA) dot = ArithDS<60>::fold1(dot, m);
        }
        dot = ArithDS<60>::fold1(dot, m);  // < 2^60 + 8c = b + 9c
        if constexpr (!LAZY) dot = csub(dot, b);
    } else {
#pragma unroll
        for (int i = 0; i < MAXA; ++i)
            if (FULL || i < C.la) dot = csub(dot + mul_shoup_lazy(vs[i], ldc(C.c, j * la + i), ldc(C.c_s, j * la + i), b), b2);
        dot = csub(dot, b);
    }
    return dot;
}
// ... - ua_j[u]  (mod b_j), canonical
template <int MAXA, bool FULL, bool DS = false>
__device__ __forceinline__ u64 base_conv_out(const BaseConv &C, int j, const u64 (&vs)[MAXA], int u) {
    const int la = FULL ? MAXA : C.la;
    const u64 dot = base_conv_dot<MAXA, FULL, DS>(C, j, vs), b = ldc(C.b_mod, j);
    const u64 sub = C.ua[j * (la + 1) + u];  // (u differs per lane: a vector load)
    return dot >= sub ? dot - sub : dot + b - sub;
}

// util/src/ring/rns.rs:83-91: in [batch][la][n] (batch stride in_bs words) -> out [batch][lb][n] (stride out_bs)
template <int MAXA, bool FULL, bool DS = false>
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
        const int u = base_conv_prepare<MAXA, FULL, DS>(C, v, vs);
        if constexpr (DS) {  // rolled over the output limbs, the table entry one limb ahead: see the edge kernels below
            const u64 *ua = C.ua + u;
            const int la = FULL ? MAXA : C.la;
            u64 nsub = ua[0];
#pragma unroll 1
            for (int j = 0; j < C.lb; ++j) {
                const u64 sub = nsub;
                if (j + 1 < C.lb) nsub = ua[(j + 1) * (la + 1)];
                const u64 b = ldc(C.b_mod, j), d = base_conv_dot<MAXA, FULL, true>(C, j, vs);
                out[p * out_bs + size_t(j) * n + i] = d >= sub ? d - sub : d + b - sub;
            }
            continue;
        }
        // output limbs in independent chains of up to MAXA at a time (the bound that serves the source base serves the target
        // base of the BASELINE shapes too): the unrolled bodies give the scheduler eight dot products to interleave
        for (int j0 = 0; j0 < C.lb; j0 += MAXA) {
#pragma unroll
            for (int jj = 0; jj < MAXA; ++jj)
                if (j0 + jj < C.lb) out[p * out_bs + size_t(j0 + jj) * n + i] = base_conv_out<MAXA, FULL, DS>(C, j0 + jj, vs, u);
        }
    }
}

struct RescaleConsts {
    int L, K;
    const u64 *q_mod, *p_mod;          // [L], [K]
    const u64 *half_q, *half_p;        // floor(P/2) mod q_i, mod p_j        (rns.rs:120-125 `round`)
    const u64 *pinv, *pinv_s;          // P^-1 mod q_i + Shoup               (rns.rs:127-132 `div`)
    const u64 *red_mu;                 // [L] floor(2^64 / q_i): 64-bit Barrett for `vp % q_i` in the K == 1 path
    const uint4 *pinv_ds;              // [L] P^-1 mod q_i in the two-operand form (with p2q.*_ds)
    BaseConv p2q;                      // switch_bases P -> Q                (rns.rs:93-97)
};

// One q-limb of `rescale_k` (rns.rs:103-118): x = the limb's value, vp / vs / u = the rounded p-limbs and their conversion state
template <int MAXA, bool FULL, bool DS>
__device__ __forceinline__ u64 rescale_limb(const RescaleConsts &R, int l, u64 q, u64 x, const u64 (&vp)[MAXA], const u64 (&vs)[MAXA], int u) {
    const u64 vq = csub(x + ldc(R.half_q, l), q);
    u64 sw;
    if (R.K == 1) {  // rns.rs:108-111: `*vq -= vp.to_u64()` -> vp % q_i
        const u64 y = vp[0];
        sw = y - __umul64hi(y, ldc(R.red_mu, l)) * q;
        sw = csub(csub(sw, q), q);
    } else {
        sw = base_conv_out<MAXA, FULL, DS>(R.p2q, l, vs, u);
    }
    const u64 diff = vq >= sw ? vq - sw : vq + q - sw;
    if constexpr (DS) return csub(ArithDS<60>::mul(diff, ldc4(R.pinv_ds, l), rns_dsk(q, ldc(R.p2q.b_c, l), R.p2q.pw)), q);
    else return csub(mul_shoup_lazy(diff, ldc(R.pinv, l), ldc(R.pinv_s, l), q), q);
}

// util/src/ring/rns.rs:103-118 `rescale_k(K)`: in [batch][L+K][n] -> out [batch][L][n] (+ addend [batch][L][n] if non-null)
// `out` may alias `addend` (each thread reads its addend element before it writes the same slot)
template <int MAXA, bool FULL, bool DS = false>
__global__ void rns_rescale_kernel(const u64 *__restrict__ in, size_t in_bs, u64 *out, size_t out_bs,
                                   const u64 *addend, size_t add_bs, size_t n, size_t batch, RescaleConsts R) {
    const size_t total = n * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n, i = idx - p * n;
        u64 vp[MAXA], vs[MAXA];
#pragma unroll
        for (int j = 0; j < MAXA; ++j) vp[j] = (FULL || j < R.K) ? csub(in[p * in_bs + size_t(R.L + j) * n + i] + ldc(R.half_p, j), ldc(R.p_mod, j)) : 0;
        int u = 0;
        if (R.K > 1) u = base_conv_prepare<MAXA, FULL, DS>(R.p2q, vp, vs);
        if constexpr (DS) {  // rolled over the q-limbs, memory operands one limb ahead, the limb's chain unreduced: see the edge kernels
            const u64 *src = in + p * in_bs + i, *ad = addend ? addend + p * add_bs + i : nullptr;
            const u64 *ua = R.p2q.ua + u;
            const int la = FULL ? MAXA : R.p2q.la;
            u64 nx = src[0], na = ad ? ad[0] : 0, nsub = R.K > 1 ? ua[0] : 0;
#pragma unroll 1
            for (int l = 0; l < R.L; ++l) {
                const u64 x = nx, a = na, sub = nsub;
                if (l + 1 < R.L) {
                    nx = src[size_t(l + 1) * n];
                    if (ad) na = ad[size_t(l + 1) * n];
                    if (R.K > 1) nsub = ua[(l + 1) * (la + 1)];
                }
                const u64 q = ldc(R.q_mod, l);
                const DsK m = rns_dsk(q, ldc(R.p2q.b_c, l), R.p2q.pw);
                const u64 vq = x + ldc(R.half_q, l);  // < 2q
                u64 sw;                               // < 3q
                if (R.K == 1) {  // rns.rs:108-111: `*vq -= vp.to_u64()` -> vp % q_i
                    sw = vp[0] - __umul64hi(vp[0], ldc(R.red_mu, l)) * q;
                    sw = csub(csub(sw, q), q);
                } else {
                    sw = base_conv_dot<MAXA, FULL, true, true>(R.p2q, l, vs) + q - sub;
                }
                u64 r = csub(ArithDS<60>::mul(vq + (m.q + m.q2) - sw, ldc4(R.pinv_ds, l), m), q);
                if (ad) r = csub(r + a, q);
                out[p * out_bs + size_t(l) * n + i] = r;
            }
            continue;
        }
        for (int l0 = 0; l0 < R.L; l0 += MAXA) {
#pragma unroll
          for (int ll = 0; ll < MAXA; ++ll) {
            const int l = l0 + ll;
            if (l >= R.L) continue;
            const u64 q = ldc(R.q_mod, l);
            u64 r = rescale_limb<MAXA, FULL, DS>(R, l, q, in[p * in_bs + size_t(l) * n + i], vp, vs, u);
            if (addend) r = csub(r + addend[p * add_bs + size_t(l) * n + i], q);
            out[p * out_bs + size_t(l) * n + i] = r;
          }
        }
    }
}

// ---- the key switch at N = 2^15: the transforms' OUTERMOST layer runs in their neighbours -------------------------------------
// A 2^15 ring does not fit the two-workgroups-per-CU kernel of ntt14w.hpp; as ONE workgroup per CU (R0 = 4) its memory phases
// and its butterflies do not overlap across workgroups and it runs at two thirds of the 2^14 kernel's rate.  But layer 0 of
// the forward transform (util/src/ring/fft.rs:42-52 with m = 1: the pairs (i, i + n/2), ONE twiddle tw[1]) and layer 0 of the
// inverse (fft.rs:62-76, twi[1], then n^-1) are elementwise over such pairs -- exactly the shape of the per-coefficient kernels
// that produce the forward's input (`extend_bases`) and consume the inverse's output (`rescale_k`).  With a thread of those
// kernels owning the pair (i, i + n/2) the layer costs them the products the transform would have spent on it, and what is left
// of every transform is two INDEPENDENT 2^14 sub-transforms (ntt14w PFX form, pb = 1).  All arithmetic is exact mod q_l, so the
// coefficient-domain results are bit-identical.  Two-operand products only (every modulus a 60-bit pseudo-Mersenne prime).
struct EdgeConsts {
    const uint4 *fwd_w;   // [L + K] tw[1] of every modulus (qs then ps)
    const uint4 *inv_n;   // [L + K] n^-1
    const uint4 *inv_nw;  // [L + K] n^-1 twi[1]
};
// (X, Y) <- (X + w Y, X - w Y), canonical in and out (fft.rs:96-101 `dit`)
__device__ __forceinline__ void edge_ct(u64 &X, u64 &Y, const uint4 &w, const DsK &m) {
    const u64 t = csub(ArithDS<60>::mul(Y, w, m), m.q), x = X;
    X = csub(x + t, m.q);
    Y = x >= t ? x - t : x + m.q - t;
}
// (X, Y) <- ((X + Y) n^-1, (X - Y) twi[1] n^-1), canonical in and out (fft.rs:108-113 `dif`, then fft.rs:73-76)
__device__ __forceinline__ void edge_gs(u64 &X, u64 &Y, const uint4 &nv, const uint4 &nw, const DsK &m) {
    const u64 s = X + Y, d = X + m.q - Y;
    X = csub(ArithDS<60>::mul(s, nv, m), m.q);
    Y = csub(ArithDS<60>::mul(d, nw, m), m.q);
}

// How the two kernels below are laid out (measured on the first, fully unrolled version: 230 scalar registers spilled into vector
// lanes and read back -- 480 of 3470 vector instructions -- and, in the rescale, every limb's loads issued right before their use
// behind the previous limb's stores, vmcnt retiring in order: 55 % of the wave cycles parked):
//   * the loop over the OUTPUT limbs is rolled (`#pragma unroll 1`): one row of conversion constants (32 scalar registers) is live
//     at a time, fetched by scalar loads at a run-time row index; the limb vectors stay in registers (indexed by unrolled loops);
//   * everything a limb needs from memory (its coefficients, its addend, its (u A) mod b table entry) is requested ONE LIMB AHEAD,
//     before the current limb's products start and before its results are stored, so no wait ever stands behind a store.

// extend_bases + layer 0 of the forward transform of all la + lb limbs: in [batch][la][n] -> out [batch][la + lb][n]
template <int MAXA, bool FULL>
__global__ void rns_extend_edge_kernel(const u64 *__restrict__ in, size_t in_bs, u64 *__restrict__ out, size_t out_bs, size_t n, size_t batch,
                                       BaseConv C, const uint4 *__restrict__ fwd_w) {
    const size_t h = n >> 1, total = h * batch;
    const int la = FULL ? MAXA : C.la;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / h, i = idx - p * h;
        const u64 *src = in + p * in_bs + i;
        u64 *dst = out + p * out_bs + i;
        u64 v0[MAXA], v1[MAXA], vs0[MAXA], vs1[MAXA];
#pragma unroll
        for (int l = 0; l < MAXA; ++l) {
            v0[l] = (FULL || l < C.la) ? src[size_t(l) * n] : 0;
            v1[l] = (FULL || l < C.la) ? src[size_t(l) * n + h] : 0;
        }
        const int u0 = base_conv_prepare<MAXA, FULL, true>(C, v0, vs0);
        const int u1 = base_conv_prepare<MAXA, FULL, true>(C, v1, vs1);
        const u64 *ua0 = C.ua + u0, *ua1 = C.ua + u1;
        u64 nsub0 = ua0[0], nsub1 = ua1[0];  // output limb 0's table entries fly under the source limbs' butterflies
#pragma unroll
        for (int l = 0; l < MAXA; ++l)
            if (FULL || l < C.la) {
                edge_ct(v0[l], v1[l], ldc4(fwd_w, l), rns_dsk(ldc(C.a_mod, l), ldc(C.a_c, l), C.pw));
                dst[size_t(l) * n] = v0[l];
                dst[size_t(l) * n + h] = v1[l];
            }
#pragma unroll 1
        for (int j = 0; j < C.lb; ++j) {
            const u64 sub0 = nsub0, sub1 = nsub1;
            if (j + 1 < C.lb) { nsub0 = ua0[(j + 1) * (la + 1)]; nsub1 = ua1[(j + 1) * (la + 1)]; }
            const u64 b = ldc(C.b_mod, j);
            const u64 d0 = base_conv_dot<MAXA, FULL, true>(C, j, vs0), d1 = base_conv_dot<MAXA, FULL, true>(C, j, vs1);
            u64 o0 = d0 >= sub0 ? d0 - sub0 : d0 + b - sub0, o1 = d1 >= sub1 ? d1 - sub1 : d1 + b - sub1;
            edge_ct(o0, o1, ldc4(fwd_w, la + j), rns_dsk(b, ldc(C.b_c, j), C.pw));
            dst[size_t(la + j) * n] = o0;
            dst[size_t(la + j) * n + h] = o1;
        }
    }
}

// layer 0 of the inverse transform (+ n^-1) of all L + K limbs + rescale_k: in [batch][L+K][n] -> out [batch][L][n] (+ addend).
// ONE coefficient per lane: lanes 0..31 of a wave own coefficients i0 .. i0 + 31, lanes 32..63 own i0 + n/2 .. -- a lane's partner in
// the layer-0 butterfly is lane ^ 32, reached with ds_bpermute (the LDS crossbar: no vector-ALU instruction, no LDS memory).  Two limb
// vectors per lane (the pair-per-thread form of the extend kernel) need 128 registers here and spill.
// canonical in; out < q + 9c (LAZY) or canonical
template <bool LAZY>
__device__ __forceinline__ void lane_gs(u64 &z, bool hi, const uint4 &nv, const uint4 &nw, const DsK &m) {
    const u64 pz = __shfl_xor(z, 32);
    const u64 t = pz + (hi ? m.q - z : z);  // low lane: x + y; high lane: x + q - y
    const uint4 c{hi ? nw.x : nv.x, hi ? nw.y : nv.y, hi ? nw.z : nv.z, hi ? nw.w : nv.w};
    z = ArithDS<60>::mul(t, c, m);
    if constexpr (!LAZY) z = csub(z, m.q);
}
template <int MAXA, bool FULL>
__global__ void rns_rescale_edge_kernel(const u64 *__restrict__ in, size_t in_bs, u64 *out, size_t out_bs, const u64 *addend, size_t add_bs,
                                        size_t n, size_t batch, RescaleConsts R, EdgeConsts E) {
    const size_t h = n >> 1, wpp = h >> 5, total_w = wpp * batch;  // wpp: waves per polynomial (n >= 64)
    const int lane = threadIdx.x & 63, la = FULL ? MAXA : R.p2q.la;
    const bool hi = lane >> 5;
    for (size_t W = (blockIdx.x * size_t(blockDim.x) + threadIdx.x) >> 6; W < total_w; W += (size_t(gridDim.x) * blockDim.x) >> 6) {
        const size_t p = W / wpp, i = ((W - p * wpp) << 5) + (lane & 31) + (hi ? h : 0);
        const u64 *src = in + p * in_bs + i;
        const u64 *ad = addend ? addend + p * add_bs + i : nullptr;
        u64 *dst = out + p * out_bs + i;
        u64 vp[MAXA], vs[MAXA];
#pragma unroll
        for (int j = 0; j < MAXA; ++j) vp[j] = (FULL || j < R.K) ? src[size_t(R.L + j) * n] : 0;
        u64 nx = src[0], na = ad ? ad[0] : 0;  // q-limb 0 flies under the p-limbs' work
#pragma unroll
        for (int j = 0; j < MAXA; ++j)
            if (FULL || j < R.K) {
                const u64 pm = ldc(R.p_mod, j);
                lane_gs<false>(vp[j], hi, ldc4(E.inv_n, R.L + j), ldc4(E.inv_nw, R.L + j), rns_dsk(pm, ldc(R.p2q.a_c, j), R.p2q.pw));
                vp[j] = csub(vp[j] + ldc(R.half_p, j), pm);
            }
        int u = 0;
        if (R.K > 1) u = base_conv_prepare<MAXA, FULL, true>(R.p2q, vp, vs);
        const u64 *ua = R.p2q.ua + u;
        u64 nsub = R.K > 1 ? ua[0] : 0;
#pragma unroll 1
        for (int l = 0; l < R.L; ++l) {  // rolled, one limb ahead: see above
            u64 x = nx;
            const u64 a = na, sub = nsub;
            if (l + 1 < R.L) {
                nx = src[size_t(l + 1) * n];
                if (ad) na = ad[size_t(l + 1) * n];
                if (R.K > 1) nsub = ua[(l + 1) * (la + 1)];
            }
            const u64 q = ldc(R.q_mod, l);
            const DsK m = rns_dsk(q, ldc(R.p2q.b_c, l), R.p2q.pw);
            // this limb's chain stays unreduced until its last product (which takes any 64-bit multiplicand): the same residues
            lane_gs<true>(x, hi, ldc4(E.inv_n, l), ldc4(E.inv_nw, l), m);  // < q + 9c
            const u64 vq = x + ldc(R.half_q, l);                            // < 2q + 9c
            u64 sw;                                                         // < 3q
            if (R.K == 1) {  // rns.rs:108-111: `*vq -= vp.to_u64()` -> vp % q_i
                sw = vp[0] - __umul64hi(vp[0], ldc(R.red_mu, l)) * q;
                sw = csub(csub(sw, q), q);
            } else {
                sw = base_conv_dot<MAXA, FULL, true, true>(R.p2q, l, vs) + q - sub;
            }
            u64 r = csub(ArithDS<60>::mul(vq + (m.q + m.q2) - sw, ldc4(R.pinv_ds, l), m), q);
            if (ad) r = csub(r + a, q);
            dst[size_t(l) * n] = r;
        }
    }
}

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
