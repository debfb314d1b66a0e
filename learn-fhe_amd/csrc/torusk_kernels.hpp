// Row T for ANY TGLWE rank k (the reference's `TglweParam::n`, scheme/tfhe/src/tglwe.rs:11-35; its own TGLWE / TGGSW tests run at
// k = 2, N = 256: tglwe.rs:138-166, tggsw.rs:134-181).  The k = 1 entry points keep one ciphertext in the registers of a team
// (torus_kernels.hpp, torus30_kernels.hpp); a rank-k ciphertext has k + 1 polynomials and (k + 1) d gadget limbs against a
// (k + 1) d x (k + 1) key matrix, which no longer fits one team's registers at the parameter sets that use k > 1 -- here the same
// exact arithmetic (two 60-bit primes, CRT back to Z/2^64) is composed from streaming kernels around the batched transforms:
//     limbs (decompose + residues) -> forward NTT -> limb-by-key multiply-accumulate -> inverse NTT -> CRT (+ addend).
// Layouts: a TGLWE ciphertext is [k + 1][n] torus words (a_0 .. a_{k-1}, b); residues / evaluations are [..][2 primes][n].
#pragma once
#include "torus_kernels.hpp"

namespace fhe {

// coefficient j of poly * X^r (0 <= r < 2n): torus_monomial_kernel's map read from the output side
__device__ __forceinline__ u64 torusk_rot_read(const u64 *__restrict__ poly, unsigned n, unsigned j, unsigned r) {
    const unsigned t = (j - r) & (2 * n - 1);
    const u64 v = poly[t & (n - 1)];
    return t >= n ? (u64)0 - v : v;
}

// Source polynomials -> residues mod (p0, p1) of their gadget digits (P.d > 0: decompose.rs:114-135, digit-major per polynomial as
// tggsw.rs:106-108 `flat_map(decompose)` orders them) or of the signed values themselves (P.d == 0).
//   src [batch][src_polys][n], the first `polys` polynomials of each are used -> out [batch][polys * max(d, 1)][2][n]
//   sub != null: the source is src - sub (tggsw.rs:120 `ct1 - ct0`), same layout
//   rot != null: the source is src * X^rot[c * rot_stride] - src (bootstrapping.rs:93-95: `acc.rotate(a) - acc` inside the CMUX)
FHE_HEADER_KERNEL void torusk_limbs_kernel(const u64 *__restrict__ src, const u64 *__restrict__ sub, u64 *__restrict__ out, unsigned n, unsigned polys,
                                           unsigned src_polys, size_t batch, TDecomp P, u64 p0, u64 p1, const u64 *__restrict__ rot,
                                           size_t rot_stride) {
    const size_t total = size_t(n) * polys * batch;
    const unsigned d = P.d > 0 ? (unsigned)P.d : 1u;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const unsigned i = unsigned(idx % n);
        const size_t pc = idx / n, c = pc / polys;
        const unsigned pl = unsigned(pc - c * polys);
        const u64 *poly = src + (c * src_polys + pl) * n;
        u64 v = poly[i];
        if (rot) v = torusk_rot_read(poly, n, i, (unsigned)rot[c * rot_stride] & (2 * n - 1)) - v;
        if (sub) v -= sub[(c * src_polys + pl) * n + i];
        u64 *o = out + ((c * polys + pl) * d) * 2 * size_t(n) + i;
        if (P.d > 0) {
            u64 st = tdecomp_init(v, P);
            for (unsigned j = 0; j < d; ++j) {
                const u64 dig = tdecomp_next(st, P);
                o[(2 * j) * size_t(n)] = signed_residue(dig, p0);
                o[(2 * j + 1) * size_t(n)] = signed_residue(dig, p1);
            }
        } else {
            o[0] = signed_residue(v, p0);
            o[n] = signed_residue(v, p1);
        }
    }
}

// out[c][col][pr][i] = sum_r limbs[c][r][pr][i] * key[r][col][pr][i] mod p_pr   (tggsw.rs:109-110: `ct0.a().dot(&ct1_limbs)`,
// `ct0.b().dot(&ct1_limbs)` in the evaluation domain).  limbs [batch][rows][2][n], key [rows][cols][2][n], out [batch][cols][2][n];
// canonical in, canonical out.
FHE_HEADER_KERNEL void torusk_mac_kernel(const u64 *__restrict__ limbs, const u64 *__restrict__ key, u64 *__restrict__ out, unsigned n, unsigned rows,
                                         unsigned cols, size_t batch, Barrett B0, Barrett B1) {
    const size_t total = size_t(2) * n * cols * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const unsigned i = unsigned(idx % n);
        size_t rest = idx / n;
        const unsigned pr = unsigned(rest & 1);
        rest >>= 1;
        const size_t c = rest / cols;
        const unsigned col = unsigned(rest - c * cols);
        const Barrett &B = pr ? B1 : B0;
        const u64 *l = limbs + (c * rows * 2 + pr) * size_t(n) + i;
        const u64 *k = key + (size_t(col) * 2 + pr) * n + i;
        u64 acc = 0;
        for (unsigned r = 0; r < rows; ++r) acc = csub(acc + mulmod_barrett(l[size_t(r) * 2 * n], k[size_t(r) * cols * 2 * n], B), B.q);
        out[idx] = acc;
    }
}

// CRT back to Z/2^64 with the additions the callers need folded in:
//   r [batch][cnt][2][n] -> out polynomial (c * out_polys + out_off + j), j < cnt
//   same: addend in out's own layout (CMUX: `ct0 + ...`, may be `out` itself), e: [batch][cnt][n] noise, pt: [pt_rows][cnt][n] cycled
FHE_HEADER_KERNEL void torusk_crt_kernel(const u64 *__restrict__ r, u64 *out, unsigned n, unsigned cnt, unsigned out_polys, unsigned out_off, size_t batch,
                                         TorusConsts T, const u64 *same, const u64 *__restrict__ e, const u64 *__restrict__ pt, size_t pt_rows) {
    const size_t total = size_t(n) * cnt * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const unsigned i = unsigned(idx % n);
        const size_t pc = idx / n, c = pc / cnt;
        const unsigned j = unsigned(pc - c * cnt);
        u64 v = crt2_mod64(r[(pc * 2) * n + i], r[(pc * 2 + 1) * n + i], T);
        const size_t o = (c * out_polys + out_off + j) * n + i;
        if (same) v += same[o];
        if (e) v += e[idx];
        if (pt) v += pt[((c % pt_rows) * cnt + j) * n + i];
        out[o] = v;
    }
}

// bootstrapping.rs:90-92: acc = (0, .., 0, v).rotate(-b~) per ciphertext; acc [batch][k + 1][n]
FHE_HEADER_KERNEL void torusk_init_acc_kernel(const u64 *__restrict__ v, const u64 *__restrict__ b_tilde, u64 *__restrict__ acc, unsigned n, unsigned k1,
                                              size_t batch) {
    const size_t total = size_t(n) * k1 * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const unsigned i = unsigned(idx % n);
        const size_t pc = idx / n, c = pc / k1;
        const unsigned pl = unsigned(pc - c * k1);
        const unsigned rneg = (0u - (unsigned)b_tilde[c]) & (2 * n - 1);
        acc[idx] = pl + 1 == k1 ? torusk_rot_read(v, n, i, rneg) : 0;
    }
}

// tglwe.rs:115-127 for rank k: a = concat_j (a_j[..=i].rev() ++ a_j[i+1..].rev().neg()), b = b[i]; ct [batch][k + 1][n] ->
// out_a [batch][k n], out_b [batch]
FHE_HEADER_KERNEL void tglwek_sample_extract_kernel(const u64 *__restrict__ ct, unsigned n, unsigned k, size_t batch, unsigned i, u64 *__restrict__ out_a,
                                                    u64 *__restrict__ out_b) {
    const size_t total = size_t(n) * k * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const unsigned j = unsigned(idx % n);
        const size_t pc = idx / n, c = pc / k;
        const unsigned pl = unsigned(pc - c * k);
        const u64 *a = ct + (c * (k + 1) + pl) * n;
        out_a[idx] = j <= i ? a[i - j] : (u64)0 - a[n + i - j];
        if (j == 0 && pl == 0) out_b[c] = ct[(c * (k + 1) + k) * n + i];
    }
}

// tggsw.rs:80-87 for rank k: rows [count][(k + 1) d][k + 1][n]; row (col d + j) gets pt 2^(rb + j log_b) added to its polynomial
// `col` (col < k: a_col, col = k: b)
FHE_HEADER_KERNEL void tggswk_add_gadget_kernel(u64 *__restrict__ rows, const u64 *__restrict__ pt, unsigned n, unsigned k1, size_t count, int d, int rounding_bits,
                                                int log_b) {
    const size_t total = size_t(n) * d * k1 * count;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const unsigned i = unsigned(idx % n);
        size_t rest = idx / n;
        const unsigned j = unsigned(rest % d);
        rest /= d;
        const unsigned col = unsigned(rest % k1);
        const size_t c = rest / k1;
        rows[(((c * k1 + col) * d + j) * k1 + col) * n + i] += pt[c * n + i] << (rounding_bits + (int)j * log_b);
    }
}

}  // namespace fhe
