// Row T: TFHE torus path (util/src/torus.rs, util/src/misc/decompose.rs:66-135, util/src/ring/fft/c64.rs,
// scheme/tfhe/src/{tggsw,tglwe,tlwe,bootstrapping}.rs), k = 1.
//
// The reference multiplies torus polynomials (Z_{2^64}[X]/(X^N+1)) with an f64 FFT whose low bits carry rounding
// noise (its own test bounds it by 2^(64 + log_b + log_n - 53), c64.rs:186-208).  Here the product is EXACT: both
// operands are read as signed 64-bit integers (exactly what to_c64_twisted does, c64.rs:23-27), the integer negacyclic
// product is computed modulo two 60-bit NTT primes with the same wave-private transforms as the FHEW kernels, and the
// Chinese remainder of the two residues is reduced mod 2^64.  Exact needs |coefficient| < p0 p1 / 2 ~ 2^119: true for
// every gadget product (digits <= 2^(log_b - 1), (k+1) d N 2^(62 + log_b) < 2^119 is checked at key preparation).
// Never less exact than the reference (SURVEY.md section 8(a) row T acceptance rule), so decode-level results agree.
#pragma once
#include "fhew_kernels.hpp"

namespace fhe {

// ---- T64 gadget decomposition (decompose.rs:66-81 `new`, 114-135) ------------------------------------------------
struct TDecomp {
    u64 rnd;   // (1 << rounding_bits) >> 1
    u64 mask;  // 2^log_b - 1
    int log_b, d, rb;
};

__device__ __forceinline__ u64 tdecomp_init(u64 v, const TDecomp &P) { return (v + P.rnd) >> P.rb; }  // rounding_shr (wrapping add)

__device__ __forceinline__ u64 tdecomp_next(u64 &c, const TDecomp &P) {
    const u64 limb = c & P.mask;
    c >>= P.log_b;
    const u64 carry = (((limb - 1) | c) & limb) >> (P.log_b - 1);
    c += carry;
    return limb - (carry << P.log_b);  // signed digit as a two's-complement u64
}

// in [polys][n] -> out [polys][d][n]
FHE_HEADER_KERNEL void torus_decompose_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t n, size_t polys, TDecomp P) {
    const size_t total = n * polys;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n, i = idx - p * n;
        u64 c = tdecomp_init(in[idx], P);
        for (int j = 0; j < P.d; ++j) out[(p * P.d + j) * n + i] = tdecomp_next(c, P);
    }
}

// signed 64-bit value -> canonical residue mod p (2^59 < p < 2^60)
__device__ __forceinline__ u64 signed_residue(u64 v, u64 p) {
    const bool neg = (long long)v < 0;
    u64 m = neg ? (0 - v) : v;                 // |v| <= 2^63 < 16 p
    m = csub(m, 8 * p); m = csub(m, 4 * p); m = csub(m, 2 * p); m = csub(m, p);
    return (neg && m) ? p - m : m;
}

// rows [rows][n] signed torus values -> residues mod the prime of descs[prime]
FHE_HEADER_KERNEL void torus_residue_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t count, u64 p) {
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < count; idx += size_t(gridDim.x) * blockDim.x)
        out[idx] = signed_residue(in[idx], p);
}

struct TorusConsts {
    const ModDesc *descs;     // two primes p0, p1 (pseudo-Mersenne, 60 bits)
    u64 p0, p1;
    u64 inv01, inv01_s;       // p0^-1 mod p1 and its Shoup companion
    u64 P_lo;                 // (p0 p1) mod 2^64
    u64 Ph_hi, Ph_lo;         // floor(p0 p1 / 2)
    Barrett B0, B1;           // unused by the pseudo-Mersenne multiply-accumulate, kept for the policy interface
};

// x = r0 (mod p0), x = r1 (mod p1), |x| < p0 p1 / 2  ->  x mod 2^64
__device__ __forceinline__ u64 crt2_mod64(u64 r0, u64 r1, const TorusConsts &T) {
    const u64 r0m = csub(r0, T.p1);                             // p0 < 2 p1
    const u64 diff = r1 >= r0m ? r1 - r0m : r1 + T.p1 - r0m;
    const u64 t = csub(mul_shoup_lazy(diff, T.inv01, T.inv01_s, T.p1), T.p1);
    const u64 lo_prod = T.p0 * t, hi_prod = __umul64hi(T.p0, t);
    const u64 lo = lo_prod + r0;
    const u64 hi = hi_prod + (lo < lo_prod);                    // x = r0 + p0 t in [0, p0 p1)
    const bool upper = hi > T.Ph_hi || (hi == T.Ph_hi && lo > T.Ph_lo);
    return upper ? lo - T.P_lo : lo;                            // centred representative, mod 2^64
}

// One gadget product pass for ONE prime over register-resident operands (coefficient layout): outputs the residues of
// sum_l rows[l].a * limb_l and sum_l rows[l].b * limb_l (exact integers) mod that prime, coefficient layout.
template <class W>
__device__ __forceinline__ void team_torus_gadget(const u64 (&da)[W::E], const u64 (&db)[W::E],
                                                  const u64 *__restrict__ rows, const TDecomp &P, u64 p, int lane, u64 *lds, const Barrett &B,
                                                  const typename ArithPM<60>::K &k, u64 (&sa)[W::E],
                                                  u64 (&sb)[W::E]) {
    using A = ArithPM<60>;
    constexpr int E = W::E;
    RingConsts K;
    K.desc = nullptr;
    K.B = B;
    typename A::MacAcc ma[E], mb[E];
    u64 st[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { ma[e] = mb[e] = A::mac_zero(); st[e] = tdecomp_init(da[e], P); }
    KeyRow<W> kr;  // one limb ahead (fhew_kernels.hpp)
    load_row<W>(kr, rows, lane);
#pragma unroll 1
    for (int j = 0; j < 2 * P.d; ++j) {
        if (j == P.d) {
#pragma unroll
            for (int e = 0; e < E; ++e) st[e] = tdecomp_init(db[e], P);
        }
        u64 x[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const u64 dg = tdecomp_next(st[e], P);
            x[e] = (long long)dg < 0 ? p - (0 - dg) : dg;  // |digit| <= 2^(log_b-1) < p
        }
        fwd_run<A, typename W::C, W::LOG_N, W::LOG_E, 0, true, W::WAVE>(x, lane, nullptr, lds, true, k);
        mac_row<A, W>(x, ma, mb, kr, j, K, k);
        if (j + 1 < 2 * P.d) load_row<W>(kr, rows + size_t(j + 1) * 2 * W::N, lane);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) { sa[e] = A::mac_finish(ma[e], k); sb[e] = A::mac_finish(mb[e], k); }
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
        inv_run<A, typename W::C, W::LOG_N, W::LOG_E, W::LOG_N, true, W::WAVE>(sa, lane, nullptr, lds, true, k);
#pragma unroll
        for (int e = 0; e < E; ++e) { const u64 t = sa[e]; sa[e] = sb[e]; sb[e] = t; }
    }
}

// c <- c * X^r on the torus (ring.rs:299-313; negation = wrapping_neg), r in [0, 2N), through the team's LDS image
template <class W>
__device__ __forceinline__ void team_torus_rotate(u64 (&c)[W::E], unsigned r, int lane, u64 *lds) {
    constexpr int E = W::E, N = W::N;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned pos = (unsigned(coef_index<W>(lane, e)) + r) & (2 * N - 1);
        lds[lds_phys(pos & (N - 1))] = pos < N ? c[e] : 0 - c[e];
    }
    exchange_sync<W::WAVE>();
#pragma unroll
    for (int e = 0; e < E; ++e) c[e] = lds[lds_phys(coef_index<W>(lane, e))];
    exchange_sync<W::WAVE>();
}

// (xa, xb) <- external_product(key, (da, db)), exact: both primes, then the Chinese remainder mod 2^64
// (scheme/tfhe/src/tggsw.rs:100-112 for k = 1); everything stays in the team's registers
template <class W>
__device__ __forceinline__ void team_torus_external_product(const u64 (&da)[W::E], const u64 (&db)[W::E],
                                                            const u64 *__restrict__ rows0, const u64 *__restrict__ rows1, const TDecomp &P,
                                                            const TorusConsts &T, int lane, u64 *lds, u64 (&xa)[W::E],
                                                            u64 (&xb)[W::E]) {
    using A = ArithPM<60>;
    constexpr int E = W::E;
    // the first prime's residues wait in the team's LDS parking area (each lane its own slots: no synchronisation) while the
    // second prime's pass needs the registers
    u64 *park = lds + W::PN;
    {
        const typename A::K k0 = A::make(T.descs[0], W::LOG_N, 0, 0);
        team_torus_gadget<W>(da, db, rows0, P, T.p0, lane, lds, T.B0, k0, xa, xb);
#pragma unroll
        for (int e = 0; e < E; ++e) { park[e * W::TEAM + lane] = xa[e]; park[(E + e) * W::TEAM + lane] = xb[e]; }
    }
    {
        const typename A::K k1 = A::make(T.descs[1], W::LOG_N, 0, 0);
        team_torus_gadget<W>(da, db, rows1, P, T.p1, lane, lds, T.B1, k1, xa, xb);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        xa[e] = crt2_mod64(park[e * W::TEAM + lane], xa[e], T);
        xb[e] = crt2_mod64(park[(E + e) * W::TEAM + lane], xb[e], T);
    }
}

// One CMUX step of the blind rotation on a register-resident accumulator (tggsw.rs:114-121, bootstrapping.rs:94-95):
// acc <- acc + external_product(key, acc X^r - acc); r = 0 leaves acc untouched (the external product of zero is exactly zero)
template <class W>
__device__ __forceinline__ void team_torus_cmux(u64 (&ca)[W::E], u64 (&cb)[W::E], unsigned r,
                                                const u64 *__restrict__ rows0, const u64 *__restrict__ rows1, const TDecomp &P,
                                                const TorusConsts &T, int lane, u64 *lds) {
    constexpr int E = W::E;
    if (r == 0) return;  // team-uniform
    u64 da[E], db[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { da[e] = ca[e]; db[e] = cb[e]; }
    team_torus_rotate<W>(da, r, lane, lds);
    team_torus_rotate<W>(db, r, lane, lds);
#pragma unroll
    for (int e = 0; e < E; ++e) { da[e] -= ca[e]; db[e] -= cb[e]; }
    u64 xa[E], xb[E];
    team_torus_external_product<W>(da, db, rows0, rows1, P, T, lane, lds, xa, xb);
#pragma unroll
    for (int e = 0; e < E; ++e) { ca[e] += xa[e]; cb[e] += xb[e]; }
}

// scheme/tfhe/src/tggsw.rs:100-121 for k = 1, one team (fhew_kernels.hpp: WaveRing) per ciphertext.
//   rot == nullptr: (a, b) <- external_product(key, (a, b))
//   rot != nullptr: one CMUX step: (a, b) <- (a, b) + external_product(key, (a, b) X^r - (a, b)), r = rot[ct * rot_stride] mod 2N
template <class W>
__global__ __launch_bounds__(W::THREADS, W::MIN_WAVES) void torus_cmux_kernel(
    u64 *__restrict__ acc_a, u64 *__restrict__ acc_b, unsigned batch, const u64 *__restrict__ rows0, const u64 *__restrict__ rows1,
    TDecomp P, const u64 *__restrict__ rot, size_t rot_stride, TorusConsts T) {
    constexpr int E = W::E, N = W::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const unsigned ct = blockIdx.x * W::TEAMS + team;
    if (ct >= batch) return;
    u64 *lds = reinterpret_cast<u64 *>(smem_raw) + team * W::TORUS_LDS_WORDS;
    u64 *ga = acc_a + size_t(ct) * N, *gb = acc_b + size_t(ct) * N;
    u64 ca[E], cb[E];
    wave_load<W>(ca, ga, lane);
    wave_load<W>(cb, gb, lane);
    if (rot != nullptr) {
        team_torus_cmux<W>(ca, cb, unsigned(rot[size_t(ct) * rot_stride]) & (2 * N - 1), rows0, rows1, P, T, lane, lds);
    } else {
        u64 xa[E], xb[E];
        team_torus_external_product<W>(ca, cb, rows0, rows1, P, T, lane, lds, xa, xb);
#pragma unroll
        for (int e = 0; e < E; ++e) { ca[e] = xa[e]; cb[e] = xb[e]; }
    }
    wave_store<W>(ca, ga, lane);
    wave_store<W>(cb, gb, lane);
}

// scheme/tfhe/src/bootstrapping.rs:84-96 `blind_rotate` (k = 1), the whole fold in ONE launch: acc = (0, v X^-b), then
// acc <- cmux(brk_i, acc, acc X^(a_i)) for i = 0 .. n_lwe-1, the accumulator in the team's registers throughout.
// rows0 / rows1: the key set [n_lwe][2d][2][N] per prime (key_perm layout); a_tilde [batch][n_lwe], b_tilde [batch] mod 2N.
template <class W>
__global__ __launch_bounds__(W::THREADS, W::MIN_WAVES) void torus_blind_rotate_kernel(
    const u64 *__restrict__ v, const u64 *__restrict__ a_tilde, const u64 *__restrict__ b_tilde, unsigned n_lwe, unsigned batch,
    const u64 *__restrict__ rows0, const u64 *__restrict__ rows1, TDecomp P, TorusConsts T, u64 *__restrict__ out_a, u64 *__restrict__ out_b) {
    constexpr int E = W::E, N = W::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const unsigned ct = blockIdx.x * W::TEAMS + team;
    if (ct >= batch) return;
    u64 *lds = reinterpret_cast<u64 *>(smem_raw) + team * W::TORUS_LDS_WORDS;
    u64 ca[E], cb[E];
    wave_load<W>(cb, v, lane);
#pragma unroll
    for (int e = 0; e < E; ++e) ca[e] = 0;
    team_torus_rotate<W>(cb, (2 * N - (unsigned(b_tilde[ct]) & (2 * N - 1))) & (2 * N - 1), lane, lds);  // (0, v).rotate(-b)
    const size_t per = size_t(2 * P.d) * 2 * N;
    const u64 *a = a_tilde + size_t(ct) * n_lwe;
#pragma unroll 1
    for (unsigned i = 0; i < n_lwe; ++i) {
        const unsigned r = __builtin_amdgcn_readfirstlane(unsigned(a[i]) & (2 * N - 1));
        team_torus_cmux<W>(ca, cb, r, rows0 + i * per, rows1 + i * per, P, T, lane, lds);
    }
    wave_store<W>(ca, out_a + size_t(ct) * N, lane);
    wave_store<W>(cb, out_b + size_t(ct) * N, lane);
}

// exact torus product building blocks for fhe_torus_mul: c = a * b with |b| small (two-prime CRT)
FHE_HEADER_KERNEL void torus_crt_kernel(const u64 *__restrict__ r /* [batch][2][n] */, u64 *__restrict__ out, size_t n, size_t batch, TorusConsts T) {
    const size_t total = n * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n, i = idx - p * n;
        out[idx] = crt2_mod64(r[(p * 2) * n + i], r[(p * 2 + 1) * n + i], T);
    }
}

// a, b [batch][n] signed torus values -> ra, rb [batch][2][n] residues
FHE_HEADER_KERNEL void torus_residue2_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t n, size_t batch, u64 p0, u64 p1) {
    const size_t total = n * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n, i = idx - p * n;
        const u64 v = in[idx];
        out[(p * 2) * n + i] = signed_residue(v, p0);
        out[(p * 2 + 1) * n + i] = signed_residue(v, p1);
    }
}

// ra <- ra (.) rb per prime; ra [batch][2][n], rb [b_rows][2][n] cycled over the batch (b_rows = 1: one multiplier for all)
FHE_HEADER_KERNEL void torus_pointwise_kernel(u64 *__restrict__ ra, const u64 *__restrict__ rb, size_t n, size_t batch, size_t b_rows, Barrett B0,
                                              Barrett B1) {
    const size_t total = 2 * n * batch, period = 2 * n * b_rows;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const bool second = (idx / n) & 1;
        ra[idx] = mulmod_barrett(ra[idx], rb[idx % period], second ? B1 : B0);
    }
}

// scheme/tfhe/src/tglwe.rs:115-127 (k = 1): [batch][n] a, b -> TLWE a [batch][n], b [batch]
FHE_HEADER_KERNEL void tglwe_sample_extract_kernel(const u64 *__restrict__ ct_a, const u64 *__restrict__ ct_b, unsigned n, size_t batch, unsigned i,
                                            u64 *__restrict__ out_a, u64 *__restrict__ out_b) {
    const size_t total = size_t(n) * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n;
        const unsigned j = unsigned(idx - p * n);
        // a[..i+1].rev() ++ a[i+1..].rev().map(neg)
        out_a[idx] = j <= i ? ct_a[p * n + (i - j)] : 0 - ct_a[p * n + (n - 1 - (j - i - 1))];
        if (j == 0) out_b[p] = ct_b[p * n + i];
    }
}

// scheme/tfhe/src/tlwe.rs:144-153: one thread per (ciphertext, output coefficient); ksk_a [n_in*d][n_out], ksk_b [n_in*d]
// (row index j * n_in + i: digit-major, as `decompose(a).flatten()` orders the limbs); output index n_out carries b
FHE_HEADER_KERNEL void tlwe_key_switch_kernel(const u64 *__restrict__ ct_a, const u64 *__restrict__ ct_b, unsigned n_in, unsigned n_out,
                                       size_t batch, const u64 *__restrict__ ksk_a, const u64 *__restrict__ ksk_b, TDecomp P,
                                       u64 *__restrict__ out_a, u64 *__restrict__ out_b) {
    const size_t total = size_t(n_out + 1) * batch;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / (n_out + 1);
        const unsigned kcol = unsigned(idx - p * (n_out + 1));
        u64 acc = 0;
        for (unsigned i = 0; i < n_in; ++i) {
            u64 c = tdecomp_init(ct_a[p * n_in + i], P);
            for (int j = 0; j < P.d; ++j) {
                const u64 dg = tdecomp_next(c, P);
                const size_t row = size_t(j) * n_in + i;
                acc += (kcol < n_out ? ksk_a[row * n_out + kcol] : ksk_b[row]) * dg;  // wrapping: arithmetic mod 2^64
            }
        }
        if (kcol < n_out) out_a[p * n_out + kcol] = acc;
        else out_b[p] = acc + ct_b[p];
    }
}

// util/src/torus-side rounding_shr used by bootstrapping.rs:99-104 `mod_switch`
// out = x + y (sub = 0) or x - y (sub = 1), wrapping mod 2^64 (util/src/torus.rs:40-60)
FHE_HEADER_KERNEL void torus_addsub_kernel(const u64 *__restrict__ x, const u64 *__restrict__ y, u64 *__restrict__ out, size_t count, int sub) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < count; i += size_t(gridDim.x) * blockDim.x) out[i] = sub ? x[i] - y[i] : x[i] + y[i];
}
// scheme/tfhe/src/tglwe.rs:61-66 `rotate(i)` = every polynomial times X^i (util/src/ring.rs:299-313, 380-406 on T64): k = i mod 2N,
// rotate right by k mod N, negate what wraps (and everything once more if k >= N).  in, out [polys][n], in != out
FHE_HEADER_KERNEL void torus_monomial_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, unsigned n, size_t polys, unsigned k) {
    const size_t total = size_t(n) * polys;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t p = idx / n;
        const unsigned i = unsigned(idx - p * n), j = (i + k) & (2 * n - 1);  // X^i X^k = X^j, X^n = -1
        const u64 v = in[idx];
        out[p * n + (j & (n - 1))] = j >= n ? (u64)0 - v : v;
    }
}

FHE_HEADER_KERNEL void torus_rounding_shr_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t count, int bits) {
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < count; idx += size_t(gridDim.x) * blockDim.x)
        out[idx] = (in[idx] + ((u64(1) << bits) >> 1)) >> bits;
}

}  // namespace fhe
