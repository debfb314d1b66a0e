// Unreduced dot products modulo pseudo-Mersenne primes: the arithmetic of the RNS base conversions (rns_kernels.hpp).
//
// `Rns::extend_bases` and `rescale_k` (util/src/ring/rns.rs:103-132, 331-345) are, per coefficient, a small dense contraction
//     out_j = sum_i M[j][i] vs_i  (+ terms in u and in the limb's own value)      (mod b_j),     i < la <= 32, vs_i < 2^60,
// with constants M[j][i] < b_j.  Reduced term by term that is one modular product per (j, i) -- six multiply-adds and a fold each in
// the two-operand form of arith.hpp.  Here NOTHING is reduced until the sum is complete.  Every operand is cut at 30 bits,
//     M = k0 + k1 2^30,   vs = y0 + y1 2^30            (k0, k1, y0, y1 < 2^30  because every modulus is below 2^60),
// so that eight products of one kind fit a 64-bit accumulator without a carry,
//     S00 = sum k0 y0 < 2^63,   S11 = sum k1 y1 < 2^63,   S01 = sum (k0 y1 + k1 y0) < 2^64 - 2^35,
// and the cross sum comes from Karatsuba's identity, S01 = SK - S00 - S11 with SK = sum (k0 + k1)(y0 + y1) taken mod 2^64 (the
// true S01 is below 2^64, so the wrapped difference is exact): THREE v_mad_u64_u32 per term, the multiplier a scalar register.
// One reduction per output then folds  X = S00 + S01 2^30 + S11 2^60  with 2^60 = c60 (mod b), c60 = (2^B - b) 2^(60 - B) < 2^27:
// about 17 instructions for up to eight terms where the term-by-term form spent 8 x 11.
//
// Everything here is plain integer arithmetic on values whose bounds are stated; the file compiles for the host too
// (tests/test_pmdot_cpu.py builds tools/pmdot_host_test.cpp against it and checks every function against unsigned __int128).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define FHE_PD __host__ __device__ __forceinline__
#else
#define FHE_PD inline
#endif

namespace fhe {
namespace pd {

typedef unsigned long long u64;

// what all moduli of one context share: their common bit length B (34 <= B <= 60), b = 2^B - c with c <= 2^(B-33)
struct Uni {
    unsigned sh;     // B - 32
    unsigned mhi;    // 2^(B-32) - 1: mask of the high word below bit B
    unsigned pw;     // 2^(B-31)      } multipliers handed over as VALUES the compiler cannot see: as literals it turns the
    unsigned two30;  // 2^30          } multiply-add into a 64-bit shift plus a 64-bit add
};
FHE_PD Uni make_uni(int B) { return Uni{(unsigned)(B - 32), (1u << (B - 32)) - 1, 1u << (B - 31), 1u << 30}; }

// a residue below 2^60 cut at 30 bits
struct Y3 {
    unsigned y0, y1, yk;  // y = y0 + y1 2^30, yk = y0 + y1 (< 2^31)
};
FHE_PD Y3 split30(u64 y) {
    const unsigned y0 = (unsigned)y & 0x3fffffffu, y1 = (unsigned)(y >> 30);
    return Y3{y0, y1, y0 + y1};
}

// x mod~ b at bit B: < 2^B + (x >> B) c
FHE_PD u64 fold(u64 x, unsigned c, const Uni &U) {
    const unsigned hi = (unsigned)(x >> 32), h = hi >> U.sh;
    const u64 base = ((u64)(hi & U.mhi) << 32) | (unsigned)x;
    return (u64)h * c + base;
}
FHE_PD u64 csub(u64 x, u64 m) { return x >= m ? x - m : x; }

// w y mod b, unreduced, for a FIXED operand in the two-operand form {a0, a1, b0, b1} of arith.hpp (ArithDS: w = a0 + a1 2^(B-31),
// w 2^32 mod b = b0 + b1 2^(B-31)) and ANY 64-bit y: < 2^(B+3).  c2 = 2 c.
FHE_PD u64 ds_mul_raw(u64 y, unsigned a0, unsigned a1, unsigned b0, unsigned b1, unsigned c2, const Uni &U) {
    const unsigned y0 = (unsigned)y, y1 = (unsigned)(y >> 32);
    const u64 s1 = (u64)a1 * y0 + (u64)b1 * y1;
    const u64 s0 = (u64)a0 * y0 + (u64)b0 * y1;
    const u64 v = (u64)(unsigned)s1 * U.pw + s0;
    return (u64)(unsigned)(s1 >> 32) * c2 + v;
}

// X = s00 + (s01a + s01b) 2^30 + s11 2^60  (mod b), b = 2^B - c, c60 = c 2^(60-B) < 2^27.
// Requires  s00 < 11 * 2^60,  s11 < 10 * 2^60,  s01b < 2^62  (s01a: any 64-bit value).
// Returns a value < 2^B + 2^(64-B) c congruent to X (one conditional subtraction away from canonical).
//   H  = s11 + (s01a >> 30) + (s01b >> 30)                   < 10 * 2^60 + 2^34 + 2^32
//   T  = s00 + ((s01a mod 2^30) + (s01b mod 2^30)) 2^30      < 11 * 2^60 + 2^61 = 13 * 2^60
//   X  = T + H 2^60 = T + H c60 = T + lo32(H) c60 + hi32(H) c60 2^32,      g = hi32(H) c60 < 2^59
//   g 2^32 = (g mod 2^28) 2^32 + (g >> 28) 2^60 = (g mod 2^28) 2^32 + (g >> 28) c60
//   r  = T + lo32(H) c60 [< 2^59] + (g mod 2^28) 2^32 [< 2^60] + (g >> 28) c60 [< 2^58]   < 15 * 2^60 < 2^64
FHE_PD u64 reduce_lazy(u64 s00, u64 s01a, u64 s01b, u64 s11, unsigned c, unsigned c60, const Uni &U) {
    const unsigned m0 = ((unsigned)s01a & 0x3fffffffu) + ((unsigned)s01b & 0x3fffffffu);
    const u64 H = s11 + (s01a >> 30) + (s01b >> 30);
    const u64 T = (u64)m0 * U.two30 + s00;
    const u64 r1 = (u64)(unsigned)H * c60 + T;
    const u64 g = (u64)(unsigned)(H >> 32) * c60;
    const u64 r2 = r1 + ((u64)((unsigned)g & 0x0fffffffu) << 32);
    const u64 r3 = (u64)(unsigned)(g >> 28) * c60 + r2;
    return fold(r3, c, U);
}

}  // namespace pd
}  // namespace fhe
