// Device-side 64-bit modular arithmetic for gfx950.  Everything stays in registers.
//
// Lazy (Harvey) representation: forward-transform values live in [0, 4q), inverse-transform values
// in [0, 2q); a multiplication by a FIXED operand w uses Shoup's precomputed quotient
// ws = floor(w * 2^64 / q) and returns a value in [0, 2q) for ANY 64-bit multiplicand.  Requires
// q < 2^62.  Outputs are canonicalised to [0, q) before they leave a kernel, so results are
// bit-identical to the reference's `(a * b) as u128 % q` arithmetic (util/src/zq.rs:156-196).
#pragma once
#include <hip/hip_runtime.h>

namespace fhe {

typedef unsigned long long u64;

struct __attribute__((aligned(16))) TwPair {
    u64 w;   // twiddle, in [0, q)
    u64 ws;  // floor(w * 2^64 / q)
};

// w * y mod q up to one multiple of q: result in [0, 2q), y arbitrary
__device__ __forceinline__ u64 mul_shoup_lazy(u64 y, u64 w, u64 ws, u64 q) {
    u64 hi = __umul64hi(ws, y);
    return w * y - hi * q;
}

__device__ __forceinline__ u64 csub(u64 x, u64 m) { return x >= m ? x - m : x; }

// Cooley-Tukey (DIT) butterfly, util/src/ring/fft.rs:92-98: (X, Y) <- (X + wY, X - wY).
// in: X, Y in [0, 4q)   out: X, Y in [0, 4q)
__device__ __forceinline__ void ct_bfly(u64 &X, u64 &Y, u64 w, u64 ws, u64 q, u64 q2) {
    u64 x = csub(X, q2);
    u64 t = mul_shoup_lazy(Y, w, ws, q);
    X = x + t;
    Y = x - t + q2;
}

// Gentleman-Sande (DIF) butterfly, util/src/ring/fft.rs:100-106: (X, Y) <- (X + Y, (X - Y) w).
// in: X, Y in [0, 2q)   out: X, Y in [0, 2q)
__device__ __forceinline__ void gs_bfly(u64 &X, u64 &Y, u64 w, u64 ws, u64 q, u64 q2) {
    u64 s = csub(X + Y, q2);
    u64 d = X - Y + q2;
    X = s;
    Y = mul_shoup_lazy(d, w, ws, q);
}

// [0, 4q) -> [0, q)
__device__ __forceinline__ u64 canon4(u64 x, u64 q, u64 q2) { return csub(csub(x, q2), q); }

// a*b mod q for two VARIABLE operands: classical Barrett (HAC 14.42) on the 128-bit product.
// n = bit length of q (2 <= n <= 62), mu = floor(2^(2n) / q) < 2^(n+1).
struct Barrett {
    u64 q;
    u64 mu;
    int sh1;  // n - 1
    int sh2;  // n + 1
};

// host: HAC 14.42 constants for a modulus of 2 .. 62 bits
inline Barrett make_barrett(u64 q) {
    const int nbits = 64 - __builtin_clzll(q);
    Barrett B;
    B.q = q;
    B.mu = (u64)((((unsigned __int128)1) << (2 * nbits)) / q);
    B.sh1 = nbits - 1;
    B.sh2 = nbits + 1;
    return B;
}

// a, b in [0, q)  ->  a*b mod q in [0, q)
__device__ __forceinline__ u64 mulmod_barrett(u64 a, u64 b, const Barrett &B) {
    u64 lo = a * b, hi = __umul64hi(a, b);
    u64 A = (hi << (64 - B.sh1)) | (lo >> B.sh1);  // floor(P / 2^(n-1)) < 2^(n+1)
    u64 plo = A * B.mu, phi = __umul64hi(A, B.mu);
    u64 qhat = (phi << (64 - B.sh2)) | (plo >> B.sh2);  // true quotient - {0,1,2}
    u64 r = lo - qhat * B.q;                            // in [0, 3q)
    return csub(csub(r, B.q), B.q);
}

}  // namespace fhe
