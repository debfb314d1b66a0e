// 30-bit arithmetic policy for the EXACT torus products of row T (torus_kernels.hpp).
//
// The exact negacyclic product of a digit polynomial with a key polynomial needs |coefficient| < P/2 for the CRT modulus P.  Two
// 60-bit primes give 2^119; the BASELINE config 5 shape needs only ~2^82, which THREE primes below 2^30 cover (2^89) -- and a
// 30-bit modular butterfly maps onto the hardware's 32-bit multipliers directly: Shoup product w y - floor(ws y / 2^32) p in 4
// instructions, Harvey lazy values in [0, 4p) < 2^32, conditional subtraction as one v_min_u32.  ~9 instructions per butterfly
// against ~17 for the 60-bit pseudo-Mersenne policy, half the registers per coefficient.  Keys whose bound exceeds 2^88 (the
// reference's own N = 2048, base 2^23 set: 2^97) stay on the two-prime 60-bit path.
#pragma once
#include "arith.hpp"

namespace fhe {

struct Mod30Desc {              // one prime p < 2^30, p = 1 (mod 2^12); tables in HBM
    unsigned p;
    const uint2 *tw, *twi;      // {w, floor(w 2^32 / p)}: psi^bitrev(j), psi^-bitrev(j), j < 2^11
    unsigned ninv[12], ninv_s[12];  // (2^k)^-1 mod p and its companion
    unsigned pinv_neg;          // -p^-1 mod 2^32 (Montgomery)
    unsigned r2;                // 2^64 mod p: x -> x 2^32 mod p via one Montgomery product
    unsigned r1;                // 2^32 mod p: folds a 64-bit sum of products to below 2^32 p before its one Montgomery reduction
};

__device__ __forceinline__ unsigned csub32(unsigned x, unsigned m) { return min(x, x - m); }  // x < 2m: x mod~ m (x - m wraps huge if x < m)

struct Arith30 {
    typedef unsigned Elem;
    typedef uint2 TwRaw;
    typedef uint2 TwReg;
    static constexpr int PREFETCH = 0;
    static constexpr bool GS_FOLDS = false;
    static constexpr int CT_LAYERS = 64;  // Harvey butterflies renormalise every layer
    static constexpr bool PASS_FOLD = false;
    static constexpr int GS_SPAN = 0;
    static __device__ constexpr bool ct_fold_at(int) { return false; }
    struct K {
        unsigned p, p2;
        const FHE_CONST uint2 *tw, *twi;  // constant address space: wave-uniform entries become scalar loads
        unsigned ninv, ninv_s, pinv_neg;
        int pb, prefix;
        unsigned r1;
    };
    static __device__ __forceinline__ K make(const Mod30Desc &D, int log_n) {
        return K{D.p, 2 * D.p, (const FHE_CONST uint2 *)D.tw, (const FHE_CONST uint2 *)D.twi, D.ninv[log_n], D.ninv_s[log_n], D.pinv_neg, 0, 0, D.r1};
    }
    template <bool INV>
    static __device__ __forceinline__ TwRaw fetch(const K &k, int idx) {
        const FHE_CONST uint2 *p = (INV ? k.twi : k.tw) + idx;
        TwRaw r;
        r.x = p->x; r.y = p->y;
        return r;
    }
    static __device__ __forceinline__ TwReg prep(const TwRaw &r) { return r; }
    // w y mod~ p in [0, 2p) for any 32-bit y (w < p, ws = floor(w 2^32 / p))
    static __device__ __forceinline__ unsigned mulred(unsigned y, unsigned w, unsigned ws, unsigned p) { return w * y - __umulhi(ws, y) * p; }
    // Cooley-Tukey, values in [0, 4p)
    static __device__ __forceinline__ void ct(unsigned &X, unsigned &Y, const TwReg &w, const K &k) {
        const unsigned x = csub32(X, k.p2), t = mulred(Y, w.x, w.y, k.p);
        X = x + t;
        Y = x - t + k.p2;
    }
    // Gentleman-Sande, values in [0, 2p)
    template <int PH>
    static __device__ __forceinline__ void gs(unsigned &X, unsigned &Y, const TwReg &w, const K &k) {
        const unsigned s = csub32(X + Y, k.p2), d = X - Y + k.p2;
        X = s;
        Y = mulred(d, w.x, w.y, k.p);
    }
    static __device__ __forceinline__ unsigned gs_fold(unsigned x, const K &) { return x; }
    static __device__ __forceinline__ unsigned fold(unsigned x, const K &) { return x; }
    static __device__ __forceinline__ unsigned canon_fwd(unsigned x, const K &k) { return csub32(csub32(x, k.p2), k.p); }
    static __device__ __forceinline__ unsigned finish_inv(unsigned x, const K &k) { return csub32(mulred(x, k.ninv, k.ninv_s, k.p), k.p); }
    // multiply-accumulate against key values stored in Montgomery form (k 2^32 mod p): REDC(x k~) = x k mod~ p in [0, 2p);
    // the sum is kept in [0, 2p), a valid inverse-transform input.  x: lazy forward output in [0, 4p)
    typedef unsigned MacAcc;
    static __device__ __forceinline__ MacAcc mac_zero() { return 0; }
    static __device__ __forceinline__ unsigned redc(u64 t, const K &k) {
        const unsigned m = (unsigned)t * k.pinv_neg;
        return (unsigned)((t + (u64)m * k.p) >> 32);  // t < 2^32 p: result < 2p
    }
    static __device__ __forceinline__ MacAcc mac(MacAcc acc, unsigned xin, unsigned kmont, const K &k) {
        return csub32(acc + redc((u64)xin * kmont, k), k.p2);
    }
    // The same sum UNREDUCED: every term is one v_mad_u64_u32 into a 64-bit accumulator (x in [0, 2p) < 2^31, key < p < 2^30: a
    // product is below 2 p^2 < 2^61, eight terms stay below 2^64), and ONE reduction per output closes it: S = hi 2^32 + lo =
    // hi r1 + lo (mod p) < 2^62 + 2^32, whose Montgomery reduction is below 2^30 + 1 + p -- one conditional subtraction of 2p
    // leaves a valid inverse-transform input in [0, 2p).  6 instructions per term become 1, plus 5 per output.
    static __device__ __forceinline__ unsigned mac_close(u64 s, const K &k) {
        const u64 t = (u64)(unsigned)(s >> 32) * k.r1 + (unsigned)s;
        return csub32(redc(t, k), k.p2);
    }
};

}  // namespace fhe
