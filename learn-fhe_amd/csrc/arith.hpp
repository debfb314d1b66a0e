// Arithmetic policies shared by every transform kernel, and the in-register radix-2^R butterfly networks built on them.
//
//   ArithShoup   any prime q < 2^62: Shoup multiplication by the fixed twiddle {w, floor(w 2^64 / q)} + Harvey lazy
//                reduction (forward values in [0, 4q), inverse in [0, 2q)).
//   ArithPM<B>   pseudo-Mersenne primes q = 2^B - c (c <= 2^(B-33), 34 <= B <= 60): every prime
//                `two_adic_primes(bits, log_n)` (util/src/zq.rs:325-329) yields for the BASELINE configs.
//
// Both leave canonical values in [0, q) when a kernel finishes: results are bit-identical to the reference's
// `(a * b) as u128 % q` arithmetic (util/src/zq.rs:156-196) whichever policy ran.
#pragma once
#include <type_traits>

#include "dev_arith.hpp"

namespace fhe {

// Pointers that kernels read out of a descriptor in memory carry no address space: loads through them compile to FLAT
// instructions, which count against BOTH vmcnt and lgkmcnt and so serialise against every LDS wait.  Twiddle tables live in
// HBM: say so.
#define FHE_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const FHE_GLOBAL T *as_global(const T *p) {
    return (const FHE_GLOBAL T *)p;
}
// Twiddle tables are never written while a kernel runs: read through the CONSTANT address space a wave-uniform index becomes a
// scalar load (SGPRs, lgkmcnt) even when the kernel has stored to global memory before -- a persistent workgroup's second
// transform, where plain global loads can no longer be proven unclobbered and fall back to 64 lanes loading one address.
#define FHE_CONST __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const FHE_CONST T *as_const(const T *p) {
    return (const FHE_CONST T *)p;
}

// non-template kernels defined in headers that several translation units include (each unit uses a subset)
#define FHE_HEADER_KERNEL static __attribute__((unused)) __global__

template <int I, int END, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < END) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, END>(f);
    }
}

// per-modulus constants as the kernels read them from HBM (one entry per RNS limb)
struct ModDesc {
    u64 q;
    const TwPair *tw, *twi;
    u64 ninv[20], ninv_s[20];  // (2^k)^-1 mod q and its Shoup companion, k = log2 N
    u64 ninv_w[20], ninv_w_s[20];  // (2^k)^-1 * twi[1] mod q (+ companion): the last inverse layer's twiddle with n^-1 folded in
    u64 one_s;                 // floor(2^64 / q): Shoup companion of 1
    // pseudo-Mersenne fast path: q = 2^pm_b - pm_c, twiddles in the packed 8-byte PmRaw form; pm_b = 0: not eligible
    const u64 *tww, *twwi;
    unsigned pm_c;
    int pm_b;
    // two-operand split form of the same twiddles (ArithDS): 16-byte entries {a0, a1, b0, b1}, w = a0 + a1 2^(pm_b-31),
    // w 2^32 mod q = b0 + b1 2^(pm_b-31); ds_pow = 2^(pm_b-31), read from memory so that the compiler keeps the multiply-add
    const uint4 *twd, *twdi;
    unsigned ds_pow;
    // ... of (2^k)^-1, of (2^k)^-1 twi[1] and of 1: split on the host -- the w 2^32 mod q half is a 128-bit remainder, which a
    // kernel that formed it itself paid for in every workgroup (0.045 ms of a 0.33 ms launch of 4096 inverse transforms)
    uint4 ds_ninv[20], ds_ninv_w[20], ds_one;
    // Diagonal form of the inverse's first three layers (ntt14w.hpp, pass 3): within a block of 8 coefficients the three
    // Gentleman-Sande layers equal a fixed 8-point network (twiddles twi[1], twi[2], twi[3] only) followed by the diagonal
    // (1, t, t^2, .., t^7), t = twi[4 blk] the block's first last-layer twiddle.  twd3i[(p - 1) * twd3_stride + blk] = t^p in the
    // two-operand form, p = 1..7, blk < twd3_stride = cap / 4: valid for every ring size and sub-transform prefix like twi itself.
    const uint4 *twd3i;
    unsigned twd3_stride;
    // classical Barrett constants (dev_arith.hpp) for products of two VARIABLE operands inside a transform (fused ring products)
    u64 bar_mu;
    int bar_sh1, bar_sh2;
};

// Optional operands of a transform launch; all zero = the plain in-place transform.  Indices count sub-polynomials of the
// launch (the 2^14 blocks of a larger ring count one each).
//   src: sub-polynomial s is READ from src + (s % src_mod) * len instead of its own place (out-of-place forward; an inverse that
//        shares its evaluation-domain input between several outputs);
//   mul: (inverse only) every loaded evaluation is multiplied by the one at the same position of mul + mi * len,
//        mi = (s / mul_div) * mul_period + s % mul_period: util/src/ring/fft/zq.rs:17 (`a[i] *= b[i]`) and ring/rns.rs:148-158
//        fused into the load of the inverse transform.
struct NttIo {
    const u64 *src = nullptr;
    unsigned src_mod = 1;
    const u64 *mul = nullptr;
    unsigned mul_div = 1, mul_period = 1;
    // (forward only) up to four separate sources in ONE launch: sub-polynomial s reads from source s / src_group (src, src2, src3,
    // src4), sub-polynomial s % src_group of it.  Four caller buffers of 128 polynomials each would otherwise be four launches of
    // 128 workgroups -- half a chip each for the one-workgroup-per-CU kernels.
    const u64 *src2 = nullptr, *src3 = nullptr, *src4 = nullptr;
    unsigned src_group = 0;
    // (multiplying inverse of the wave-local kernels only) split output: with dst_period = P and dst_first2 = F, polynomial s = g P + l
    // is STORED at data + (g F + l) polynomials if l < F, at dst2 + (g (P - F) + l - F) polynomials otherwise -- a limb-sharded key
    // switch gets its q-limb products and its p-limb products (the ones the all-gather sends) as two contiguous blocks
    u64 *dst2 = nullptr;
    unsigned dst_period = 0, dst_first2 = 0;
};
// where sub-polynomial `sub` (2^log_len words) is read from; `own` = its place in the launch's buffer
__device__ __forceinline__ const u64 *ntt_src(const NttIo &io, unsigned sub, int log_len, const u64 *own) {
    if (!io.src) return own;
    if (io.src_group) {
        const unsigned gsel = sub / io.src_group, r = sub - gsel * io.src_group;
        const u64 *b = gsel == 0 ? io.src : gsel == 1 ? io.src2 : gsel == 2 ? io.src3 : io.src4;
        return b + (size_t(r) << log_len);
    }
    return io.src + (size_t(sub % io.src_mod) << log_len);
}

struct ArithShoup {
    typedef u64 Elem;  // the integer type a coefficient occupies in registers, LDS and HBM
    struct K {
        u64 q, q2;
        const FHE_CONST TwPair *tw, *twi;  // constant address space: wave-uniform entries become scalar loads (as in the other policies)
        u64 ninv, ninv_s;
        int pb, prefix;
        TwPair ninv_w;  // n^-1 * twi[1] (whole rings only, pb = 0)
        Barrett bar;
    };
    static __device__ __forceinline__ K make(const ModDesc &D, int log_n_total, int pb, int prefix) {
        return K{D.q, 2 * D.q, as_const(D.tw), as_const(D.twi), pb ? 1 : D.ninv[log_n_total], pb ? D.one_s : D.ninv_s[log_n_total], pb, prefix,
                 TwPair{D.ninv_w[log_n_total], D.ninv_w_s[log_n_total]}, Barrett{D.q, D.bar_mu, D.bar_sh1, D.bar_sh2}};
    }
    // x y mod q for two canonical variable operands (fused pointwise products): canonical
    static __device__ __forceinline__ u64 mulvar(u64 x, u64 y, const K &k) { return mulmod_barrett(x, y, k.bar); }
    // The LAST inverse layer of a whole ring with n^-1 folded in (util/src/ring/fft.rs:59-77: layer 0, then `* n_inv`): the
    // difference branch takes twi[1] n^-1 as its twiddle, only the sum branch still needs a product.  Canonical outputs.
    template <int PH = 1>
    static __device__ __forceinline__ void gs_last_scaled(u64 &X, u64 &Y, const K &k) {
        const u64 s = X + Y, d = X - Y + k.q2;
        X = csub(mul_shoup_lazy(s, k.ninv, k.ninv_s, k.q), k.q);
        Y = csub(mul_shoup_lazy(d, k.ninv_w.w, k.ninv_w.ws, k.q), k.q);
    }
    // ... and of a sub-transform (pb > 0: no scaling here): an ordinary butterfly, canonical outputs
    template <int PH = 1>
    static __device__ __forceinline__ void gs_last_plain(u64 &X, u64 &Y, const TwPair &p, const K &k) {
        gs_bfly(X, Y, p.w, p.ws, k.q, k.q2);
        X = csub(X, k.q); Y = csub(Y, k.q);
    }
    // twiddles: TwRaw is what a load brings from HBM/L2, TwReg what a butterfly consumes (prep() converts, once per twiddle)
    typedef TwPair TwRaw;
    typedef TwPair TwReg;
    static constexpr int PREFETCH = 0;  // largest twiddle set fetched one unit ahead (ntt14.hpp); 16-byte raw twiddles: none fits
    template <bool INV>
    static __device__ __forceinline__ TwRaw fetch(const K &k, int idx) {
        const FHE_CONST TwPair *p = (INV ? k.twi : k.tw) + idx;
        TwRaw r;
        r.w = p->w; r.ws = p->ws;
        return r;
    }
    static __device__ __forceinline__ TwReg prep(const TwRaw &r) { return r; }
    static __device__ __forceinline__ void ct(u64 &X, u64 &Y, const TwReg &p, const K &k) { ct_bfly(X, Y, p.w, p.ws, k.q, k.q2); }
    template <int PH>
    static __device__ __forceinline__ void gs(u64 &X, u64 &Y, const TwReg &p, const K &k) { gs_bfly(X, Y, p.w, p.ws, k.q, k.q2); }
    static constexpr bool GS_FOLDS = false;
    static constexpr int GS_SPAN = 0;
    static constexpr int CT_LAYERS = 64;  // Harvey butterflies renormalise every layer
    static constexpr bool PASS_FOLD = false;
    static __device__ constexpr bool ct_fold_at(int) { return false; }
    static __device__ __forceinline__ u64 gs_fold(u64 x, const K &) { return x; }
    static __device__ __forceinline__ u64 fold(u64 x, const K &) { return x; }
    static __device__ __forceinline__ u64 canon_fwd(u64 x, const K &k) { return canon4(x, k.q, k.q2); }
    static __device__ __forceinline__ u64 finish_inv(u64 x, const K &k) { return csub(mul_shoup_lazy(x, k.ninv, k.ninv_s, k.q), k.q); }
    // evaluation-domain multiply-accumulate against a VARIABLE operand (key rows): x = forward-transform output (lazy)
    typedef u64 MacAcc;
    static __device__ __forceinline__ MacAcc mac_zero() { return 0; }
    static __device__ __forceinline__ u64 mac_in(u64 x, const K &k) { return canon4(x, k.q, k.q2); }
    static __device__ __forceinline__ u64 mac(u64 acc, u64 xin, u64 kval, int, const K &k, const Barrett &B) {
        return csub(acc + mulmod_barrett(xin, kval, B), k.q);
    }
    static __device__ __forceinline__ u64 mac_finish(u64 acc, const K &) { return acc; }  // canonical: valid inverse input
};

// Pseudo-Mersenne product, q = 2^B - c (c < 2^(B-33), 34 <= B <= 60).
// On gfx950 every integer VALU instruction costs about the same issue time (~16 lanes/clk/SIMD; v_mul_lo/hi_u32 twice
// that), v_mad_u64_u32 included, so the product is arranged to need as FEW instructions as possible, all word aligned:
// the fixed operand w is split at B-31 bits, w = wl + wh 2^(B-31) (wh < 2^31), and kept with wlp = wl << (63-B), so that
// with y = y0 + y1 2^32 (any y < 2^63)
//     w y = z0 + z1 2^(B-31) + wh y1 2^(B+1),  z0 = wl y0,  z1 = wh y0 + wlp y1 < 2^64 (no carry),
//     z1 2^(B-31) = lo32(z1) 2^(B-31) + hi32(z1) 2^(B+1),            2^(B+1) = 2c (mod q),
//     w y = v + u 2c (mod q),   v = z0 + lo32(z1) 2^(B-31) < 2^(B+2),   u = wh y1 + hi32(z1) < 2^62 + 2^32,
// followed by one fold of the 96-bit v + u 2c at bit B.  8 multiply-adds + 4 other instructions (two of them only move a
// high word into a zero-extended pair), no mask, no compare/select, no companion table.
// Result < 2^B + 2^(64 + 2k - B) <= 1.25 * 2^B   (k = bits of c <= B - 33).
struct PmTw {  // one twiddle as a butterfly consumes it (registers only)
    unsigned wl, wlp, wh;
};
// In memory a twiddle is 8 bytes {wl, wh}; wlp is one shift, done once per loaded twiddle (a twiddle serves 2..16
// butterflies of a thread).
typedef uint2 PmRaw;
struct PmK {
    u64 q, q2, q4;
    unsigned c, c2;  // c and 2c
};

// a * b + c as ONE v_mad_u64_u32 even where the compiler would strength-reduce a power-of-two multiplier into a 64-bit
// shift plus a 64-bit add (two instructions of the same issue cost each)
__device__ __forceinline__ u64 mad_u64(unsigned a, unsigned b, u64 c) {
    u64 d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
    return d;
}

// x >> 32 as a 64-bit operand in ONE instruction (the compiler builds the zero-extended pair from two moves)
__device__ __forceinline__ u64 hi_word(u64 x) {
    u64 d;
    asm("v_lshrrev_b64 %0, 32, %1" : "=v"(d) : "v"(x));
    return d;
}

// w y = v + u 2c (mod q), unreduced: v < 2^(B+2), u < wh y1 + 2^32
template <int B>
__device__ __forceinline__ void pm_mul_vu(u64 y, const PmTw w, u64 &v, u64 &u) {
    const unsigned y0 = (unsigned)y, y1 = (unsigned)(y >> 32);
    const u64 z0 = (u64)w.wl * y0;
    const u64 z1 = (u64)w.wh * y0 + (u64)w.wlp * y1;
    v = mad_u64((unsigned)z1, 1u << (B - 31), z0);
    u = (u64)w.wh * y1 + hi_word(z1);
}

// v + u 2c folded once at bit B (v + u 2c < 2^96, u < 2^63): < 2^B + 2^(bits(v + u 2c) - B + bits(c))
template <int B>
__device__ __forceinline__ u64 pm_reduce_vu(u64 v, u64 u, const PmK &k) {
    constexpr unsigned HMASK = (1u << (B - 32)) - 1;
    const u64 r1 = (u64)(unsigned)u * k.c2 + v;
    const u64 r2 = (u64)(unsigned)(u >> 32) * k.c2 + (r1 >> 32);  // v + u 2c = r2 * 2^32 + lo32(r1)
    const unsigned h2 = (unsigned)(r2 >> (B - 32));
    const u64 l2 = ((u64)((unsigned)r2 & HMASK) << 32) | (unsigned)r1;
    return (u64)h2 * k.c + l2;
}

template <int B>
__device__ __forceinline__ u64 pm_mul(u64 y, const PmTw w, const PmK &k) {
    u64 v, u;
    pm_mul_vu<B>(y, w, v, u);
    return pm_reduce_vu<B>(v, u, k);
}

// B = bit length of q (compile time: every shift and mask is an immediate)
template <int B>
struct ArithPM {
    typedef u64 Elem;
    static constexpr u64 MASK = (u64(1) << B) - 1;
    struct K {
        PmK m;
        const FHE_CONST PmRaw *tw, *twi;
        PmTw ninv;    // n^-1 (or 1) in twiddle form
        PmTw ninv_w;  // n^-1 * twi[1] (whole rings only, pb = 0)
        int pb, prefix;
    };
    typedef PmRaw TwRaw;
    typedef PmTw TwReg;
    // ntt14.hpp fetches twiddle sets of up to PREFETCH entries one unit ahead of their butterflies.  Measured (4096
    // transforms, tools/ntt_lab.hip): none 0.436 / 0.444 ms (fwd / inv), <= 2: 0.430 / 0.445, <= 4: 0.412 / 0.435, all (8):
    // 0.433 / 0.444 -- the 8-entry sets push the kernel into 12 spilled VGPRs whose scratch traffic eats the hidden latency.
    static constexpr int PREFETCH = 4;
    // memory form of twiddle w: low word wl = w mod 2^(B-31), high word wh = w >> (B-31)
    static __host__ __device__ __forceinline__ u64 pack(u64 w) { return ((w >> (B - 31)) << 32) | (w & ((u64(1) << (B - 31)) - 1)); }
    template <bool INV>
    static __device__ __forceinline__ TwRaw fetch(const K &k, int idx) {
        const FHE_CONST PmRaw *p = (INV ? k.twi : k.tw) + idx;
        TwRaw r;
        r.x = p->x; r.y = p->y;
        return r;
    }
    static __device__ __forceinline__ TwReg prep(const TwRaw &r) { return PmTw{r.x, r.x << (63 - B), r.y}; }
    static __host__ __device__ __forceinline__ PmTw split(u64 w) {
        PmTw t;
        t.wl = (unsigned)(w & ((u64(1) << (B - 31)) - 1));
        t.wlp = t.wl << (63 - B);
        t.wh = (unsigned)(w >> (B - 31));
        return t;
    }
    static __device__ __forceinline__ K make(const ModDesc &D, int log_n_total, int pb, int prefix) {
        K k;
        k.m.q = D.q; k.m.q2 = 2 * D.q; k.m.q4 = 4 * D.q;
        k.m.c = D.pm_c; k.m.c2 = 2 * D.pm_c;
        k.tw = (const FHE_CONST PmRaw *)D.tww; k.twi = (const FHE_CONST PmRaw *)D.twwi;
        k.ninv = split(pb ? 1 : D.ninv[log_n_total]);
        k.ninv_w = split(D.ninv_w[log_n_total]);
        k.pb = pb; k.prefix = prefix;
        return k;
    }
    // x mod~ q: < 2^B + 2^(64-B) c
    static __device__ __forceinline__ u64 fold1(u64 x, const PmK &m) {
        const unsigned hi = (unsigned)(x >> 32), h = hi >> (B - 32);
        const u64 base = ((u64)(hi & (unsigned)(MASK >> 32)) << 32) | (unsigned)x;
        return (u64)h * m.c + base;
    }
    // x y mod~ q for two canonical variable operands (fused pointwise products): y is split like a twiddle; < q + eps
    static __device__ __forceinline__ u64 mulvar(u64 x, u64 y, const K &k) { return pm_mul<B>(x, split(y), k.m); }
    // Forward butterfly without any reduction.  Values grow by at most 2q per layer; a multiplicand must stay below 2^63
    // and a sum below 2^64, which holds for 4 layers after a fold (inputs < q + eps -> multiplicands < 7q, outputs < 9q).
    static __device__ __forceinline__ void ct(u64 &X, u64 &Y, const TwReg &w, const K &k) {
        const u64 t = pm_mul<B>(Y, w, k.m);
        const u64 x = X;
        X = x + t;
        Y = x - t + k.m.q2;
    }
    // Inverse butterflies come in pairs of layers: PH = 0 takes inputs < q + eps (outputs: sum < 2q + , product < q +),
    // PH = 1 takes those (outputs: sum < 4q +, product < q +); the sums of a PH = 1 layer are folded (gs_fold) before the
    // next pair.
    template <int PH>
    static __device__ __forceinline__ void gs(u64 &X, u64 &Y, const TwReg &w, const K &k) {
        const u64 s = X + Y;
        const u64 d = X - Y + (PH ? k.m.q4 : k.m.q2);
        X = s;
        Y = pm_mul<B>(d, w, k.m);
    }
    static constexpr bool GS_FOLDS = true;
    static constexpr int GS_SPAN = 0;  // pairs of layers inside a network
    // last inverse layer, n^-1 folded in (see ArithShoup::gs_last_scaled); PH = 1 inputs (sums < 2q +, products < q +)
    template <int PH = 1>
    static __device__ __forceinline__ void gs_last_scaled(u64 &X, u64 &Y, const K &k) {
        const u64 s = X + Y, d = X - Y + k.m.q4;
        X = csub(pm_mul<B>(s, k.ninv, k.m), k.m.q);
        Y = csub(pm_mul<B>(d, k.ninv_w, k.m), k.m.q);
    }
    template <int PH = 1>
    static __device__ __forceinline__ void gs_last_plain(u64 &X, u64 &Y, const PmTw &w, const K &k) {
        const u64 s = X + Y, d = X - Y + k.m.q4;
        X = csub(fold1(s, k.m), k.m.q);
        Y = csub(pm_mul<B>(d, w, k.m), k.m.q);
    }
    // forward layers between two folds: inputs < q + eps, a multiplicand of layer L is < (2L - 1) q and must be < 2^63
    static constexpr int CT_LAYERS = ((1 << (63 - B)) + 1) / 2 > 64 ? 64 : ((1 << (63 - B)) + 1) / 2;
    static constexpr bool PASS_FOLD = true;  // ... and at every pass boundary of a transform longer than that
    static __device__ constexpr bool ct_fold_at(int) { return false; }
    static __device__ __forceinline__ u64 gs_fold(u64 x, const K &k) { return fold1(x, k.m); }
    static __device__ __forceinline__ u64 fold(u64 x, const K &k) { return fold1(x, k.m); }  // between forward passes
    static __device__ __forceinline__ u64 canon_fwd(u64 x, const K &k) { return csub(fold1(x, k.m), k.m.q); }
    static __device__ __forceinline__ u64 finish_inv(u64 x, const K &k) { return csub(pm_mul<B>(x, k.ninv, k.m), k.m.q); }
    // Multiply-accumulate against key values (stored pre-split: one shift completes the operand); the lazy forward output
    // (< 2^63) is the multiplicand as it stands.  The products stay UNREDUCED: each term adds its (v, u) pair (w y = v + u 2c)
    // to a pair of 64-bit sums and ONE reduction per MAC_TERMS terms (and one in mac_finish) folds them: 6 instructions + 2
    // adds per term instead of 12 + 1.  Bounds: v < 2^(B+2); u < 2^31 y1 + 2^32 with y1 = multiplicand >> 32.
    //  * B <= 56: multiplicands are forward outputs that may never have been folded (ntt_kernels.hpp: fwd_run), < (2 log2 N + 1) q
    //    <= 27 q < 2^(B+5), so u < 2^(B+4) + 2^32: T terms keep sum u < T 2^(B+4) + T 2^32 < 2^64 for T = 32 at B <= 54 (2^63 + 2^37),
    //    T = 16 at B = 55, T = 8 at B = 56; sum v < 2^(B+7).
    //  * B > 56: multiplicands < 2^63, u < 2^62 + 2^32: 3 terms keep sum u < 2^64 and sum v < 2^64 - 2^50.
    // In every case v + u 2c < 2^(65 + bits(c)) folds to < 2^B + 2^(2 bits(c) + 6) = q + eps.
    static constexpr int MAC_TERMS = B <= 54 ? 32 : B == 55 ? 16 : B == 56 ? 8 : 3;
    struct MacAcc { u64 v, u; };
    static __device__ __forceinline__ MacAcc mac_zero() { return MacAcc{0, 0}; }
    static __device__ __forceinline__ u64 mac_in(u64 x, const K &) { return x; }
    static __device__ __forceinline__ MacAcc mac(MacAcc acc, u64 xin, u64 kval, int term, const K &k, const Barrett &) {
        const unsigned wl = (unsigned)kval;
        const PmTw w{wl, wl << (63 - B), (unsigned)(kval >> 32)};
        u64 v, u;
        pm_mul_vu<B>(xin, w, v, u);
        acc.v += v; acc.u += u;
        if ((term % MAC_TERMS) == MAC_TERMS - 1) acc = MacAcc{pm_reduce_vu<B>(acc.v, acc.u, k.m), 0};
        return acc;
    }
    static __device__ __forceinline__ u64 mac_finish(MacAcc acc, const K &k) { return pm_reduce_vu<B>(acc.v, acc.u, k.m); }  // < q + eps
};

// Pseudo-Mersenne product on a twiddle kept as TWO fixed operands, q = 2^B - c (c < 2^(B-33), 34 <= B <= 60).
// The multiplicand is cut at its register boundary, y = y0 + y1 2^32, and the twiddle is stored for both halves:
//     w0 = w = a0 + a1 2^(B-31),   w1 = w 2^32 mod q = b0 + b1 2^(B-31)      (a0, b0 < 2^(B-31);  a1, b1 < 2^31)
//     w y = w0 y0 + w1 y1 = S0 + S1 2^(B-31)  (mod q),   S0 = a0 y0 + b0 y1 < 2^(B+2),   S1 = a1 y0 + b1 y1 < 2^64
//     S1 2^(B-31) = lo32(S1) 2^(B-31) + hi32(S1) 2^(B+1),   2^(B+1) = 2c  (mod q)
//     R = S0 + lo32(S1) 2^(B-31) + hi32(S1) 2c  <  2^(B+2) + 2^(B+1) + 2^B  <  2^(B+3)               for ANY 64-bit y
// SIX v_mad_u64_u32 and nothing else (lo32 / hi32 of a register pair are its registers); one fold at bit B (mask the high
// word, shift it, one more multiply-add) brings R below 2^B + 8c.  A butterfly is 13 instructions where the single-operand
// form (ArithPM) needs 18: the price is a 16-byte twiddle.
struct DsK {
    u64 q, q2, q4;
    unsigned c, c2, pw;  // c, 2c, 2^(B-31)
};

template <int B>
struct ArithDS {
    typedef u64 Elem;
    static constexpr u64 MASK = (u64(1) << B) - 1;
    static constexpr int SPLIT = B - 31;
    typedef uint4 TwRaw;  // {a0, a1, b0, b1}: the memory form IS the register form
    typedef uint4 TwReg;
    struct K {
        DsK m;
        const FHE_CONST uint4 *tw, *twi;
        uint4 ninv, ninv_w;
        int pb, prefix;
        const FHE_CONST uint4 *tw3i;  // ModDesc::twd3i
        unsigned tw3_stride;
    };
    static constexpr int PREFETCH = 4;
    static __host__ __device__ __forceinline__ uint4 split(u64 w, u64 q) {
        const u64 w1 = (u64)((((unsigned __int128)w) << 32) % q);
        const u64 lo = (u64(1) << SPLIT) - 1;
        return uint4{(unsigned)(w & lo), (unsigned)(w >> SPLIT), (unsigned)(w1 & lo), (unsigned)(w1 >> SPLIT)};
    }
    template <bool INV>
    static __device__ __forceinline__ TwRaw fetch(const K &k, int idx) {
        const FHE_CONST uint4 *p = (INV ? k.twi : k.tw) + idx;
        TwRaw r;
        r.x = p->x; r.y = p->y; r.z = p->z; r.w = p->w;
        return r;
    }
    static __device__ __forceinline__ TwReg prep(const TwRaw &r) { return r; }
    static __device__ __forceinline__ K make(const ModDesc &D, int log_n_total, int pb, int prefix) {
        K k;
        k.m.q = D.q; k.m.q2 = 2 * D.q; k.m.q4 = 4 * D.q;
        k.m.c = D.pm_c; k.m.c2 = 2 * D.pm_c; k.m.pw = D.ds_pow;
        k.tw = as_const(D.twd); k.twi = as_const(D.twdi);
        k.ninv = pb ? D.ds_one : D.ds_ninv[log_n_total];
        k.ninv_w = D.ds_ninv_w[log_n_total];
        k.pb = pb; k.prefix = prefix;
        k.tw3i = as_const(D.twd3i); k.tw3_stride = D.twd3_stride;
        return k;
    }
    // w y mod q, unreduced: < 2^(B+3) for any y
    static __device__ __forceinline__ u64 mul_raw(u64 y, const uint4 &w, const DsK &m) {
        const unsigned y0 = (unsigned)y, y1 = (unsigned)(y >> 32);
        const u64 s1 = (u64)w.y * y0 + (u64)w.w * y1;
        const u64 s0 = (u64)w.x * y0 + (u64)w.z * y1;
        const u64 v = (u64)(unsigned)s1 * m.pw + s0;
        return (u64)(unsigned)(s1 >> 32) * m.c2 + v;
    }
    // x mod~ q: < 2^B + (x >> B) c
    // (written on the two words: as `(x & MASK) + (x >> B) * c` hipcc builds the masked value in a fresh register pair and pays a
    // v_mov per fold -- 224 of them in a 2^14 transform)
    static __device__ __forceinline__ u64 fold1(u64 x, const DsK &m) {
        const unsigned hi = (unsigned)(x >> 32), h = hi >> (B - 32);
        const u64 base = ((u64)(hi & (unsigned)(MASK >> 32)) << 32) | (unsigned)x;
        return (u64)h * m.c + base;
    }
    static __device__ __forceinline__ u64 mul(u64 y, const uint4 &w, const DsK &m) { return fold1(mul_raw(y, w, m), m); }  // < q + 9c
    // two VARIABLE canonical operands: the second one has no precomputed w 2^32 mod q, so this is the one-operand product
    static __device__ __forceinline__ u64 mulvar(u64 x, u64 y, const K &k) {
        return pm_mul<B>(x, ArithPM<B>::split(y), PmK{k.m.q, k.m.q2, k.m.q4, k.m.c, k.m.c2});
    }
    // Forward butterfly: the multiplicand may be ANY 64-bit value, the product comes back below q + 9c, so a value grows by at
    // most 2q per layer and only the sums must stay below 2^64: (2^(64-B) - 2) / 2 layers between two folds (7 at 60 bits).
    static __device__ __forceinline__ void ct(u64 &X, u64 &Y, const TwReg &w, const K &k) {
        const u64 t = mul(Y, w, k.m);
        const u64 x = X;
        X = x + t;
        Y = x + k.m.q2 - t;
    }
    static constexpr int CT_LAYERS = 64;                      // no fold tied to a network's own layer count ...
    static constexpr bool PASS_FOLD = false;                  // ... nor to pass boundaries:
    static constexpr int CT_SPAN = B >= 64 - 2 ? 1 : (((1 << (64 - B)) - 2) / 2 > 64 ? 64 : ((1 << (64 - B)) - 2) / 2);
    static __device__ constexpr bool ct_fold_at(int layer) { return layer > 0 && layer % CT_SPAN == 0; }  // before this layer of the transform
    // Inverse butterfly, PH = layers since the sums were last folded: inputs < 2^PH (q + eps), so the difference takes the offset
    // 2^(PH+1) q and the sum stays below 2^(PH+1) (q + eps): 64 - B - 1 layers between two folds (3 at 60 bits: 12 q + 8 q < 2^64).
    template <int PH>
    static __device__ __forceinline__ void gs(u64 &X, u64 &Y, const TwReg &w, const K &k) {
        static_assert(PH < 64 - B - 1, "sums overflow");
        const u64 s = X + Y;
        const u64 d = X + (k.m.q2 << PH) - Y;
        X = s;
        Y = mul(d, w, k.m);
    }
    static constexpr bool GS_FOLDS = true;
    static constexpr int GS_SPAN = 64 - B - 1 > 8 ? 8 : 64 - B - 1;  // kernels that pass their global step fold every GS_SPAN-th layer
    // last inverse layer (PH as in gs): a whole ring folds n^-1 into it, see ArithShoup::gs_last_scaled
    template <int PH = 1>
    static __device__ __forceinline__ void gs_last_scaled(u64 &X, u64 &Y, const K &k) {
        const u64 s = X + Y, d = X + (k.m.q2 << PH) - Y;
        X = csub(mul(s, k.ninv, k.m), k.m.q);
        Y = csub(mul(d, k.ninv_w, k.m), k.m.q);
    }
    template <int PH = 1>
    static __device__ __forceinline__ void gs_last_plain(u64 &X, u64 &Y, const TwReg &w, const K &k) {
        const u64 s = X + Y, d = X + (k.m.q2 << PH) - Y;
        X = csub(fold1(s, k.m), k.m.q);
        Y = csub(mul(d, w, k.m), k.m.q);
    }
    static __device__ __forceinline__ u64 gs_fold(u64 x, const K &k) { return fold1(x, k.m); }
    static __device__ __forceinline__ u64 fold(u64 x, const K &k) { return fold1(x, k.m); }
    static __device__ __forceinline__ u64 canon_fwd(u64 x, const K &k) { return csub(fold1(x, k.m), k.m.q); }
    static __device__ __forceinline__ u64 finish_inv(u64 x, const K &k) { return csub(mul(x, k.ninv, k.m), k.m.q); }
    // Multiply-accumulate against VARIABLE key values (the fused gadget products): a key value has no precomputed w 2^32 mod q,
    // so this is ArithPM's unreduced (v, u) accumulation on its 8-byte packed operands -- the transforms around it run on the
    // two-operand product.  Forward outputs here are < (2 log2 N + 1) q, inside the bound ArithPM::mac states.
    typedef typename ArithPM<B>::MacAcc MacAcc;
    static constexpr int MAC_TERMS = ArithPM<B>::MAC_TERMS;
    static __device__ __forceinline__ PmK pmk(const K &k) { return PmK{k.m.q, k.m.q2, k.m.q4, k.m.c, k.m.c2}; }
    static __device__ __forceinline__ MacAcc mac_zero() { return MacAcc{0, 0}; }
    static __device__ __forceinline__ u64 mac_in(u64 x, const K &) { return x; }
    static __device__ __forceinline__ MacAcc mac(MacAcc acc, u64 xin, u64 kval, int term, const K &k, const Barrett &) {
        const unsigned wl = (unsigned)kval;
        const PmTw w{wl, wl << (63 - B), (unsigned)(kval >> 32)};
        u64 v, u;
        pm_mul_vu<B>(xin, w, v, u);
        acc.v += v; acc.u += u;
        if ((term % MAC_TERMS) == MAC_TERMS - 1) acc = MacAcc{pm_reduce_vu<B>(acc.v, acc.u, pmk(k)), 0};
        return acc;
    }
    static __device__ __forceinline__ u64 mac_finish(MacAcc acc, const K &k) { return pm_reduce_vu<B>(acc.v, acc.u, pmk(k)); }
};

// ---------------------------------------------------------------------------------------------------------
// radix-2^R butterfly networks in registers, one reference layer (util/src/ring/fft.rs:40-77) at a time.
//
// A "unit" = one layer l of a radix-2^R network (layers L0 .. L0+R-1 of the sub-transform) applied to NREP replicas of the
// network that sit STRIDE registers apart.  SHARED: every replica has the same block prefix `top` (they differ in a bit
// BELOW the network), so 2^l twiddles serve all of them; otherwise replica rep has prefix top | rep << REP_SHIFT and its own
// 2^l twiddles.  tw_load() fetches a unit's twiddles, ct_apply()/gs_apply() run its butterflies: kernels that can afford the
// registers (A::PREFETCH) issue tw_load for the NEXT unit before ct_apply of the current one, so the L2 latency of a twiddle
// fetch hides behind a layer of butterflies instead of stalling every wave of the workgroup at every layer.
// ---------------------------------------------------------------------------------------------------------
#define FHE_SCHED_FENCE() asm volatile("" ::: "memory")

template <int L0_, int R_, int l_, int REP0_, int NREP_, int STRIDE_, bool SHARED_, int REP_SHIFT_ = 0>
struct Unit {
    static constexpr int L0 = L0_, R = R_, l = l_, REP0 = REP0_, NREP = NREP_, STRIDE = STRIDE_, REP_SHIFT = REP_SHIFT_;
    static constexpr bool SHARED = SHARED_;
    static constexpr int NT = (SHARED_ ? 1 : NREP_) << l_;  // twiddles of the unit
    static constexpr int half = 1 << (R_ - 1 - l_);
};

#ifdef FHE_LAB_ABLATE  // developer lab only (tools/ntt_lab2.hip): a policy may replace the twiddle loads of chosen layers by computed values
template <class A, class = void> struct tw_skip_mask { static constexpr unsigned value = 0; };
template <class A> struct tw_skip_mask<A, decltype((void)A::SKIP_MASK)> { static constexpr unsigned value = A::SKIP_MASK; };
#endif

template <class A, bool INV, class U, int NTW>
__device__ __forceinline__ void tw_load(typename A::TwRaw (&raw)[NTW], int top, const typename A::K &k) {
    static_assert(U::NT <= NTW, "twiddle buffer too small for this unit");
#pragma unroll
    for (int rr = 0; rr < (U::SHARED ? 1 : U::NREP); ++rr)
#pragma unroll
        for (int b = 0; b < (1 << U::l); ++b) {
            const int tp = U::SHARED ? top : (top | ((U::REP0 + rr) << U::REP_SHIFT));
            const int idx = (1 << (U::L0 + U::l + k.pb)) + ((((k.prefix << U::L0) | tp)) << U::l) + b;
#ifdef FHE_LAB_ABLATE
            if constexpr ((tw_skip_mask<A>::value >> (U::L0 + U::l)) & 1) { raw[(rr << U::l) | b] = A::fake(k, idx); continue; }
#endif
            raw[(rr << U::l) | b] = A::template fetch<INV>(k, idx);
        }
}

template <class A, class U, int NTW, int E>
__device__ __forceinline__ void ct_apply(typename A::Elem (&x)[E], const typename A::TwRaw (&raw)[NTW], const typename A::K &k) {
    // lazy policies bound how many layers may run between two folds (multiplicands must stay below 2^63)
    if constexpr ((U::l > 0 && U::l % A::CT_LAYERS == 0) || A::ct_fold_at(U::L0 + U::l)) {
#pragma unroll
        for (int rr = 0; rr < U::NREP; ++rr)
#pragma unroll
            for (int r = 0; r < (1 << U::R); ++r) x[(U::REP0 + rr) * U::STRIDE + r] = A::fold(x[(U::REP0 + rr) * U::STRIDE + r], k);
    }
    if constexpr (U::SHARED) {
#pragma unroll
        for (int b = 0; b < (1 << U::l); ++b) {
            const typename A::TwReg w = A::prep(raw[b]);
#pragma unroll
            for (int rr = 0; rr < U::NREP; ++rr)
#pragma unroll
                for (int j = 0; j < U::half; ++j) {
                    const int o = (U::REP0 + rr) * U::STRIDE + b * 2 * U::half + j;
                    A::ct(x[o], x[o + U::half], w, k);
                }
        }
    } else {
#pragma unroll
        for (int rr = 0; rr < U::NREP; ++rr)
#pragma unroll
            for (int b = 0; b < (1 << U::l); ++b) {
                const typename A::TwReg w = A::prep(raw[(rr << U::l) | b]);
#pragma unroll
                for (int j = 0; j < U::half; ++j) {
                    const int o = (U::REP0 + rr) * U::STRIDE + b * 2 * U::half + j;
                    A::ct(x[o], x[o + U::half], w, k);
                }
            }
    }
}

// inverse: the unit of layer l is step R-1-l of its network; PH = step & 1 pairs layers for the lazy policies, sums are
// folded after every second layer and after the last layer of a network with an odd layer count.  GSTEP >= 0 (kernels that know
// where a unit sits in the WHOLE transform, ntt14w.hpp): policies with a GS_SPAN fold every GS_SPAN-th layer of the transform
// instead (PH = layers since the last fold), whatever the pass boundaries are.
template <class A, class U, int GSTEP = -1, int NTW, int E>
__device__ __forceinline__ void gs_apply(typename A::Elem (&x)[E], const typename A::TwRaw (&raw)[NTW], const typename A::K &k) {
    constexpr bool GLOBAL = GSTEP >= 0 && A::GS_SPAN > 0;
    constexpr int step = U::R - 1 - U::l, PH = GLOBAL ? GSTEP % (A::GS_SPAN > 0 ? A::GS_SPAN : 1) : (step & 1);
    constexpr bool FOLD = A::GS_FOLDS && (GLOBAL ? PH == A::GS_SPAN - 1 : (PH == 1 || step == U::R - 1));
    static_for<0, (U::SHARED ? (1 << U::l) : U::NREP)>([&](auto oc) {
        constexpr int outer = decltype(oc)::value;
        static_for<0, (U::SHARED ? U::NREP : (1 << U::l))>([&](auto ic) {
            constexpr int inner = decltype(ic)::value;
            constexpr int b = U::SHARED ? outer : inner, rr = U::SHARED ? inner : outer;
            const typename A::TwReg w = A::prep(raw[U::SHARED ? b : ((rr << U::l) | b)]);
#pragma unroll
            for (int j = 0; j < U::half; ++j) {
                const int o = (U::REP0 + rr) * U::STRIDE + b * 2 * U::half + j;
                A::template gs<PH>(x[o], x[o + U::half], w, k);
                if constexpr (FOLD) x[o] = A::gs_fold(x[o], k);
            }
        });
    });
}

// one whole network with just-in-time twiddles (a layer's loads stay inside that layer): the generic kernels, the
// 2^15..2^17 passes and the wave-per-ciphertext FHEW kernels
template <class A, int L0, int R, int OFF, int E>
__device__ __forceinline__ void ct_net(typename A::Elem (&x)[E], int top, const typename A::K &k) {
    static_for<0, R>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        typedef Unit<L0, R, l, 1, 1, OFF, true> U;  // one replica at register offset REP0 * STRIDE = OFF
        if constexpr (l > 0) FHE_SCHED_FENCE();
        typename A::TwRaw raw[1 << l];
        tw_load<A, false, U>(raw, top, k);
        ct_apply<A, U>(x, raw, k);
    });
}

template <class A, int L0, int R, int OFF, int E>
__device__ __forceinline__ void gs_net(typename A::Elem (&x)[E], int top, const typename A::K &k) {
    static_for<0, R>([&](auto sc) {
        constexpr int l = R - 1 - decltype(sc)::value;
        typedef Unit<L0, R, l, 1, 1, OFF, true> U;
        if constexpr (l < R - 1) FHE_SCHED_FENCE();
        typename A::TwRaw raw[1 << l];
        tw_load<A, true, U>(raw, top, k);
        gs_apply<A, U>(x, raw, k);
    });
}

}  // namespace fhe
