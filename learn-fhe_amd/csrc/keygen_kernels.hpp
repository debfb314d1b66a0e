// Key-material producers on the device: a counter-based generator and the reference's
// samplers, so that bootstrapping / key-switching keys are built in HBM where the hot path consumes them.
//
// Randomness.  The reference draws from `thread_rng()` (ChaCha12 behind a thread-local handle, seeded with 256 bits of OS entropy),
// one value after another; a GPU needs every lane to find ITS value without a shared state.  Here value i of a draw is word
// (i mod 8) of ChaCha20 block (i div 8) (the block function of RFC 8439 in the ORIGINAL layout: 64-bit block counter, 64-bit nonce;
// known-answer tested on the host, tests/test_abi_cpu.py) under a per-call key derived from an `fhe_rng`'s 256-bit key (the caller's
// 32 bytes, or getrandom's), the call's stream id and a purpose tag (call_key): reproducible, order independent, and exactly as
// strong as the key the caller supplied.  Draws are NOT parity-relevant (the reference's are unseeded): these producers are
// validated at decrypt level and statistically.
#pragma once
#include <atomic>
#include <cmath>
#include "arith.hpp"

namespace fhe {

struct ChaChaKey {
    unsigned k[8];
    unsigned nonce[2];
};

__host__ __device__ __forceinline__ unsigned rotl32(unsigned x, int r) { return (x << r) | (x >> (32 - r)); }

#define FHE_CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7);

// one 64-byte block as eight u64 words
__host__ __device__ inline void chacha20_block(const ChaChaKey &K, unsigned long long counter, unsigned long long (&out)[8]) {
    unsigned s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, K.k[0], K.k[1], K.k[2], K.k[3], K.k[4], K.k[5], K.k[6], K.k[7],
                      (unsigned)counter, (unsigned)(counter >> 32), K.nonce[0], K.nonce[1]};
    unsigned x[16];
    for (int i = 0; i < 16; ++i) x[i] = s[i];
    for (int r = 0; r < 10; ++r) {
        FHE_CHACHA_QR(x[0], x[4], x[8], x[12]) FHE_CHACHA_QR(x[1], x[5], x[9], x[13]) FHE_CHACHA_QR(x[2], x[6], x[10], x[14]) FHE_CHACHA_QR(x[3], x[7], x[11], x[15])
        FHE_CHACHA_QR(x[0], x[5], x[10], x[15]) FHE_CHACHA_QR(x[1], x[6], x[11], x[12]) FHE_CHACHA_QR(x[2], x[7], x[8], x[13]) FHE_CHACHA_QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 8; ++i) out[i] = (unsigned long long)(x[2 * i] + s[2 * i]) | ((unsigned long long)(x[2 * i + 1] + s[2 * i + 1]) << 32);
}

// What a draw is FOR.  The key a call draws under is derived from the generator's 256-bit key, the caller's stream id AND this
// tag (call_key below), so two entry points never share keystream even when a caller reuses (generator, stream id): a secret key
// sampled with fhe_sample_binary(rng, 7) shares nothing with the public mask of an fhe_tlwe_sk_encrypt(rng, 7).
enum RngPurpose : unsigned {
    RNG_SAMPLE_UNIFORM = 1, RNG_SAMPLE_TORUS, RNG_SAMPLE_DG, RNG_SAMPLE_ZO, RNG_SAMPLE_TDG, RNG_SAMPLE_BINARY,
    RNG_RLWE_ENC, RNG_RGSW_ENC, RNG_RLWE_KSK, RNG_LWE_ENC, RNG_LWE_KSK, RNG_CKKS_ENC, RNG_CKKS_KSK,
    RNG_TLWE_ENC, RNG_TLWE_KSK, RNG_TGLWE_ENC, RNG_TGGSW_ENC, RNG_TGLWEK_ENC, RNG_TGGSWK_ENC, RNG_CKKS_PK_ENC, RNG_RLWE_SHARE, RNG_RLWE_PK_ENC, RNG_LWE_SHARE, RNG_LWE_KSK_SHARE
};
}  // namespace fhe

// the generator behind every key-material entry point (include/fhe_ring.h: fhe_rng_create)
struct fhe_rng {
    unsigned key[8];                                   // 256 bits: from the caller, or from the operating system
    mutable std::atomic<unsigned long long> next{0};   // FHE_STREAM_AUTO: a fresh stream id per call
};

namespace fhe {
// per-call key = the first 32 bytes of the ChaCha20 block under the generator key with nonce (purpose, "fhek") and the stream id
// as the block counter: a PRF of (key, stream id, purpose).  Inside a call the block counter partitions that key's stream
// between the call's draws (mask, noise, ...): distinct blocks of one ChaCha stream are independent.
inline ChaChaKey call_key(const fhe_rng *rng, unsigned long long stream_id, unsigned purpose) {
    ChaChaKey M;
    for (int i = 0; i < 8; ++i) M.k[i] = rng->key[i];
    M.nonce[0] = purpose; M.nonce[1] = 0x6b656866u;  // "fhek"
    if (stream_id == ~0ull) stream_id = (1ull << 63) | rng->next.fetch_add(1, std::memory_order_relaxed);  // FHE_STREAM_AUTO
    unsigned long long w[8];
    chacha20_block(M, stream_id, w);
    ChaChaKey K;
    for (int i = 0; i < 4; ++i) { K.k[2 * i] = (unsigned)w[i]; K.k[2 * i + 1] = (unsigned)(w[i] >> 32); }
    K.nonce[0] = 0; K.nonce[1] = 0;
    return K;
}
// TESTS ONLY (fhe_rng_create_from_seed): 64 bits of entropy stretched to a key by SplitMix64
inline void seed_to_key(unsigned long long seed, unsigned (&key)[8]) {
    unsigned long long z = seed;
    for (int i = 0; i < 4; ++i) {
        z += 0x9E3779B97F4A7C15ull;
        unsigned long long v = z;
        v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull; v = (v ^ (v >> 27)) * 0x94D049BB133111EBull; v ^= v >> 31;
        key[2 * i] = (unsigned)v; key[2 * i + 1] = (unsigned)(v >> 32);
    }
}

// util/src/zq.rs:91-93 `Zq::sample_uniform` (rand's `Uniform::new(0, q)`): uniform in [0, q).  Two 64-bit words per value, reduced
// as a 128-bit integer: bias below 2^-64 (rand's own sampler rejects instead; both are exact to any observable precision).
FHE_HEADER_KERNEL void sample_uniform_kernel(u64 *__restrict__ out, size_t count, Barrett B, ChaChaKey K, unsigned long long first) {
    const u64 two64 = (u64)((((unsigned __int128)1) << 64) % B.q);  // per thread, set-up code
    for (size_t blk = blockIdx.x * size_t(blockDim.x) + threadIdx.x; blk * 4 < count; blk += size_t(gridDim.x) * blockDim.x) {
        unsigned long long w[8];
        chacha20_block(K, first + blk, w);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = blk * 4 + j;
            if (i < count) {
                const u64 hi = w[2 * j] % B.q, lo = w[2 * j + 1] % B.q;
                out[i] = csub(mulmod_barrett(hi, two64, B) + lo, B.q);
            }
        }
    }
}
// torus values: uniform u64 (util/src/torus.rs `T64::sample_uniform`)
FHE_HEADER_KERNEL void sample_u64_kernel(u64 *__restrict__ out, size_t count, ChaChaKey K, unsigned long long first) {
    for (size_t blk = blockIdx.x * size_t(blockDim.x) + threadIdx.x; blk * 8 < count; blk += size_t(gridDim.x) * blockDim.x) {
        unsigned long long w[8];
        chacha20_block(K, first + blk, w);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (blk * 8 + j < count) out[blk * 8 + j] = w[j];
    }
}

// util/src/misc/distribution.rs:23-46 `dg(std_dev, n)`: integers in [-max, max], max = floor(n std_dev), weight of i =
// cdf(i + 0.5) - cdf(i - 0.5) with the reference's erf approximation (Abramowitz-Stegun 7.1.26); `WeightedIndex` = inverse
// transform on the cumulative weights, which the host tabulates (cum[j] = sum of the first j + 1 weights, cum[2 max] = total).
// Output as `Zq::from_i64` (zq.rs:63-69) when q != 0, as two's-complement i64 otherwise.
constexpr int DG_MAX_TABLE = 129;
struct DgTable {
    double cum[DG_MAX_TABLE];
    int len, max;
};
// the table itself, evaluated on the host exactly as the reference does (distribution.rs:25-45); false: unsupported shape
inline bool make_dg_table(double std_dev, int n_sigma, DgTable *T) {
    auto erf_as = [](double x) {
        const double p = 0.3275911, a1 = 0.254829592, a2 = -0.284496736, a3 = 1.421413741, a4 = -1.453152027, a5 = 1.061405429;
        const double t = 1.0 / (1.0 + p * std::fabs(x));
        const double pos = 1.0 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * std::exp(-x * x);
        return std::signbit(x) ? -pos : pos;
    };
    auto cdf = [&](double x) { return (1.0 + erf_as(x / (std_dev * 1.4142135623730951))) / 2.0; };
    const long long mx = (long long)std::floor((double)n_sigma * std_dev);
    if (mx < 0 || 2 * mx + 1 > DG_MAX_TABLE) return false;
    T->max = (int)mx; T->len = (int)(2 * mx + 1);
    double acc = 0;
    for (long long i = -mx; i <= mx; ++i) {
        acc += cdf((double)i + 0.5) - cdf((double)i - 0.5);
        T->cum[i + mx] = acc;
    }
    return true;
}
FHE_HEADER_KERNEL void sample_dg_kernel(u64 *__restrict__ out, size_t count, u64 q, DgTable T, ChaChaKey K, unsigned long long first) {
    for (size_t blk = blockIdx.x * size_t(blockDim.x) + threadIdx.x; blk * 8 < count; blk += size_t(gridDim.x) * blockDim.x) {
        unsigned long long w[8];
        chacha20_block(K, first + blk, w);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const size_t i = blk * 8 + j;
            if (i >= count) continue;
            const double u = (double)(w[j] >> 11) * (1.0 / 9007199254740992.0) * T.cum[T.len - 1];  // [0, total)
            int idx = 0;
            while (idx < T.len - 1 && u >= T.cum[idx]) ++idx;
            const long long v = (long long)idx - T.max;
            out[i] = q ? (v < 0 ? q - (u64)(-v) % q : (u64)v % q) % q : (u64)v;
        }
    }
}

// util/src/misc/decompose.rs:35-40 `power_up`: out[p][j][i] = in[p][i] * base_j mod q, base_j = 2^(rounding_bits + j log_b) mod q
FHE_HEADER_KERNEL void power_up_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t n, size_t polys, int d, int rounding_bits, int log_b,
                                       Barrett B, int negate) {
    const size_t total = n * polys * d;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t i = idx % n, j = (idx / n) % d, p = idx / (n * d);
        u64 base = 1;  // 2^(rb + j log_b) mod q by repeated doubling (d, log_b small: set-up code)
        for (int s = 0; s < rounding_bits + (int)j * log_b; ++s) base = csub(base + base, B.q);
        u64 v = in[p * n + i];
        if (negate) v = v ? B.q - v : 0;
        out[idx] = mulmod_barrett(v, base, B);
    }
}

// b <- b + e + pt (rlwe.rs:153), optionally pt = 0 (null)
FHE_HEADER_KERNEL void add3_kernel(u64 *__restrict__ b, const u64 *__restrict__ e, const u64 *__restrict__ pt, size_t count, size_t pt_mod, u64 q) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < count; i += size_t(gridDim.x) * blockDim.x) {
        u64 v = csub(b[i] + e[i], q);
        if (pt) v = csub(v + pt[i % pt_mod], q);
        b[i] = v;
    }
}
// small two's-complement integers (zo / dg draws) as residues: `Zq::from_i64` (zq.rs:63-69) for |x| < q, in place
FHE_HEADER_KERNEL void small_i64_to_zq_kernel(u64 *__restrict__ x, size_t count, u64 q) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < count; i += size_t(gridDim.x) * blockDim.x) {
        const u64 v = x[i];
        x[i] = (long long)v < 0 ? q - (0 - v) : v;
    }
}
// out[i] = b[i] - out[i] mod q (rlwe.rs:172-175 `decrypt`: b - a sk)
FHE_HEADER_KERNEL void rsub_kernel(u64 *__restrict__ out, const u64 *__restrict__ b, size_t count, u64 q) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < count; i += size_t(gridDim.x) * blockDim.x) {
        const u64 x = b[i], y = out[i];
        out[i] = x >= y ? x - y : x + q - y;
    }
}
// out[i] += add[i] over a strided set of polynomials (rgsw.rs:101-103: `ct.0 += pt` on rows 0..d, `ct.1 += pt` on rows d..2d)
FHE_HEADER_KERNEL void add_assign_kernel(u64 *__restrict__ out, const u64 *__restrict__ add, size_t count, u64 q) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < count; i += size_t(gridDim.x) * blockDim.x) out[i] = csub(out[i] + add[i], q);
}

// util/src/misc/distribution.rs:10-21 `zo(rho)`: -1 with probability rho / 2, +1 with rho / 2, 0 otherwise; two's-complement i64
// (scheme/ckks/src/ckks.rs:139-141 `Ckks::sk_gen`: rho = 0.5)
FHE_HEADER_KERNEL void sample_zo_kernel(u64 *__restrict__ out, size_t count, double rho, ChaChaKey K, unsigned long long first) {
    for (size_t blk = blockIdx.x * size_t(blockDim.x) + threadIdx.x; blk * 8 < count; blk += size_t(gridDim.x) * blockDim.x) {
        unsigned long long w[8];
        chacha20_block(K, first + blk, w);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (blk * 8 + j >= count) continue;
            const double v = (double)(w[j] >> 11) * (1.0 / 9007199254740992.0);  // `Standard` f64: [0, 1)
            out[blk * 8 + j] = v <= rho / 2.0 ? ~0ull : (v <= rho ? 1ull : 0ull);
        }
    }
}

// util/src/ring/rns.rs:61-63 `RnsRq::from_i64`: out[l][i] = v[i] mod m_l (zq.rs:63-69), v two's-complement i64, times mult[l] if given
FHE_HEADER_KERNEL void rns_from_i64_kernel(const u64 *__restrict__ v, u64 *__restrict__ out, size_t n, int limbs, const Barrett *__restrict__ B,
                                           const u64 *__restrict__ mult, size_t polys = 1) {
    const size_t total = n * limbs * polys;  // v [polys][n] -> out [polys][limbs][n]
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t i = idx % n, l = (idx / n) % limbs, c = idx / (n * limbs);
        const Barrett b = B[l];
        const long long x = (long long)v[c * n + i];
        u64 r = x < 0 ? (u64)(-x) % b.q : (u64)x % b.q;
        if (x < 0 && r) r = b.q - r;
        out[idx] = mult ? mulmod_barrett(r, mult[l], b) : r;
    }
}
// x <- (negate ? -x : x) + e + add over [batch][limbs][n]; e [batch][n] i64 or null; add [add_batch or 1][limbs][n] or null.
// ckks.rs:222 `sk_encrypt` (b = -(a s) + e + pt: negate), 235-236 `pk_encrypt` (pk u + e (+ pt)), 246 `decrypt` (a s + b)
FHE_HEADER_KERNEL void ckks_finish_b_kernel(u64 *__restrict__ b, const u64 *__restrict__ e, const u64 *__restrict__ pt, size_t n, int limbs, size_t batch,
                                            size_t pt_batch, const Barrett *__restrict__ B, int negate) {
    const size_t total = batch * limbs * n;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t i = idx % n, l = (idx / n) % limbs, c = idx / (n * limbs);
        const u64 q = B[l].q;
        u64 ev = 0;
        if (e) {
            const long long x = (long long)e[c * n + i];
            ev = x < 0 ? (u64)(-x) % q : (u64)x % q;
            if (x < 0 && ev) ev = q - ev;
        }
        const u64 as = b[idx];
        u64 r = csub((negate ? (as ? q - as : 0) : as) + ev, q);
        if (pt) r = csub(r + pt[((c % pt_batch) * limbs + l) * n + i], q);
        b[idx] = r;
    }
}
// max |x_i| of a two's-complement i64 vector, atomically into *out (a bound check on secret keys; set-up code)
FHE_HEADER_KERNEL void max_abs_i64_kernel(const u64 *__restrict__ in, size_t n, unsigned long long *out) {
    unsigned long long m = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const long long v = (long long)in[i];
        const unsigned long long a = v < 0 ? (unsigned long long)0 - (unsigned long long)v : (unsigned long long)v;
        m = a > m ? a : m;
    }
    atomicMax(out, m);
}
// centred lift of a polynomial mod q to two's-complement i64 (the integer square of a small secret key, ckks.rs:78-80)
FHE_HEADER_KERNEL void centre_to_i64_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t n, u64 q) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const u64 v = in[i];
        out[i] = v > q / 2 ? (u64)(-(long long)(q - v)) : v;
    }
}

// util/src/misc/distribution.rs:49-54 `tdg(std_dev)`: v ~ N(0, std_dev), frac = v - round(v), T64 = round(frac 2^64) as i64
// (Rust's float -> int cast saturates).  The normal deviate is Box-Muller on two words of the generator: the reference's
// `rand_distr::Normal` is a different sampler of the same distribution, and its draws are unseeded.
FHE_HEADER_KERNEL void sample_tdg_kernel(u64 *__restrict__ out, size_t count, double std_dev, ChaChaKey K, unsigned long long first) {
    for (size_t blk = blockIdx.x * size_t(blockDim.x) + threadIdx.x; blk * 4 < count; blk += size_t(gridDim.x) * blockDim.x) {
        unsigned long long w[8];
        chacha20_block(K, first + blk, w);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t i = blk * 4 + j;
            if (i >= count) continue;
            const double u1 = (double)((w[2 * j] >> 11) + 1) * (1.0 / 9007199254740992.0);  // (0, 1]
            const double u2 = (double)(w[2 * j + 1] >> 11) * (1.0 / 9007199254740992.0);    // [0, 1)
            const double v = std_dev * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
            const double x = round((v - round(v)) * 18446744073709551616.0);
            const long long r = x >= 9223372036854775807.0 ? 0x7fffffffffffffffll : (x <= -9223372036854775808.0 ? (long long)0x8000000000000000ull : (long long)x);
            out[i] = (u64)r;
        }
    }
}
// util/src/misc/distribution.rs `binary()`: uniform bits, one per output word (scheme/tfhe/src/tlwe.rs:96-98 `sk_gen`)
FHE_HEADER_KERNEL void sample_binary_kernel(u64 *__restrict__ out, size_t count, ChaChaKey K, unsigned long long first) {
    for (size_t blk = blockIdx.x * size_t(blockDim.x) + threadIdx.x; blk * 512 < count; blk += size_t(gridDim.x) * blockDim.x) {
        unsigned long long w[8];
        chacha20_block(K, first + blk, w);
        for (int j = 0; j < 512; ++j)
            if (blk * 512 + j < count) out[blk * 512 + j] = (w[j >> 6] >> (j & 63)) & 1;
    }
}

// scheme/tfhe/src/tlwe.rs:122-132 `Tlwe::sk_encrypt` for `rows` plaintexts on the torus (wrapping u64): b[r] = <a[r], sk> + e[r] + pt[r].
// sk1 != null: pt[r] = -sk1[r % n1] * 2^(rounding_bits + (r / n1) log_b) (tlwe.rs:100-111 `ksk_gen`: `power_up(-sk1).flatten()`)
FHE_HEADER_KERNEL void tlwe_encrypt_kernel(const u64 *__restrict__ a, const u64 *__restrict__ sk, const u64 *__restrict__ e, const u64 *__restrict__ pt,
                                           u64 *__restrict__ b, size_t n, size_t rows, const u64 *__restrict__ sk1, size_t n1, int rounding_bits, int log_b) {
    for (size_t r = blockIdx.x * size_t(blockDim.x) + threadIdx.x; r < rows; r += size_t(gridDim.x) * blockDim.x) {
        u64 acc = 0;
        for (size_t j = 0; j < n; ++j) acc += a[r * n + j] * sk[j];
        u64 p;
        if (sk1) p = (0 - sk1[r % n1]) << (rounding_bits + (int)(r / n1) * log_b);
        else p = pt ? pt[r] : 0;
        b[r] = acc + e[r] + p;
    }
}
// torus polynomials: b <- b + e + pt (tglwe.rs:101), pt [pt_rows][n] cycled or null
FHE_HEADER_KERNEL void torus_add3_kernel(u64 *__restrict__ b, const u64 *__restrict__ e, const u64 *__restrict__ pt, size_t count, size_t pt_mod) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < count; i += size_t(gridDim.x) * blockDim.x)
        b[i] += e[i] + (pt ? pt[i % pt_mod] : 0);
}
// scheme/tfhe/src/tggsw.rs:80-87 (k = 1): rows [count][2d][n]; row j < d: a += pt 2^(rb + j log_b); row d + j: b += the same
FHE_HEADER_KERNEL void tggsw_add_gadget_kernel(u64 *__restrict__ rows_a, u64 *__restrict__ rows_b, const u64 *__restrict__ pt, size_t n, size_t count, int d,
                                               int rounding_bits, int log_b) {
    const size_t total = count * d * n;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const size_t i = idx % n, j = (idx / n) % d, c = idx / (n * d);
        const u64 v = pt[c * n + i] << (rounding_bits + (int)j * log_b);
        rows_a[(c * 2 * d + j) * n + i] += v;
        rows_b[(c * 2 * d + d + j) * n + i] += v;
    }
}

// scheme/fhew/src/lwe.rs:128-139 `Lwe::sk_encrypt` for `rows` plaintexts: b[r] = <a[r], sk> + pt[r] + e[r] over any modulus q < 2^62
// (q_ks = 2^16 in the reference's parameter sets).  NEG_POWER_UP: pt[r] = -sk1[r % n1] * base_{r / n1} (lwe.rs:108-119 `ksk_gen`:
// `power_up(-sk1).flatten()`, digit-major), computed here instead of read.
FHE_HEADER_KERNEL void lwe_encrypt_kernel(const u64 *__restrict__ a, const u64 *__restrict__ sk, const u64 *__restrict__ e, const u64 *__restrict__ pt,
                                          u64 *__restrict__ b, size_t n, size_t rows, Barrett B, const u64 *__restrict__ sk1, size_t n1, int rounding_bits,
                                          int log_b) {
    for (size_t r = blockIdx.x * size_t(blockDim.x) + threadIdx.x; r < rows; r += size_t(gridDim.x) * blockDim.x) {
        u64 acc = 0;
        for (size_t j = 0; j < n; ++j) acc = csub(acc + mulmod_barrett(a[r * n + j], sk[j], B), B.q);
        u64 p;
        if (sk1) {
            u64 base = 1 % B.q;
            for (int s = 0; s < rounding_bits + (int)(r / n1) * log_b; ++s) base = csub(base + base, B.q);
            const u64 v = sk1[r % n1];
            p = mulmod_barrett(v ? B.q - v : 0, base, B);
        } else {
            p = pt ? pt[r] : 0;
        }
        b[r] = csub(csub(acc + p, B.q) + e[r], B.q);
    }
}

// util/src/ring.rs:328-341 `Rq: Sum`: out[i] = sum_k in[k][i] mod q over `count` polynomials (any modulus q < 2^62)
FHE_HEADER_KERNEL void rq_sum_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t len, size_t count, u64 q) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < len; i += size_t(gridDim.x) * blockDim.x) {
        u64 acc = 0;
        for (size_t k = 0; k < count; ++k) acc = csub(acc + in[k * len + i], q);
        out[i] = acc;
    }
}

}  // namespace fhe
