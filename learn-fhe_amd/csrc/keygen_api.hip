// extern "C" entry points of the key-material producers: samplers, power_up, RLWE secret-key
// encryption, RGSW / key-switching / automorphism key generation, all on the device.
#include <hip/hip_runtime.h>

#include <sys/random.h>

#include <cmath>
#include <cstring>
#include <new>

#include "../../include/fhe_ring.h"
#include "api_common.hpp"
#include "ctx.hpp"
#include "keygen_kernels.hpp"

using fhe::u64;

namespace {
inline unsigned grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b ? b : 1));
}

// distribution.rs:25-45 (keygen_kernels.hpp: fhe::make_dg_table) with the entry points' status codes
int dg_table_rc(double std_dev, int n_sigma, fhe::DgTable *T) {
    if (!(std_dev > 0) || n_sigma < 1) return FHE_ERR_INVALID;
    return fhe::make_dg_table(std_dev, n_sigma, T) ? FHE_OK : FHE_ERR_UNSUPPORTED;
}

// blocks of the generator a draw of `count` values occupies: callers advance `first` by this to chain independent draws on one stream id
inline unsigned long long blocks_uniform(size_t count) { return (count + 3) / 4; }
inline unsigned long long blocks_words(size_t count) { return (count + 7) / 8; }

int sample_uniform_dev(u64 q, const fhe::ChaChaKey &K, unsigned long long first, u64 *out, size_t count, hipStream_t st) {
    hipLaunchKernelGGL(fhe::sample_uniform_kernel, dim3(grid_for(blocks_uniform(count))), dim3(256), 0, st, out, count, fhe::make_barrett(q), K, first);
    return hipGetLastError() == hipSuccess ? FHE_OK : FHE_ERR_HIP;
}
int sample_dg_dev(u64 q, const fhe::DgTable &T, const fhe::ChaChaKey &K, unsigned long long first, u64 *out, size_t count, hipStream_t st) {
    hipLaunchKernelGGL(fhe::sample_dg_kernel, dim3(grid_for(blocks_words(count))), dim3(256), 0, st, out, count, q, T, K, first);
    return hipGetLastError() == hipSuccess ? FHE_OK : FHE_ERR_HIP;
}

// rlwe.rs:146-156 for `rows` ciphertexts under one secret key, device buffers: a uniform, e <- dg(3.2, 6), b = a sk + e + pt
// (pt: [pt_rows][n] cycled, or null for encryptions of zero).  sk_eval: the secret key in the evaluation domain.
int rlwe_sk_encrypt_dev(const fhe_ctx *ctx, const u64 *sk_eval, const u64 *pt, size_t pt_rows, u64 *ct_a, u64 *ct_b, size_t n, size_t rows,
                        const fhe::ChaChaKey &K, unsigned long long *cursor, hipStream_t st) {
    const size_t count = rows * n;
    const int log_n = ilog2(n);
    fhe::DgTable T;
    int rc = dg_table_rc(3.2, 6, &T);
    if (rc != FHE_OK) return rc;
    StreamWs we(count * sizeof(u64), st);
    if (we.rc != FHE_OK) return we.rc;
    u64 *e = we.as<u64>();
    rc = sample_uniform_dev(ctx->q, K, *cursor, ct_a, count, st);
    *cursor += blocks_uniform(count);
    if (rc == FHE_OK) rc = sample_dg_dev(ctx->q, T, K, *cursor, e, count, st);
    *cursor += blocks_words(count);
    // b = a * sk: forward transform of a out of place into b, inverse with the (broadcast) evaluation-domain key on its load
    if (rc == FHE_OK && n > 1) {
        fhe::NttIo src;
        src.src = ct_a; src.src_mod = (unsigned)rows;
        rc = fhe::ntt_fwd_multi(ctx->d_desc, 1, ct_b, log_n, rows, st, ctx->pm_b, src);
        if (rc == FHE_OK) {
            fhe::NttIo mul;
            mul.mul = sk_eval; mul.mul_div = (unsigned)rows; mul.mul_period = 1;
            rc = fhe::ntt_inv_multi(ctx->d_desc, 1, ct_b, log_n, rows, st, ctx->pm_b, mul);
        }
    } else if (rc == FHE_OK) {
        return FHE_ERR_UNSUPPORTED;
    }
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::add3_kernel, dim3(grid_for(count)), dim3(256), 0, st, ct_b, (const u64 *)e, pt, count, pt ? pt_rows * n : 1, (u64)ctx->q);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    return rc;
}

int check_ring(const fhe_ctx *ctx, size_t n) {
    if (!ctx || !is_pow2(n) || n < 2) return FHE_ERR_INVALID;
    if (ctx->device < 0) return FHE_ERR_NO_DEVICE;
    const int log_n = ilog2(n);
    if (log_n > ctx->s - 1) return FHE_ERR_NO_ROOT;
    if (log_n > ctx->log_cap) return FHE_ERR_UNSUPPORTED;
    return FHE_OK;
}

int gadget_geometry(u64 q, int log_b, int d, int *rounding_bits) {
    if (q < 2 || log_b < 1 || log_b > 62 || d < 1 || d > 64) return FHE_ERR_INVALID;
    const int log_q = q <= 1 ? 0 : 64 - __builtin_clzll(q - 1);  // q.next_power_of_two().ilog2() (decompose.rs:49-64)
    *rounding_bits = log_q - log_b * d > 0 ? log_q - log_b * d : 0;
    return FHE_OK;
}
}  // namespace

extern "C" {

// ---- the generator (include/fhe_ring.h) ------------------------------------------------------------------------------------
int fhe_rng_create(const uint8_t *key32, fhe_rng **out) {
    if (!out) return FHE_ERR_INVALID;
    *out = nullptr;
    fhe_rng *r = new (std::nothrow) fhe_rng();
    if (!r) return FHE_ERR_INVALID;
    unsigned char buf[32];
    if (key32) {
        std::memcpy(buf, key32, 32);
    } else {  // 256 bits from the operating system, as the reference's thread_rng() is seeded
        size_t got = 0;
        while (got < 32) {
            const ssize_t k = getrandom(buf + got, 32 - got, 0);
            if (k <= 0) { delete r; return FHE_ERR_UNSUPPORTED; }
            got += (size_t)k;
        }
    }
    for (int i = 0; i < 8; ++i) r->key[i] = (unsigned)buf[4 * i] | ((unsigned)buf[4 * i + 1] << 8) | ((unsigned)buf[4 * i + 2] << 16) | ((unsigned)buf[4 * i + 3] << 24);
    *out = r;
    return FHE_OK;
}
int fhe_rng_create_from_seed(uint64_t seed, fhe_rng **out) {  // tests and reproducible runs ONLY: 64 bits of entropy
    if (!out) return FHE_ERR_INVALID;
    fhe_rng *r = new (std::nothrow) fhe_rng();
    if (!r) { *out = nullptr; return FHE_ERR_INVALID; }
    fhe::seed_to_key(seed, r->key);
    *out = r;
    return FHE_OK;
}
void fhe_rng_destroy(fhe_rng *rng) { delete rng; }
// the block function itself, for known-answer tests (host only): key 32 bytes little endian, 64-bit nonce, 64-bit block counter
int fhe_chacha20_block(const uint8_t *key32, uint64_t nonce, uint64_t counter, uint8_t *out64) {
    if (!key32 || !out64) return FHE_ERR_INVALID;
    fhe::ChaChaKey K;
    for (int i = 0; i < 8; ++i) K.k[i] = (unsigned)key32[4 * i] | ((unsigned)key32[4 * i + 1] << 8) | ((unsigned)key32[4 * i + 2] << 16) | ((unsigned)key32[4 * i + 3] << 24);
    K.nonce[0] = (unsigned)nonce; K.nonce[1] = (unsigned)(nonce >> 32);
    unsigned long long w[8];
    fhe::chacha20_block(K, counter, w);
    for (int i = 0; i < 8; ++i)
        for (int b = 0; b < 8; ++b) out64[8 * i + b] = (uint8_t)(w[i] >> (8 * b));
    return FHE_OK;
}

int fhe_sample_uniform(uint64_t q, const fhe_rng *rng, uint64_t stream_id, uint64_t *out, size_t count, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    if (q < 2 || (q >> 62) || (!out && count)) return FHE_ERR_INVALID;
    if (count == 0) return FHE_OK;
    PtrDeviceGuard pguard(out, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    Mirror mo(out, count, mem, false, st);
    if (mo.rc != FHE_OK) return mo.rc;
    int rc = sample_uniform_dev(q, fhe::call_key(rng, stream_id, fhe::RNG_SAMPLE_UNIFORM), 0, mo.d, count, st);
    return rc == FHE_OK ? mo.sync_out(st) : rc;
}

int fhe_sample_torus(const fhe_rng *rng, uint64_t stream_id, uint64_t *out, size_t count, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    if (!out && count) return FHE_ERR_INVALID;
    if (count == 0) return FHE_OK;
    PtrDeviceGuard pguard(out, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    Mirror mo(out, count, mem, false, st);
    if (mo.rc != FHE_OK) return mo.rc;
    hipLaunchKernelGGL(fhe::sample_u64_kernel, dim3(grid_for(blocks_words(count))), dim3(256), 0, st, mo.d, count, fhe::call_key(rng, stream_id, fhe::RNG_SAMPLE_TORUS), 0ull);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

int fhe_sample_dg(uint64_t q, double std_dev, int n_sigma, const fhe_rng *rng, uint64_t stream_id, uint64_t *out, size_t count, fhe_mem mem,
                  void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    if ((q >> 62) || q == 1 || (!out && count)) return FHE_ERR_INVALID;
    fhe::DgTable T;
    int rc = dg_table_rc(std_dev, n_sigma, &T);
    if (rc != FHE_OK) return rc;
    if (count == 0) return FHE_OK;
    PtrDeviceGuard pguard(out, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    Mirror mo(out, count, mem, false, st);
    if (mo.rc != FHE_OK) return mo.rc;
    rc = sample_dg_dev(q, T, fhe::call_key(rng, stream_id, fhe::RNG_SAMPLE_DG), 0, mo.d, count, st);
    return rc == FHE_OK ? mo.sync_out(st) : rc;
}

int fhe_power_up(uint64_t q, int log_b, int d, const uint64_t *in, size_t n, size_t polys, uint64_t *out, fhe_mem mem, void *stream) {
    int rb = 0;
    int rc = gadget_geometry(q, log_b, d, &rb);
    if (rc != FHE_OK) return rc;
    if ((q >> 62) || ((!in || !out) && n * polys)) return FHE_ERR_INVALID;
    if (n * polys == 0) return FHE_OK;
    PtrDeviceGuard pguard(in, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    Mirror mi(in, n * polys, mem, true, st), mo(out, n * polys * d, mem, false, st);
    if (mi.rc | mo.rc) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::power_up_kernel, dim3(grid_for(n * polys * d)), dim3(256), 0, st, (const u64 *)mi.d, mo.d, n, polys, d, rb, log_b,
                       fhe::make_barrett(q), 0);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

int fhe_rlwe_sk_encrypt(const fhe_ctx *ctx, const uint64_t *sk, const uint64_t *pt, size_t n, size_t batch, const fhe_rng *rng, uint64_t stream_id,
                        uint64_t *ct_a, uint64_t *ct_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rc = check_ring(ctx, n);
    if (rc != FHE_OK) return rc;
    if (!sk || ((!ct_a || !ct_b) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    Mirror msk(sk, n, mem, true, st), mpt(pt, pt ? n * batch : 0, mem, true, st), ma(ct_a, n * batch, mem, false, st), mb(ct_b, n * batch, mem, false, st);
    if (msk.rc | mpt.rc | ma.rc | mb.rc) return FHE_ERR_HIP;
    StreamWs wsk(n * sizeof(u64), st);
    if (wsk.rc != FHE_OK) return wsk.rc;
    fhe::NttIo src;
    src.src = msk.d; src.src_mod = 1;
    rc = fhe::ntt_fwd_multi(ctx->d_desc, 1, wsk.as<u64>(), ilog2(n), 1, st, ctx->pm_b, src);
    unsigned long long cursor = 0;
    if (rc == FHE_OK) rc = rlwe_sk_encrypt_dev(ctx, wsk.as<u64>(), pt ? mpt.d : nullptr, batch, ma.d, mb.d, n, batch, fhe::call_key(rng, stream_id, fhe::RNG_RLWE_ENC), &cursor, st);
    if (rc == FHE_OK) rc = ma.sync_out(st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    return rc;
}

// scheme/fhew/src/rlwe.rs:237-249 `Rlwe::share_encrypt(param, a, sk, pt)`: b = a sk + e + pt for a GIVEN mask a -- what `pk_share_gen`
// (217-225: pt = 0, the common reference string as a), `ksk_share_gen` / `ak_share_gen` (276-303: one call per gadget row) and
// `share_decrypt` (260-269: pt = 0, a = the ciphertext's mask) are made of.  a [rows][n] (or ONE polynomial shared by all rows when
// a_rows = 1), pt [rows][n] or NULL; out_b [rows][n]
int fhe_rlwe_share_encrypt(const fhe_ctx *ctx, const uint64_t *a, size_t a_rows, const uint64_t *sk, const uint64_t *pt, size_t n, size_t rows, const fhe_rng *rng,
                           uint64_t stream_id, uint64_t *out_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rc = check_ring(ctx, n);
    if (rc != FHE_OK) return rc;
    if (!sk || !a || (a_rows != 1 && a_rows != rows) || (!out_b && rows)) return FHE_ERR_INVALID;
    if (rows == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t count = rows * n;
    Mirror msk(sk, n, mem, true, st), ma(a, a_rows * n, mem, true, st), mpt(pt, pt ? count : 0, mem, true, st), mb(out_b, count, mem, false, st);
    if (msk.rc | ma.rc | mpt.rc | mb.rc) return FHE_ERR_HIP;
    StreamWs ws((n + count) * sizeof(u64), st);  // sk in the evaluation domain | the noise
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *sk_eval = ws.as<u64>(), *e = sk_eval + n;
    const int log_n = ilog2(n);
    fhe::NttIo src;
    src.src = msk.d; src.src_mod = 1;
    rc = fhe::ntt_fwd_multi(ctx->d_desc, 1, sk_eval, log_n, 1, st, ctx->pm_b, src);
    fhe::DgTable T;
    if (rc == FHE_OK) rc = dg_table_rc(3.2, 6, &T);
    if (rc == FHE_OK) rc = sample_dg_dev(ctx->q, T, fhe::call_key(rng, stream_id, fhe::RNG_RLWE_SHARE), 0, e, count, st);
    if (rc == FHE_OK) {
        fhe::NttIo sa;
        sa.src = ma.d; sa.src_mod = (unsigned)a_rows;
        rc = fhe::ntt_fwd_multi(ctx->d_desc, 1, mb.d, log_n, rows, st, ctx->pm_b, sa);
    }
    if (rc == FHE_OK) {
        fhe::NttIo mul;
        mul.mul = sk_eval; mul.mul_div = (unsigned)rows; mul.mul_period = 1;
        rc = fhe::ntt_inv_multi(ctx->d_desc, 1, mb.d, log_n, rows, st, ctx->pm_b, mul);
    }
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::add3_kernel, dim3(grid_for(count)), dim3(256), 0, st, mb.d, (const u64 *)e, pt ? (const u64 *)mpt.d : nullptr, count, count, (u64)ctx->q);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    return rc == FHE_OK ? mb.sync_out(st) : rc;
}

// rlwe.rs:158-170 `Rlwe::pk_encrypt` for `batch` plaintexts: u <- zo(0.5), e0, e1 <- dg(3.2, 6) per ciphertext; a = pk.a u + e0,
// b = pk.b u + e1 + pt.  pk_a, pk_b [n] (rlwe.rs:98-101 `pk_gen` = fhe_rlwe_sk_encrypt of zero, or a merged multi-party key); pt NULL = zeros
int fhe_rlwe_pk_encrypt(const fhe_ctx *ctx, const uint64_t *pk_a, const uint64_t *pk_b, const uint64_t *pt, size_t n, size_t batch, const fhe_rng *rng,
                        uint64_t stream_id, uint64_t *ct_a, uint64_t *ct_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rc = check_ring(ctx, n);
    if (rc != FHE_OK) return rc;
    if (!pk_a || !pk_b || ((!ct_a || !ct_b) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t count = batch * n;
    Mirror mpa(pk_a, n, mem, true, st), mpb(pk_b, n, mem, true, st), mpt(pt, pt ? count : 0, mem, true, st), ma(ct_a, count, mem, false, st),
        mb(ct_b, count, mem, false, st);
    if (mpa.rc | mpb.rc | mpt.rc | ma.rc | mb.rc) return FHE_ERR_HIP;
    StreamWs ws((2 * n + 2 * count) * sizeof(u64), st);  // pk.a, pk.b in the evaluation domain | e0 | e1
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *pk_eval = ws.as<u64>(), *e = pk_eval + 2 * n;
    const int log_n = ilog2(n);
    const fhe::ChaChaKey K = fhe::call_key(rng, stream_id, fhe::RNG_RLWE_PK_ENC);
    fhe::NttIo src;
    src.src = mpa.d; src.src_mod = 1;
    rc = fhe::ntt_fwd_multi(ctx->d_desc, 1, pk_eval, log_n, 1, st, ctx->pm_b, src);
    src.src = mpb.d;
    if (rc == FHE_OK) rc = fhe::ntt_fwd_multi(ctx->d_desc, 1, pk_eval + n, log_n, 1, st, ctx->pm_b, src);
    if (rc != FHE_OK) return rc;
    hipLaunchKernelGGL(fhe::sample_zo_kernel, dim3(grid_for(blocks_words(count))), dim3(256), 0, st, ma.d, count, 0.5, K, 0ull);
    hipLaunchKernelGGL(fhe::small_i64_to_zq_kernel, dim3(grid_for(count)), dim3(256), 0, st, ma.d, count, (u64)ctx->q);
    HIP_TRY(hipGetLastError());
    fhe::DgTable T;
    rc = dg_table_rc(3.2, 6, &T);
    if (rc == FHE_OK) rc = sample_dg_dev(ctx->q, T, K, blocks_words(count), e, 2 * count, st);
    if (rc == FHE_OK) rc = fhe::ntt_fwd_multi(ctx->d_desc, 1, ma.d, log_n, batch, st, ctx->pm_b);  // u, once; both products read it
    if (rc == FHE_OK && hipMemcpyAsync(mb.d, ma.d, count * sizeof(u64), hipMemcpyDeviceToDevice, st) != hipSuccess) rc = FHE_ERR_HIP;
    for (int half = 0; half < 2 && rc == FHE_OK; ++half) {
        fhe::NttIo mul;
        mul.mul = pk_eval + half * n; mul.mul_div = (unsigned)batch; mul.mul_period = 1;
        rc = fhe::ntt_inv_multi(ctx->d_desc, 1, half ? mb.d : ma.d, log_n, batch, st, ctx->pm_b, mul);
        if (rc != FHE_OK) break;
        hipLaunchKernelGGL(fhe::add3_kernel, dim3(grid_for(count)), dim3(256), 0, st, half ? mb.d : ma.d, (const u64 *)(e + half * count),
                           (half && pt) ? (const u64 *)mpt.d : nullptr, count, count, (u64)ctx->q);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK) rc = ma.sync_out(st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    return rc;
}

// rlwe.rs:172-175 `Rlwe::decrypt`: pt = b - a sk for `batch` ciphertexts; pt may alias ct_b
int fhe_rlwe_decrypt(const fhe_ctx *ctx, const uint64_t *sk, const uint64_t *ct_a, const uint64_t *ct_b, size_t n, size_t batch, uint64_t *pt, fhe_mem mem,
                     void *stream) {
    int rc = check_ring(ctx, n);
    if (rc != FHE_OK) return rc;
    if (!sk || ((!ct_a || !ct_b || !pt) && batch) || (batch && pt == ct_a)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t count = batch * n;
    Mirror msk(sk, n, mem, true, st), ma(ct_a, count, mem, true, st), mb(ct_b, count, mem, true, st), mo(pt, count, mem, false, st);
    if (msk.rc | ma.rc | mb.rc | mo.rc) return FHE_ERR_HIP;
    StreamWs ws((n + count) * sizeof(u64), st);
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *sk_eval = ws.as<u64>(), *as = sk_eval + n;
    const int log_n = ilog2(n);
    fhe::NttIo src;
    src.src = msk.d; src.src_mod = 1;
    rc = fhe::ntt_fwd_multi(ctx->d_desc, 1, sk_eval, log_n, 1, st, ctx->pm_b, src);
    if (rc == FHE_OK) {
        fhe::NttIo sa;
        sa.src = ma.d; sa.src_mod = (unsigned)batch;
        rc = fhe::ntt_fwd_multi(ctx->d_desc, 1, as, log_n, batch, st, ctx->pm_b, sa);
    }
    if (rc == FHE_OK) {
        fhe::NttIo mul;
        mul.mul = sk_eval; mul.mul_div = (unsigned)batch; mul.mul_period = 1;
        rc = fhe::ntt_inv_multi(ctx->d_desc, 1, as, log_n, batch, st, ctx->pm_b, mul);
    }
    if (rc != FHE_OK) return rc;
    hipLaunchKernelGGL(fhe::rsub_kernel, dim3(grid_for(count)), dim3(256), 0, st, as, (const u64 *)mb.d, count, (u64)ctx->q);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(mo.d, as, count * sizeof(u64), hipMemcpyDeviceToDevice, st));
    return mo.sync_out(st);
}

// rgsw.rs:84-105 for `count` plaintext polynomials under one secret key: rows_a / rows_b [count][2d][n]
int fhe_rgsw_encrypt(const fhe_ctx *ctx, int log_b, int d, const uint64_t *sk, const uint64_t *pt, size_t n, size_t count, const fhe_rng *rng, uint64_t stream_id, uint64_t *rows_a, uint64_t *rows_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rc = check_ring(ctx, n), rb = 0;
    if (rc == FHE_OK) rc = gadget_geometry(ctx->q, log_b, d, &rb);
    if (rc != FHE_OK) return rc;
    if (!sk || ((!pt || !rows_a || !rows_b) && count)) return FHE_ERR_INVALID;
    if (count == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t rows = count * 2 * d;
    Mirror msk(sk, n, mem, true, st), mpt(pt, n * count, mem, true, st), ma(rows_a, rows * n, mem, false, st), mb(rows_b, rows * n, mem, false, st);
    if (msk.rc | mpt.rc | ma.rc | mb.rc) return FHE_ERR_HIP;
    StreamWs ws((n + count * d * n) * sizeof(u64), st);  // sk in the evaluation domain | power_up(pt): [count][d][n]
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *sk_eval = ws.as<u64>(), *pw = sk_eval + n;
    fhe::NttIo src;
    src.src = msk.d; src.src_mod = 1;
    rc = fhe::ntt_fwd_multi(ctx->d_desc, 1, sk_eval, ilog2(n), 1, st, ctx->pm_b, src);
    unsigned long long cursor = 0;
    // 2d encryptions of zero per plaintext (rgsw.rs:91-99) ...
    if (rc == FHE_OK) rc = rlwe_sk_encrypt_dev(ctx, sk_eval, nullptr, 0, ma.d, mb.d, n, rows, fhe::call_key(rng, stream_id, fhe::RNG_RGSW_ENC), &cursor, st);
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::power_up_kernel, dim3(grid_for(n * count * d)), dim3(256), 0, st, (const u64 *)mpt.d, pw, n, count, d, rb, log_b,
                           fhe::make_barrett(ctx->q), 0);
        // ... rows 0..d: a += pt base_j; rows d..2d: b += pt base_j (rgsw.rs:100-103)
        for (size_t c = 0; c < count && rc == FHE_OK; ++c) {
            hipLaunchKernelGGL(fhe::add_assign_kernel, dim3(grid_for(d * n)), dim3(256), 0, st, ma.d + c * 2 * d * n, (const u64 *)(pw + c * d * n), (size_t)d * n, (u64)ctx->q);
            hipLaunchKernelGGL(fhe::add_assign_kernel, dim3(grid_for(d * n)), dim3(256), 0, st, mb.d + (c * 2 * d + d) * n, (const u64 *)(pw + c * d * n), (size_t)d * n, (u64)ctx->q);
        }
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK) rc = ma.sync_out(st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    return rc;
}

// rgsw.rs:75-83 `Rgsw::pk_encrypt` (84-105 with `Either::Right(pk)`): 2d public-key encryptions of zero per plaintext, then the gadget
// terms as in fhe_rgsw_encrypt -- what the reference's RGSW tests encrypt with, and how `Bootstrapping::key_share_gen`
// (bootstrapping.rs:277-283) makes a party's blind-rotation key rows under the merged public key.  rows_a / rows_b [count][2d][n]
int fhe_rgsw_pk_encrypt(const fhe_ctx *ctx, int log_b, int d, const uint64_t *pk_a, const uint64_t *pk_b, const uint64_t *pt, size_t n, size_t count,
                        const fhe_rng *rng, uint64_t stream_id, uint64_t *rows_a, uint64_t *rows_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rc = check_ring(ctx, n), rb = 0;
    if (rc == FHE_OK) rc = gadget_geometry(ctx->q, log_b, d, &rb);
    if (rc != FHE_OK) return rc;
    if (!pk_a || !pk_b || ((!pt || !rows_a || !rows_b) && count)) return FHE_ERR_INVALID;
    if (count == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t rows = count * 2 * d;
    Mirror mpa(pk_a, n, mem, true, st), mpb(pk_b, n, mem, true, st), mpt(pt, n * count, mem, true, st), ma(rows_a, rows * n, mem, false, st),
        mb(rows_b, rows * n, mem, false, st);
    if (mpa.rc | mpb.rc | mpt.rc | ma.rc | mb.rc) return FHE_ERR_HIP;
    StreamWs ws(count * d * n * sizeof(u64), st);  // power_up(pt): [count][d][n]
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *pw = ws.as<u64>();
    rc = fhe_rlwe_pk_encrypt(ctx, (const uint64_t *)mpa.d, (const uint64_t *)mpb.d, nullptr, n, rows, rng, stream_id, (uint64_t *)ma.d, (uint64_t *)mb.d, FHE_MEM_DEVICE,
                             stream);
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::power_up_kernel, dim3(grid_for(n * count * d)), dim3(256), 0, st, (const u64 *)mpt.d, pw, n, count, d, rb, log_b,
                           fhe::make_barrett(ctx->q), 0);
        for (size_t c = 0; c < count; ++c) {
            hipLaunchKernelGGL(fhe::add_assign_kernel, dim3(grid_for(d * n)), dim3(256), 0, st, ma.d + c * 2 * d * n, (const u64 *)(pw + c * d * n), (size_t)d * n, (u64)ctx->q);
            hipLaunchKernelGGL(fhe::add_assign_kernel, dim3(grid_for(d * n)), dim3(256), 0, st, mb.d + (c * 2 * d + d) * n, (const u64 *)(pw + c * d * n), (size_t)d * n, (u64)ctx->q);
        }
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK) rc = ma.sync_out(st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    return rc;
}

// rlwe.rs:109-132: key-switching key sk1 -> sk0 (rows encrypt -sk1 base_j under sk0); t != 0: automorphism key, sk1 = sk0(X^t)
int fhe_rlwe_ksk_gen(const fhe_ctx *ctx, int log_b, int d, const uint64_t *sk0, const uint64_t *sk1, int64_t t, size_t n, const fhe_rng *rng, uint64_t stream_id, uint64_t *rows_a, uint64_t *rows_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rc = check_ring(ctx, n), rb = 0;
    if (rc == FHE_OK) rc = gadget_geometry(ctx->q, log_b, d, &rb);
    if (rc != FHE_OK) return rc;
    if (!sk0 || (!sk1 && t == 0) || !rows_a || !rows_b) return FHE_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    Mirror m0(sk0, n, mem, true, st), m1(t == 0 ? sk1 : sk0, n, mem, true, st), ma(rows_a, d * n, mem, false, st), mb(rows_b, d * n, mem, false, st);
    if (m0.rc | m1.rc | ma.rc | mb.rc) return FHE_ERR_HIP;
    StreamWs ws((2 * n + d * n) * sizeof(u64), st);  // sk0 evaluation domain | sk1 (or its automorphism) | power_up(-sk1)
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *sk_eval = ws.as<u64>(), *s1 = sk_eval + n, *pw = s1 + n;
    fhe::NttIo src;
    src.src = m0.d; src.src_mod = 1;
    rc = fhe::ntt_fwd_multi(ctx->d_desc, 1, sk_eval, ilog2(n), 1, st, ctx->pm_b, src);
    if (rc == FHE_OK && t != 0) rc = fhe_automorphism(ctx->q, t, (const uint64_t *)m0.d, (uint64_t *)s1, n, 1, FHE_MEM_DEVICE, stream);  // rlwe.rs:129
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::power_up_kernel, dim3(grid_for(n * d)), dim3(256), 0, st, (const u64 *)(t != 0 ? s1 : m1.d), pw, n, (size_t)1, d, rb, log_b,
                           fhe::make_barrett(ctx->q), 1);  // power_up(-sk1)
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    unsigned long long cursor = 0;
    if (rc == FHE_OK) rc = rlwe_sk_encrypt_dev(ctx, sk_eval, pw, d, ma.d, mb.d, n, d, fhe::call_key(rng, stream_id, fhe::RNG_RLWE_KSK), &cursor, st);
    if (rc == FHE_OK) rc = ma.sync_out(st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    return rc;
}

// scheme/fhew/src/lwe.rs:128-139 for `rows` plaintexts (pt [rows] or NULL = zeros): out_a [rows][n] uniform, out_b [rows]
// a_given: the masks come from the caller (the multi-party shares of lwe.rs:169-226: a common reference string, or a ciphertext's
// mask) and out_a is not written
static int lwe_encrypt_common(uint64_t q, const u64 *sk, const u64 *pt, const u64 *sk1, size_t n1, int rb, int log_b, size_t n, size_t rows,
                              const fhe::ChaChaKey &K, u64 *out_a, u64 *out_b, hipStream_t st, const u64 *a_given = nullptr) {
    fhe::DgTable T;
    int rc = dg_table_rc(3.2, 6, &T);
    if (rc != FHE_OK) return rc;
    StreamWs we(rows * sizeof(u64), st);
    if (we.rc != FHE_OK) return we.rc;
    if (!a_given) rc = sample_uniform_dev(q, K, 0, out_a, rows * n, st);
    if (rc == FHE_OK) rc = sample_dg_dev(q, T, K, blocks_uniform(rows * n), we.as<u64>(), rows, st);
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::lwe_encrypt_kernel, dim3(grid_for(rows)), dim3(256), 0, st, a_given ? a_given : (const u64 *)out_a, sk, (const u64 *)we.as<u64>(), pt,
                           out_b, n, rows, fhe::make_barrett(q), sk1, n1 ? n1 : 1, rb, log_b);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    return rc;
}

int fhe_lwe_sk_encrypt(uint64_t q, const uint64_t *sk, const uint64_t *pt, size_t n, size_t rows, const fhe_rng *rng, uint64_t stream_id,
                       uint64_t *out_a, uint64_t *out_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    if (q < 2 || (q >> 62) || !sk || n == 0 || ((!out_a || !out_b) && rows)) return FHE_ERR_INVALID;
    if (rows == 0) return FHE_OK;
    PtrDeviceGuard pguard(out_a, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    Mirror msk(sk, n, mem, true, st), mpt(pt, pt ? rows : 0, mem, true, st), ma(out_a, rows * n, mem, false, st), mb(out_b, rows, mem, false, st);
    if (msk.rc | mpt.rc | ma.rc | mb.rc) return FHE_ERR_HIP;
    int rc = lwe_encrypt_common(q, msk.d, pt ? mpt.d : nullptr, nullptr, 0, 0, 0, n, rows, fhe::call_key(rng, stream_id, fhe::RNG_LWE_ENC), ma.d, mb.d, st);
    if (rc == FHE_OK) rc = ma.sync_out(st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    return rc;
}

// scheme/fhew/src/lwe.rs:108-119 `Lwe::ksk_gen(param, sk0, sk1)`: rows r = j n1 + i encrypt -sk1[i] base_j under sk0:
// ksk_a [n1 d][n0], ksk_b [n1 d], the layout fhe_lwe_key_switch takes (n_in = n1, n_out = n0)
int fhe_lwe_ksk_gen(uint64_t q, int log_b, int d, const uint64_t *sk0, size_t n0, const uint64_t *sk1, size_t n1, const fhe_rng *rng, uint64_t stream_id, uint64_t *ksk_a, uint64_t *ksk_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rb = 0;
    int rc = gadget_geometry(q, log_b, d, &rb);
    if (rc != FHE_OK) return rc;
    if ((q >> 62) || !sk0 || !sk1 || n0 == 0 || n1 == 0 || !ksk_a || !ksk_b) return FHE_ERR_INVALID;
    PtrDeviceGuard pguard(ksk_a, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = n1 * d;
    Mirror m0(sk0, n0, mem, true, st), m1(sk1, n1, mem, true, st), ma(ksk_a, rows * n0, mem, false, st), mb(ksk_b, rows, mem, false, st);
    if (m0.rc | m1.rc | ma.rc | mb.rc) return FHE_ERR_HIP;
    rc = lwe_encrypt_common(q, m0.d, nullptr, m1.d, n1, rb, log_b, n0, rows, fhe::call_key(rng, stream_id, fhe::RNG_LWE_KSK), ma.d, mb.d, st);
    if (rc == FHE_OK) rc = ma.sync_out(st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    return rc;
}

// scheme/fhew/src/lwe.rs:169-183 `Lwe::sk_share_encrypt(param, a, sk, pt)` and 197-207 `share_decrypt` (pt NULL): b[r] = <a[r], sk> + pt[r] +
// e[r] for GIVEN masks a [rows][n]; the merges (185-195, 209-212) are sums: fhe_rq_sum / fhe_rq_sub with len = 1 ... or on the host
int fhe_lwe_share_encrypt(uint64_t q, const uint64_t *a, const uint64_t *sk, const uint64_t *pt, size_t n, size_t rows, const fhe_rng *rng, uint64_t stream_id,
                          uint64_t *out_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    if (q < 2 || (q >> 62) || !sk || !a || n == 0 || (!out_b && rows)) return FHE_ERR_INVALID;
    if (rows == 0) return FHE_OK;
    PtrDeviceGuard pguard(out_b, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    Mirror msk(sk, n, mem, true, st), ma(a, rows * n, mem, true, st), mpt(pt, pt ? rows : 0, mem, true, st), mb(out_b, rows, mem, false, st);
    if (msk.rc | ma.rc | mpt.rc | mb.rc) return FHE_ERR_HIP;
    int rc = lwe_encrypt_common(q, msk.d, pt ? mpt.d : nullptr, nullptr, 0, 0, 0, n, rows, fhe::call_key(rng, stream_id, fhe::RNG_LWE_SHARE), nullptr, mb.d, st, ma.d);
    return rc == FHE_OK ? mb.sync_out(st) : rc;
}

// lwe.rs:214-226 `Lwe::ksk_share_gen(param, crs, sk0, sk1)`: row r = j n1 + i: b = <crs[r], sk0> - sk1[i] base_j + e.  crs [n1 d][n0] (the
// common reference string: the ksk_a every party and the merged key share), out_b [n1 d]; the merged key is (crs, sum of the shares)
int fhe_lwe_ksk_share_gen(uint64_t q, int log_b, int d, const uint64_t *crs, const uint64_t *sk0, size_t n0, const uint64_t *sk1, size_t n1, const fhe_rng *rng,
                          uint64_t stream_id, uint64_t *out_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rb = 0;
    int rc = gadget_geometry(q, log_b, d, &rb);
    if (rc != FHE_OK) return rc;
    if ((q >> 62) || !crs || !sk0 || !sk1 || n0 == 0 || n1 == 0 || !out_b) return FHE_ERR_INVALID;
    PtrDeviceGuard pguard(out_b, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = n1 * d;
    Mirror m0(sk0, n0, mem, true, st), m1(sk1, n1, mem, true, st), ma(crs, rows * n0, mem, true, st), mb(out_b, rows, mem, false, st);
    if (m0.rc | m1.rc | ma.rc | mb.rc) return FHE_ERR_HIP;
    rc = lwe_encrypt_common(q, m0.d, nullptr, m1.d, n1, rb, log_b, n0, rows, fhe::call_key(rng, stream_id, fhe::RNG_LWE_KSK_SHARE), nullptr, mb.d, st, ma.d);
    return rc == FHE_OK ? mb.sync_out(st) : rc;
}

// util/src/ring.rs:328-341 `Rq: Sum` over `count` polynomials of `len` coefficients: in [count][len] -> out [len]
int fhe_rq_sum(uint64_t q, const uint64_t *in, size_t len, size_t count, uint64_t *out, fhe_mem mem, void *stream) {
    if (q < 2 || (q >> 62) || ((!in || !out) && len)) return FHE_ERR_INVALID;
    if (len == 0) return FHE_OK;
    PtrDeviceGuard pguard(in, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    Mirror mi(in, len * count, mem, true, st), mo(out, len, mem, false, st);
    if (mi.rc | mo.rc) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::rq_sum_kernel, dim3(grid_for(len)), dim3(256), 0, st, (const u64 *)mi.d, mo.d, len, count, (u64)q);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

}  // extern "C"
