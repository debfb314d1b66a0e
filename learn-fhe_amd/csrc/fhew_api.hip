// extern "C" entry points for SURVEY.md section 8(a) rows a8-a13: decomposition, automorphism, monomial
// multiply, prepared gadget keys, external product, RLWE key switch, LMKCDEY blind rotation.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <new>
#include <vector>

#include "api_common.hpp"
#include "ctx.hpp"
#include "fhew_kernels.hpp"
#include "lwe_kernels.hpp"

// defined in ring_api.hip
namespace fhe {
int ntt_fwd_device(const fhe_ctx *c, u64 *a, int log_n, size_t batch, hipStream_t st);
}

struct fhe_key {
    const fhe_ctx *ctx = nullptr;
    int log_n = 0;
    int log_b = 0, d = 0;
    int rows_per_ct = 0;  // 2d (RGSW) or d (key-switching key)
    size_t count = 0;
    u64 *d_rows = nullptr;  // [count][rows_per_ct][2][N], evaluation domain, key_perm layout
    u64 *d_rows_small = nullptr;  // N >= 1024: the same rows in the layout of the small-batch kernels (4 coefficients per lane)
    fhe::DecompParams P{};
};

struct fhe_bootstrap_key {
    const fhe_ctx *ctx = nullptr;
    const fhe_key *brk = nullptr, *ak = nullptr;
    int w = 0;
    unsigned *d_ak_t = nullptr;  // [w + 1] exponents mod 2N
    unsigned *d_dlog = nullptr;  // [2N]
    int *d_status = nullptr;     // sticky device-side error word of asynchronous (device-memory) blind rotations: fhe_bootstrap_key_status
};

namespace {

int make_decomp(uint64_t q, int log_b, int d, fhe::DecompParams *P) {
    if (q < 2 || log_b < 1 || log_b > 62 || d < 1 || d > 64) return FHE_ERR_INVALID;
    if ((q >> 62) || (1ull << log_b) >= q) return FHE_ERR_UNSUPPORTED;
    const int log_q = q <= 1 ? 0 : 64 - __builtin_clzll(q - 1);  // q.next_power_of_two().ilog2()
    const int rb = log_q - log_b * d > 0 ? log_q - log_b * d : 0;
    P->q = q;
    P->rnd = ((1ull << rb) >> 1) % q;
    P->neg_b = q - (1ull << log_b);
    P->mask = (1ull << log_b) - 1;
    P->b_by_2 = 1ull << (log_b - 1);
    P->log_b = log_b;
    P->d = d;
    P->rb = rb;
    return FHE_OK;
}

inline unsigned grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b ? b : 1));
}

fhe::RingConsts ring_consts(const fhe_ctx *c, int) {
    fhe::RingConsts K;
    K.desc = c->d_desc;
    K.B = c->barrett;
    return K;
}

// fused kernels are instantiated for Shoup arithmetic (any prime) at every supported degree and for the 54-bit
// pseudo-Mersenne primes (BASELINE config 3's modulus) at N = 512 .. 2048
// arithmetic of the fused kernels on cfg3's 54-bit moduli: -DFHE_FHEW_DS=0 keeps the single-operand product (A/B switch)
#ifndef FHE_FHEW_DS
#define FHE_FHEW_DS 1
#endif
#if FHE_FHEW_DS
#define FHEW_POLICY54 fhe::ArithDS<54>
#else
#define FHEW_POLICY54 fhe::ArithPM<54>
#endif
// ... and on the 55-bit ones of the reference's own parameter sets (scheme/fhew/examples/multi_key_uint8.rs:15-29: log_q = 55)
#define FHEW_POLICY55 fhe::ArithDS<55>
inline int fhew_pm(const fhe_ctx *c, int log_n) { return ((c->pm_b == 54 || c->pm_b == 55) && log_n >= 9) ? c->pm_b : 0; }

#define FHEW_DISPATCH(log_n, ...)                                          \
    switch (log_n) {                                                       \
        case 7: { constexpr int LN = 7; __VA_ARGS__; break; }              \
        case 8: { constexpr int LN = 8; __VA_ARGS__; break; }              \
        case 9: { constexpr int LN = 9; __VA_ARGS__; break; }              \
        case 10: { constexpr int LN = 10; __VA_ARGS__; break; }            \
        case 11: { constexpr int LN = 11; __VA_ARGS__; break; }            \
        default: return FHE_ERR_UNSUPPORTED;                               \
    }

template <typename K>
int set_lds(K kern, size_t bytes) {
    if (bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return FHE_OK;
}

int key_prepare(const fhe_ctx *ctx, int log_b, int d, int rows_per_ct, const uint64_t *rows_a, const uint64_t *rows_b, size_t n,
                size_t count, fhe_mem mem, fhe_key **out) {
    if (!out) return FHE_ERR_INVALID;
    *out = nullptr;
    if (!ctx || !rows_a || !rows_b || !is_pow2(n) || count == 0) return FHE_ERR_INVALID;
    if (ctx->device < 0) return FHE_ERR_NO_DEVICE;
    const int log_n = ilog2(n);
    if (log_n > ctx->s - 1) return FHE_ERR_NO_ROOT;
    if (log_n < 7 || log_n > 11) return FHE_ERR_UNSUPPORTED;
    fhe::DecompParams P;
    int rc = make_decomp(ctx->q, log_b, d, &P);
    if (rc != FHE_OK) return rc;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t rows = count * rows_per_ct, words = rows * n;
    hipStream_t st = nullptr;
    u64 *ta = nullptr, *tb = nullptr, *dst = nullptr;
    HIP_TRY(hipMalloc((void **)&ta, 2 * words * sizeof(u64)));
    tb = ta + words;
    hipError_t e = hipMalloc((void **)&dst, 2 * words * sizeof(u64));
    hipMemcpyKind kind = mem == FHE_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (e == hipSuccess) e = hipMemcpyAsync(ta, rows_a, words * sizeof(u64), kind, st);
    if (e == hipSuccess) e = hipMemcpyAsync(tb, rows_b, words * sizeof(u64), kind, st);
    rc = e == hipSuccess ? FHE_OK : FHE_ERR_HIP;
    if (e != hipSuccess) g_last_hip = (int)e;
    // rows -> evaluation domain once (what Rgsw::internal_product does per call, rgsw.rs:136-138)
    if (rc == FHE_OK) rc = fhe::ntt_fwd_device(ctx, ta, log_n, 2 * rows, st);
    if (rc == FHE_OK) {
        FHEW_DISPATCH(log_n, hipLaunchKernelGGL(fhe::key_permute_kernel<fhe::WaveRing<LN>>, dim3(grid_for(words)), dim3(256), 0, st, ta, tb, dst, rows,
                                                 fhew_pm(ctx, log_n)));
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    u64 *dst_small = nullptr;
    if (rc == FHE_OK && log_n >= 10) {
        if (hipMalloc((void **)&dst_small, 2 * words * sizeof(u64)) != hipSuccess) rc = FHE_ERR_HIP;
        if (rc == FHE_OK) {
            const int pm = fhew_pm(ctx, log_n);
            if (log_n == 10) hipLaunchKernelGGL((fhe::key_permute_kernel<fhe::WaveRing<10, 2>>), dim3(grid_for(words)), dim3(256), 0, st, ta, tb, dst_small, rows, pm);
            else hipLaunchKernelGGL((fhe::key_permute_kernel<fhe::WaveRing<11, 2>>), dim3(grid_for(words)), dim3(256), 0, st, ta, tb, dst_small, rows, pm);
            if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == FHE_OK) rc = FHE_ERR_HIP;
    (void)hipFree(ta);
    if (rc != FHE_OK) { if (dst) (void)hipFree(dst); if (dst_small) (void)hipFree(dst_small); return rc; }
    fhe_key *k = new (std::nothrow) fhe_key();
    if (!k) { (void)hipFree(dst); if (dst_small) (void)hipFree(dst_small); return FHE_ERR_INVALID; }
    k->ctx = ctx; k->log_n = log_n; k->log_b = log_b; k->d = d; k->rows_per_ct = rows_per_ct; k->count = count;
    k->d_rows = dst; k->d_rows_small = dst_small; k->P = P;
    *out = k;
    return FHE_OK;
}

// Which instantiation a batch runs at N >= 1024: 8 coefficients per lane (2 / 4 waves per ciphertext, 250 registers: 2 waves per SIMD,
// FOUR ciphertexts per CU) or 4 per lane (4 / 8 waves per ciphertext, 148 registers: 3 waves per SIMD, THREE ciphertexts per CU, but 12
// resident waves instead of 8).  Measured at cfg3 (tools/fhew_shape_lab.py, blind rotations/s, 8-per-lane | 4-per-lane):
//   batch 1: 77 | 120     64: 4.8 k | 7.6 k     768: 53.0 k | 64.3 k     1024: 66.3 k | 59.0 k     1536: 51.2 k | 68.1 k
//   2048: 67.7 k | 69.5 k     3072: 67.7 k | 72.2 k     4096: 68.2 k | 73.0 k
// The 4-per-lane form is the faster kernel per ciphertext (more waves to hide the key-row and LDS latencies behind); the
// 8-per-lane form wins only where its larger generation (1024 ciphertexts on 256 CUs against 768) saves a whole pass: batches of
// 769 .. 1024.  (N = 2048 keeps the rule it was measured with: 4 per lane up to 512.)
// In compute units of the device the call runs on (256 on MI355X: the figures above): a generation of the 4-per-lane form is 3
// ciphertexts per CU, of the 8-per-lane form 4.
inline bool small_shape(int log_n, size_t batch) {
    const long lab = fhe::opt(fhe::OPT_SMALL_BATCH);  // lab override (api_common.hpp)
    if (lab >= 0) return log_n >= 10 && batch <= (size_t)lab;
    const size_t cus = (size_t)fhe::current_cu_count();
    if (log_n == 10) return batch <= 3 * cus || batch > 4 * cus;
    return log_n >= 10 && batch <= 2 * cus;
}

fhe::FhewKey key_view(const fhe_key *k, bool small = false) {
    fhe::FhewKey v;
    v.rows = small ? k->d_rows_small : k->d_rows;
    v.rows_per_ct = k->rows_per_ct;
    v.P = k->P;
    return v;
}

int check_ct_call(const fhe_ctx *ctx, const fhe_key *key, size_t index, const void *a, const void *b, size_t batch) {
    if (!ctx || !key || ((!a || !b) && batch)) return FHE_ERR_INVALID;
    if (key->ctx != ctx) return FHE_ERR_MODULUS;
    if (index >= key->count) return FHE_ERR_INVALID;
    if (ctx->device < 0) return FHE_ERR_NO_DEVICE;
    if (batch > 0xffffffffull) return FHE_ERR_UNSUPPORTED;
    return FHE_OK;
}

}  // namespace

extern "C" {

int fhe_decompose(uint64_t q, int log_b, int d, const uint64_t *in, size_t n, size_t polys, uint64_t *out, fhe_mem mem,
                  void *stream) {
    PtrDeviceGuard pguard(in, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    fhe::DecompParams P;
    int rc = make_decomp(q, log_b, d, &P);
    if (rc != FHE_OK) return rc;
    if ((!in || !out) && n * polys) return FHE_ERR_INVALID;
    if (n * polys == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    Mirror mi(in, n * polys, mem, true, st), mo(out, n * polys * d, mem, false, st);
    if (mi.rc != FHE_OK || mo.rc != FHE_OK) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::decompose_kernel, dim3(grid_for(n * polys)), dim3(256), 0, st, mi.d, mo.d, n, polys, P);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

int fhe_automorphism(uint64_t q, int64_t t, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem,
                     void *stream) {
    PtrDeviceGuard pguard(in, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    if (q < 2 || !is_pow2(n) || n > (1u << 30) || ((!in || !out) && batch) || in == out) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    const int64_t two_n = 2 * (int64_t)n;
    const unsigned tt = (unsigned)(((t % two_n) + two_n) % two_n);  // t.rem_euclid(2n), avec.rs:38
    hipStream_t st = (hipStream_t)stream;
    Mirror mi(in, n * batch, mem, true, st), mo(out, n * batch, mem, true, st);  // copy_in: untouched slots keep `in` values
    if (mi.rc != FHE_OK || mo.rc != FHE_OK) return FHE_ERR_HIP;
    // `let mut v = self.clone()` (avec.rs:36): positions no X^(i t) lands on keep the input value (even t)
    HIP_TRY(hipMemcpyAsync(mo.d, mi.d, n * batch * sizeof(u64), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(fhe::automorphism_kernel, dim3(grid_for(n * batch)), dim3(256), 0, st, mi.d, mo.d, (unsigned)n, batch, tt, (u64)q);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

int fhe_monomial_mul(uint64_t q, int64_t k, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem,
                     void *stream) {
    PtrDeviceGuard pguard(in, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    if (q < 2 || !is_pow2(n) || n > (1u << 30) || ((!in || !out) && batch) || in == out) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    const int64_t two_n = 2 * (int64_t)n;
    const unsigned kk = (unsigned)(((k % two_n) + two_n) % two_n);  // rem_euclid(2n), ring.rs:305
    hipStream_t st = (hipStream_t)stream;
    Mirror mi(in, n * batch, mem, true, st), mo(out, n * batch, mem, false, st);
    if (mi.rc != FHE_OK || mo.rc != FHE_OK) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::monomial_mul_kernel, dim3(grid_for(n * batch)), dim3(256), 0, st, mi.d, mo.d, (unsigned)n, batch, kk, (u64)q);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

int fhe_rgsw_prepare(const fhe_ctx *ctx, int log_b, int d, const uint64_t *rows_a, const uint64_t *rows_b, size_t n,
                     size_t count, fhe_mem mem, fhe_key **out) {
    return key_prepare(ctx, log_b, d, 2 * d, rows_a, rows_b, n, count, mem, out);
}

int fhe_ksk_prepare(const fhe_ctx *ctx, int log_b, int d, const uint64_t *rows_a, const uint64_t *rows_b, size_t n,
                    size_t count, fhe_mem mem, fhe_key **out) {
    return key_prepare(ctx, log_b, d, d, rows_a, rows_b, n, count, mem, out);
}

void fhe_key_destroy(fhe_key *k) {
    if (!k) return;
    if (k->ctx && k->ctx->device >= 0) {
        DeviceGuard guard(k->ctx->device);
        if (k->d_rows) (void)hipFree(k->d_rows);
        if (k->d_rows_small) (void)hipFree(k->d_rows_small);
    }
    delete k;
}

static int gadget_entry(const fhe_ctx *ctx, const fhe_key *key, size_t index, bool both, bool is_auto, int64_t t, uint64_t *ct_a,
                        uint64_t *ct_b, size_t batch, fhe_mem mem, void *stream) {
    int rc = check_ct_call(ctx, key, index, ct_a, ct_b, batch);
    if (rc != FHE_OK) return rc;
    if (key->rows_per_ct != (both ? 2 : 1) * key->d) return FHE_ERR_INVALID;  // RGSW key vs key-switching key
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t n = size_t(1) << key->log_n;
    unsigned tt = 1;
    if (is_auto) {
        const int64_t two_n = 2 * (int64_t)n;
        tt = (unsigned)(((t % two_n) + two_n) % two_n);
        if ((tt & 1) == 0) return FHE_ERR_INVALID;  // only odd t are ring automorphisms
    }
    Mirror ma(ct_a, n * batch, mem, true, st), mb(ct_b, n * batch, mem, true, st);
    if (ma.rc != FHE_OK || mb.rc != FHE_OK) return FHE_ERR_HIP;
    const bool small = small_shape(key->log_n, batch);
#define GP_LAUNCH_W(AR, WR)                                                                                                  \
    {                                                                                                                        \
        const size_t lds = WR::LDS_BYTES;                                                                                    \
        rc = set_lds(fhe::gadget_product_kernel<AR, WR>, lds);                                                               \
        if (rc != FHE_OK) return rc;                                                                                         \
        hipLaunchKernelGGL((fhe::gadget_product_kernel<AR, WR>), dim3((unsigned)((batch + WR::TEAMS - 1) / WR::TEAMS)),      \
                           dim3(WR::THREADS), lds, st, ma.d, mb.d, (unsigned)batch, key_view(key, small), (unsigned)index,   \
                           both ? 1u : 0u, tt, ring_consts(ctx, 0));                                                         \
    }
#define GP_LAUNCH(AR, LN) GP_LAUNCH_W(AR, fhe::WaveRing<LN>)
#define GP_LAUNCH_BIG(AR, LN)                                                                \
    {                                                                                        \
        if (small) { typedef fhe::WaveRing<LN, 2> WS; GP_LAUNCH_W(AR, WS) }                  \
        else { typedef fhe::WaveRing<LN> WD; GP_LAUNCH_W(AR, WD) }                           \
    }
    const int pmv = fhew_pm(ctx, key->log_n);
    if (pmv == 54) {
        switch (key->log_n) {
            case 9: GP_LAUNCH(FHEW_POLICY54, 9) break;
            case 10: GP_LAUNCH_BIG(FHEW_POLICY54, 10) break;
            case 11: GP_LAUNCH_BIG(FHEW_POLICY54, 11) break;
            default: return FHE_ERR_UNSUPPORTED;
        }
    } else if (pmv == 55) {
        switch (key->log_n) {
            case 9: GP_LAUNCH(FHEW_POLICY55, 9) break;
            case 10: GP_LAUNCH_BIG(FHEW_POLICY55, 10) break;
            case 11: GP_LAUNCH_BIG(FHEW_POLICY55, 11) break;
            default: return FHE_ERR_UNSUPPORTED;
        }
    } else {
        switch (key->log_n) {
            case 7: GP_LAUNCH(fhe::ArithShoup, 7) break;
            case 8: GP_LAUNCH(fhe::ArithShoup, 8) break;
            case 9: GP_LAUNCH(fhe::ArithShoup, 9) break;
            case 10: GP_LAUNCH_BIG(fhe::ArithShoup, 10) break;
            case 11: GP_LAUNCH_BIG(fhe::ArithShoup, 11) break;
            default: return FHE_ERR_UNSUPPORTED;
        }
    }
#undef GP_LAUNCH_BIG
#undef GP_LAUNCH_W
#undef GP_LAUNCH
    HIP_TRY(hipGetLastError());
    rc = ma.sync_out(st);
    return rc != FHE_OK ? rc : mb.sync_out(st);
}

int fhe_external_product(const fhe_ctx *ctx, const fhe_key *rgsw, size_t index, uint64_t *ct_a, uint64_t *ct_b, size_t batch,
                         fhe_mem mem, void *stream) {
    return gadget_entry(ctx, rgsw, index, true, false, 1, ct_a, ct_b, batch, mem, stream);
}

// scheme/fhew/src/rgsw.rs:130-150: one external product of rgsw[index] per RLWE row of each right-hand RGSW ciphertext
int fhe_rgsw_internal_product(const fhe_ctx *ctx, const fhe_key *rgsw, size_t index, uint64_t *ct1_a, uint64_t *ct1_b, size_t count,
                              fhe_mem mem, void *stream) {
    if (!rgsw || rgsw->rows_per_ct != 2 * rgsw->d) return FHE_ERR_INVALID;
    return gadget_entry(ctx, rgsw, index, true, false, 1, ct1_a, ct1_b, count * 2 * (size_t)rgsw->d, mem, stream);
}

int fhe_rlwe_key_switch(const fhe_ctx *ctx, const fhe_key *ksk, size_t index, uint64_t *ct_a, uint64_t *ct_b, size_t batch,
                        fhe_mem mem, void *stream) {
    return gadget_entry(ctx, ksk, index, false, false, 1, ct_a, ct_b, batch, mem, stream);
}

int fhe_rlwe_automorphism(const fhe_ctx *ctx, const fhe_key *ak, size_t index, int64_t t, uint64_t *ct_a, uint64_t *ct_b,
                          size_t batch, fhe_mem mem, void *stream) {
    return gadget_entry(ctx, ak, index, false, true, t, ct_a, ct_b, batch, mem, stream);
}

// util/src/zq.rs:128-140 via scheme/fhew/src/lwe.rs:90-99: v -> round(v * q_prime / q) (odd != 0: `mod_switch_odd`)
int fhe_lwe_mod_switch(uint64_t q, uint64_t q_prime, const uint64_t *in, uint64_t *out, size_t count, int odd, fhe_mem mem, void *stream) {
    PtrDeviceGuard pguard(in, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    if (q < 2 || q_prime < 2 || ((!in || !out) && count)) return FHE_ERR_INVALID;
    if (count == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    Mirror mi(in, count, mem, true, st), mo(out, count, mem, false, st);
    if (mi.rc | mo.rc) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::lwe_mod_switch_kernel, dim3(grid_for(count)), dim3(256), 0, st, (const u64 *)mi.d, mo.d, count, (u64)q, (u64)q_prime, odd);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

// scheme/fhew/src/lwe.rs:151-160 `Lwe::key_switch` over modulus q < 2^32
int fhe_lwe_key_switch(uint64_t q, int log_b, int d, const uint64_t *ksk_a, const uint64_t *ksk_b, const uint64_t *ct_a,
                       const uint64_t *ct_b, size_t n_in, size_t n_out, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem,
                       void *stream) {
    PtrDeviceGuard pguard(ct_a, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    fhe::DecompParams P;
    int rc = make_decomp(q, log_b, d, &P);
    if (rc != FHE_OK) return rc;
    if (q >> 32) return FHE_ERR_UNSUPPORTED;
    if (!ksk_a || !ksk_b || n_in == 0 || n_out == 0 || ((!ct_a || !ct_b || !out_a || !out_b) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = n_in * d;
    Mirror mka(ksk_a, rows * n_out, mem, true, st), mkb(ksk_b, rows, mem, true, st), ma(ct_a, n_in * batch, mem, true, st),
        mb(ct_b, batch, mem, true, st), moa(out_a, n_out * batch, mem, false, st), mob(out_b, batch, mem, false, st);
    if (mka.rc | mkb.rc | ma.rc | mb.rc | moa.rc | mob.rc) return FHE_ERR_HIP;
    const int tile = fhe::ks_tile(batch, rows);
    if (tile && batch < (size_t(1) << 31)) {  // tiled kernel (lwe_kernels.hpp); sums wrap and are masked once when q is a power of two
        const bool pow2 = (q & (q - 1)) == 0;
        const int bad = pow2 ? fhe::launch_key_switch_tiled(fhe::KsZqPow2{P}, ma.d, mb.d, n_in, n_out, batch, mka.d, mkb.d, moa.d, mob.d, tile, st)
                             : fhe::launch_key_switch_tiled(fhe::KsZq{P}, ma.d, mb.d, n_in, n_out, batch, mka.d, mkb.d, moa.d, mob.d, tile, st);
        if (bad) return FHE_ERR_HIP;
    } else {
        hipLaunchKernelGGL(fhe::lwe_key_switch_kernel, dim3(grid_for((n_out + 1) * batch)), dim3(256), 0, st, (const u64 *)ma.d,
                           (const u64 *)mb.d, (unsigned)n_in, (unsigned)n_out, batch, (const u64 *)mka.d, (const u64 *)mkb.d, P, moa.d, mob.d);
    }
    HIP_TRY(hipGetLastError());
    rc = moa.sync_out(st);
    return rc != FHE_OK ? rc : mob.sync_out(st);
}

// scheme/fhew/src/lwe.rs:22-75: out = sum_k coef[k] * in[k] + addend over q (the gates' linear parts, fhew.rs:27-29, 61-69)
int fhe_lwe_lincomb(uint64_t q, int k, const int64_t *coef, const uint64_t *const *in, uint64_t addend, uint64_t *out, size_t count,
                    fhe_mem mem, void *stream) {
    PtrDeviceGuard pguard(out, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    if (q < 2 || (q >> 62) || k < 1 || k > 4 || !coef || !in || addend >= q || (!out && count)) return FHE_ERR_INVALID;
    for (int t = 0; t < k; ++t)
        if (!in[t] && count) return FHE_ERR_INVALID;
    if (count == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    Mirror m0(in[0], count, mem, true, st), m1(k > 1 ? in[1] : nullptr, k > 1 ? count : 0, mem, true, st),
        m2(k > 2 ? in[2] : nullptr, k > 2 ? count : 0, mem, true, st), m3(k > 3 ? in[3] : nullptr, k > 3 ? count : 0, mem, true, st),
        mo(out, count, mem, false, st);
    if (m0.rc | m1.rc | m2.rc | m3.rc | mo.rc) return FHE_ERR_HIP;
    fhe::LinComb lc{};
    lc.k = k;
    const u64 *ptrs[4] = {m0.d, m1.d, m2.d, m3.d};
    for (int t = 0; t < k; ++t) { lc.in[t] = ptrs[t]; lc.coef[t] = coef[t]; }
    hipLaunchKernelGGL(fhe::lwe_lincomb_kernel, dim3(grid_for(count)), dim3(256), 0, st, lc, (u64)q, (u64)addend, mo.d, count);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

// scheme/fhew/src/rlwe.rs:193-202 `Rlwe::sample_extract(ct, index)`, b += addend (mod q)
int fhe_rlwe_sample_extract(uint64_t q, const uint64_t *ct_a, const uint64_t *ct_b, size_t n, size_t index, uint64_t addend,
                            uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream) {
    PtrDeviceGuard pguard(ct_a, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    if (q < 2 || (q >> 62) || addend >= q || !is_pow2(n) || index >= n || n > (1u << 30) || ((!ct_a || !ct_b || !out_a || !out_b) && batch))
        return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    Mirror ma(ct_a, n * batch, mem, true, st), mb(ct_b, n * batch, mem, true, st), moa(out_a, n * batch, mem, false, st),
        mob(out_b, batch, mem, false, st);
    if (ma.rc | mb.rc | moa.rc | mob.rc) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::rlwe_sample_extract_kernel, dim3(grid_for(n * batch)), dim3(256), 0, st, (const u64 *)ma.d, (const u64 *)mb.d,
                       (unsigned)n, batch, (unsigned)index, (u64)q, (u64)addend, moa.d, mob.d);
    HIP_TRY(hipGetLastError());
    int rc = moa.sync_out(st);
    return rc != FHE_OK ? rc : mob.sync_out(st);
}

int fhe_bootstrap_key_create(const fhe_ctx *ctx, const fhe_key *brk, const fhe_key *ak, const int64_t *ak_t, int w,
                             fhe_bootstrap_key **out) {
    if (!out) return FHE_ERR_INVALID;
    *out = nullptr;
    if (!ctx || !brk || !ak || !ak_t || w < 1) return FHE_ERR_INVALID;
    if (brk->ctx != ctx || ak->ctx != ctx) return FHE_ERR_MODULUS;
    if (brk->log_n != ak->log_n || ak->count != (size_t)w + 1 || brk->rows_per_ct != 2 * brk->d || ak->rows_per_ct != ak->d)
        return FHE_ERR_INVALID;
    if (ctx->device < 0) return FHE_ERR_NO_DEVICE;
    const unsigned n = 1u << brk->log_n, q2 = 2 * n;
    std::vector<unsigned> t(w + 1), dlog(q2, 0xffffffffu);
    for (int i = 0; i <= w; ++i) {
        int64_t v = ((ak_t[i] % (int64_t)q2) + q2) % q2;
        if ((v & 1) == 0) return FHE_ERR_INVALID;
        t[i] = (unsigned)v;
    }
    unsigned x = 1;  // log_g_map (bootstrapping.rs:228-231): +-5^l -> l, l < n/2
    for (unsigned l = 0; l < n / 2; ++l) {
        dlog[x] = (l << 1) | 0u;
        dlog[(q2 - x) % q2] = (l << 1) | 1u;
        x = (unsigned)((uint64_t)x * 5u % q2);
    }
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    fhe_bootstrap_key *bk = new (std::nothrow) fhe_bootstrap_key();
    if (!bk) return FHE_ERR_INVALID;
    bk->ctx = ctx; bk->brk = brk; bk->ak = ak; bk->w = w;
    hipError_t e = hipMalloc((void **)&bk->d_ak_t, t.size() * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc((void **)&bk->d_dlog, dlog.size() * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemcpy(bk->d_ak_t, t.data(), t.size() * sizeof(unsigned), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(bk->d_dlog, dlog.data(), dlog.size() * sizeof(unsigned), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&bk->d_status, sizeof(int));
    if (e == hipSuccess) e = hipMemset(bk->d_status, 0, sizeof(int));
    if (e != hipSuccess) {
        g_last_hip = (int)e;
        if (bk->d_ak_t) (void)hipFree(bk->d_ak_t);
        if (bk->d_dlog) (void)hipFree(bk->d_dlog);
        if (bk->d_status) (void)hipFree(bk->d_status);
        delete bk;
        return FHE_ERR_HIP;
    }
    *out = bk;
    return FHE_OK;
}

void fhe_bootstrap_key_destroy(fhe_bootstrap_key *bk) {
    if (!bk) return;
    if (bk->ctx && bk->ctx->device >= 0) {
        DeviceGuard guard(bk->ctx->device);
        if (bk->d_ak_t) (void)hipFree(bk->d_ak_t);
        if (bk->d_dlog) (void)hipFree(bk->d_dlog);
        if (bk->d_status) (void)hipFree(bk->d_status);
    }
    delete bk;
}

// ops_out (host, optional): [batch][max_ops] with max_ops = n_lwe + N + 2, nops_out: [batch] -- lets tests compare the walk of
// blind_rotate_core with the oracle's.  Entry: bit 31 set -> automorphism ak[idx], else external product brk[idx].
int fhe_blind_rotate(const fhe_bootstrap_key *bk, const uint64_t *lwe_a, const uint64_t *lwe_b, const uint64_t *f, size_t f_stride,
                     uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream, uint32_t *ops_out,
                     uint32_t *nops_out) {
    if (!bk || ((!lwe_a || !lwe_b || !f || !out_a || !out_b) && batch)) return FHE_ERR_INVALID;
    const fhe_ctx *ctx = bk->ctx;
    if (ctx->device < 0) return FHE_ERR_NO_DEVICE;
    if (batch == 0) return FHE_OK;
    if (batch > 0x7fffffffull) return FHE_ERR_UNSUPPORTED;
    const int log_n = bk->brk->log_n;
    const size_t n = size_t(1) << log_n, n_lwe = bk->brk->count;
    if (f_stride != 0 && f_stride != n) return FHE_ERR_INVALID;
    if (n_lwe > 4096) return FHE_ERR_UNSUPPORTED;  // the schedule kernel sorts the LWE coefficients in LDS
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    Mirror ma(lwe_a, n_lwe * batch, mem, true, st), mb(lwe_b, batch, mem, true, st), mf(f, f_stride ? n * batch : n, mem, true, st);
    Mirror moa(out_a, n * batch, mem, false, st), mob(out_b, n * batch, mem, false, st);
    if (ma.rc | mb.rc | mf.rc | moa.rc | mob.rc) return FHE_ERR_HIP;
    const unsigned max_ops = (unsigned)(n_lwe + n + 2);
    const size_t ws_words = batch * max_ops + batch + 1;
    StreamWs wsp(ws_words * sizeof(unsigned), st);  // ops | nops | err
    if (wsp.rc != FHE_OK) return wsp.rc;
    unsigned *ws = wsp.as<unsigned>();
    unsigned *d_ops = ws, *d_nops = ws + batch * max_ops;
    // Device-memory calls are ASYNCHRONOUS: the data-dependent check (an even LWE coefficient) lands in the key's sticky status word,
    // read by fhe_bootstrap_key_status.  Host-memory calls (and calls that ask for the walk) synchronise anyway for their copies
    // and keep a word of their own.
    const bool async = mem == FHE_MEM_DEVICE && !(ops_out && nops_out);
    int *d_err = async ? bk->d_status : (int *)(d_nops + batch);
    int rc = FHE_OK;
    auto fail = [&](int code) { return code; };
    if (!async && hipMemsetAsync(d_err, 0, sizeof(int), st) != hipSuccess) return fail(FHE_ERR_HIP);
    hipLaunchKernelGGL(fhe::blind_rotate_schedule_kernel, dim3((unsigned)batch), dim3(64), 3 * n_lwe * sizeof(unsigned), st, ma.d,
                       (unsigned)n_lwe, (unsigned)batch, (unsigned)n, (unsigned)bk->w, bk->d_dlog, d_ops, d_nops, max_ops, d_err);
    if (hipGetLastError() != hipSuccess) return fail(FHE_ERR_HIP);
    fhe::BlindRotateParams BR;
    const bool small = small_shape(log_n, batch);
    BR.brk = key_view(bk->brk, small);
    BR.ak = key_view(bk->ak, small);
    BR.ak_t = bk->d_ak_t;
    BR.ops = d_ops;
    BR.nops = d_nops;
    BR.max_ops = max_ops;
    BR.lwe_b = mb.d;
    BR.f = mf.d;
    BR.f_stride = f_stride;
#define BR_LAUNCH_W(AR, WR)                                                                                                  \
    {                                                                                                                        \
        const size_t lds = WR::LDS_BYTES;                                                                                    \
        rc = set_lds(fhe::blind_rotate_kernel<AR, WR>, lds);                                                                 \
        if (rc == FHE_OK)                                                                                                    \
            hipLaunchKernelGGL((fhe::blind_rotate_kernel<AR, WR>), dim3((unsigned)((batch + WR::TEAMS - 1) / WR::TEAMS)),    \
                               dim3(WR::THREADS), lds, st, BR, moa.d, mob.d, (unsigned)batch, ring_consts(ctx, 0));          \
    }
#define BR_CASE(AR, LN) case LN: { typedef fhe::WaveRing<LN> WD; BR_LAUNCH_W(AR, WD) break; }
#define BR_CASE_BIG(AR, LN)                                                                  \
    case LN: {                                                                               \
        if (small) { typedef fhe::WaveRing<LN, 2> WS; BR_LAUNCH_W(AR, WS) }                  \
        else { typedef fhe::WaveRing<LN> WD; BR_LAUNCH_W(AR, WD) }                           \
        break;                                                                               \
    }
    const int pmv = fhew_pm(ctx, log_n);
    if (pmv == 54) {
        switch (log_n) {
            BR_CASE(FHEW_POLICY54, 9) BR_CASE_BIG(FHEW_POLICY54, 10) BR_CASE_BIG(FHEW_POLICY54, 11)
            default: return fail(FHE_ERR_UNSUPPORTED);
        }
    } else if (pmv == 55) {
        switch (log_n) {
            BR_CASE(FHEW_POLICY55, 9) BR_CASE_BIG(FHEW_POLICY55, 10) BR_CASE_BIG(FHEW_POLICY55, 11)
            default: return fail(FHE_ERR_UNSUPPORTED);
        }
    } else {
        switch (log_n) {
            BR_CASE(fhe::ArithShoup, 7) BR_CASE(fhe::ArithShoup, 8) BR_CASE(fhe::ArithShoup, 9) BR_CASE_BIG(fhe::ArithShoup, 10)
            BR_CASE_BIG(fhe::ArithShoup, 11)
            default: return fail(FHE_ERR_UNSUPPORTED);
        }
    }
#undef BR_CASE_BIG
#undef BR_LAUNCH_W
#undef BR_CASE
    if (rc != FHE_OK || hipGetLastError() != hipSuccess) return fail(FHE_ERR_HIP);
    if (async) return FHE_OK;  // (the workspace is stream ordered: released after the kernels above)
    int h_err = 0;
    if (hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return fail(FHE_ERR_HIP);
    if (ops_out && nops_out) {
        if (hipMemcpyAsync(ops_out, d_ops, batch * max_ops * sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(nops_out, d_nops, batch * sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess)
            return fail(FHE_ERR_HIP);
    }
    rc = moa.sync_out(st);
    if (rc == FHE_OK) rc = mob.sync_out(st);
    if (hipStreamSynchronize(st) != hipSuccess && rc == FHE_OK) rc = FHE_ERR_HIP;  // h_err (and ops_out) must have arrived
    if (rc == FHE_OK && h_err) rc = FHE_ERR_INVALID;  // an LWE coefficient outside the odd residues mod 2N (bootstrapping.rs:221)
    return rc;
}

// The sticky status of asynchronous blind rotations on this key: everything enqueued on `stream` so far is waited for, then
// FHE_ERR_INVALID if any of them met an LWE coefficient outside the odd residues mod 2N (bootstrapping.rs:221), else FHE_OK.
// clear != 0 resets the word (in stream order).
int fhe_bootstrap_key_status(const fhe_bootstrap_key *bk, void *stream, int clear) {
    if (!bk) return FHE_ERR_INVALID;
    if (bk->ctx->device < 0) return FHE_ERR_NO_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(bk->ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    int h = 0;
    HIP_TRY(hipMemcpyAsync(&h, bk->d_status, sizeof(int), hipMemcpyDeviceToHost, st));
    if (clear) HIP_TRY(hipMemsetAsync(bk->d_status, 0, sizeof(int), st));
    HIP_TRY(hipStreamSynchronize(st));
    return h ? FHE_ERR_INVALID : FHE_OK;
}

// scheme/fhew/src/bootstrapping.rs:149-155 `Bootstrapping::bootstrap(bk, f, ct)` for a batch, everything device side:
// mod_switch(Q_ks) -> Lwe::key_switch (ksk over q_ks, decomposer (ks_log_b, ks_d)) -> mod_switch_odd(2N) -> blind_rotate ->
// sample_extract(0); `addend` is added to the extracted b (Fhew::op adds Q/8, fhew.rs:39).
// ct_a [batch][N], ct_b [batch] over Q (LWE under the ring key); lwe_ksk_a [ks_d * N][n_lwe], lwe_ksk_b [ks_d * N] over q_ks.
int fhe_fhew_bootstrap(const fhe_bootstrap_key *bk, uint64_t q_ks, int ks_log_b, int ks_d, const uint64_t *lwe_ksk_a,
                       const uint64_t *lwe_ksk_b, const uint64_t *f, size_t f_stride, uint64_t addend, const uint64_t *ct_a,
                       const uint64_t *ct_b, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream) {
    if (!bk || !lwe_ksk_a || !lwe_ksk_b || !f || ((!ct_a || !ct_b || !out_a || !out_b) && batch)) return FHE_ERR_INVALID;
    {   // the key-switch gadget sizes every buffer below: validate it before anything is allocated
        fhe::DecompParams ksP;
        const int vrc = make_decomp(q_ks, ks_log_b, ks_d, &ksP);
        if (vrc != FHE_OK) return vrc;
    }
    if (batch == 0) return FHE_OK;
    const fhe_ctx *ctx = bk->ctx;
    const size_t n = size_t(1) << bk->brk->log_n, n_lwe = bk->brk->count;
    const uint64_t big_q = ctx->q;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t rows = n * ks_d;
    Mirror mka(lwe_ksk_a, rows * n_lwe, mem, true, st), mkb(lwe_ksk_b, rows, mem, true, st), mf(f, f_stride ? n * batch : n, mem, true, st);
    Mirror ma(ct_a, n * batch, mem, true, st), mb(ct_b, batch, mem, true, st);
    Mirror moa(out_a, n * batch, mem, false, st), mob(out_b, batch, mem, false, st);
    if (mka.rc | mkb.rc | mf.rc | ma.rc | mb.rc | moa.rc | mob.rc) return FHE_ERR_HIP;
    // a1 [batch][n] | b1 [batch] | a2 [batch][n_lwe] | b2 [batch] | a3 | b3 | acc_a, acc_b [batch][n]
    const size_t words = batch * (n + 1 + 2 * (n_lwe + 1) + 2 * n);
    StreamWs wsp(words * sizeof(u64), st);
    if (wsp.rc != FHE_OK) return wsp.rc;
    uint64_t *ws = wsp.as<uint64_t>();
    auto U = [](u64 *p) { return (uint64_t *)p; };
    uint64_t *a1 = ws, *b1 = a1 + batch * n, *a2 = b1 + batch, *b2 = a2 + batch * n_lwe, *a3 = b2 + batch, *b3 = a3 + batch * n_lwe,
        *ra = b3 + batch, *rb = ra + batch * n;
    int rc = fhe_lwe_mod_switch(big_q, q_ks, U(ma.d), a1, batch * n, 0, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = fhe_lwe_mod_switch(big_q, q_ks, U(mb.d), b1, batch, 0, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = fhe_lwe_key_switch(q_ks, ks_log_b, ks_d, U(mka.d), U(mkb.d), a1, b1, n, n_lwe, a2, b2, batch, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = fhe_lwe_mod_switch(q_ks, 2 * n, a2, a3, batch * n_lwe, 1, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = fhe_lwe_mod_switch(q_ks, 2 * n, b2, b3, batch, 1, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = fhe_blind_rotate(bk, a3, b3, U(mf.d), f_stride, ra, rb, batch, FHE_MEM_DEVICE, stream, nullptr, nullptr);
    if (rc == FHE_OK) rc = fhe_rlwe_sample_extract(big_q, ra, rb, n, 0, addend, U(moa.d), U(mob.d), batch, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = moa.sync_out(st);
    if (rc == FHE_OK) rc = mob.sync_out(st);
    // host-memory calls have synchronised for their outputs: report the blind rotation's data-dependent check with them
    if (rc == FHE_OK && mem == FHE_MEM_HOST) rc = fhe_bootstrap_key_status(bk, stream, 1);
    return rc;
}

}  // extern "C"
