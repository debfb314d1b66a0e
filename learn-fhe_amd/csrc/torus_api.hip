// extern "C" entry points for SURVEY.md section 8(a) row T: the TFHE torus path, k = 1.
#include <hip/hip_runtime.h>

#include <cmath>
#include <new>
#include <vector>

#include "torus_ctx.hpp"
#include "lwe_kernels.hpp"
#include "keygen_kernels.hpp"

namespace {

// ---- the three 30-bit primes of the torus30 path: the largest p < 2^30 with p = 1 (mod 2^12) ----
constexpr int T30_LOG_CAP = 11;  // tables for rings up to N = 2^11
struct Host30 {
    uint32_t p[3];
    std::vector<uint2> tw[3], twi[3];
    fhe::Mod30Desc desc[3];
};
inline uint32_t shoup32(uint64_t w, uint64_t p) { return (uint32_t)((w << 32) / p); }
void build_host30(Host30 &H) {
    int found = 0;
    for (uint64_t k = ((uint64_t(1) << 30) - 1) >> 12; k > 0 && found < 3; --k) {
        const uint64_t p = (k << 12) + 1;
        if (fhe::is_prime_u64(p)) H.p[found++] = (uint32_t)p;
    }
    for (int i = 0; i < 3; ++i) {
        const uint64_t p = H.p[i];
        const uint64_t g = fhe::smallest_nonresidue(p);
        const uint64_t psi = fhe::powmod(g, (p - 1) >> (T30_LOG_CAP + 1), p), psi_inv = fhe::invmod(psi, p);  // primitive 2^12-th root
        const size_t cap = size_t(1) << T30_LOG_CAP;
        std::vector<uint64_t> pw(cap), pwi(cap);
        uint64_t x = 1, y = 1;
        for (size_t j = 0; j < cap; ++j) { pw[j] = x; pwi[j] = y; x = fhe::mulmod(x, psi, p); y = fhe::mulmod(y, psi_inv, p); }
        H.tw[i].resize(cap); H.twi[i].resize(cap);
        for (size_t j = 0; j < cap; ++j) {
            const size_t r = fhe::bitrev((unsigned)j, T30_LOG_CAP);
            H.tw[i][j] = uint2{(uint32_t)pw[r], shoup32(pw[r], p)};
            H.twi[i][j] = uint2{(uint32_t)pwi[r], shoup32(pwi[r], p)};
        }
        fhe::Mod30Desc &D = H.desc[i];
        D.p = (uint32_t)p;
        for (int k = 0; k < 12; ++k) {
            const uint64_t ni = fhe::invmod((uint64_t(1) << k) % p, p);
            D.ninv[k] = (uint32_t)ni; D.ninv_s[k] = shoup32(ni, p);
        }
        uint32_t inv = 1;  // p^-1 mod 2^32 by Newton iteration
        for (int it = 0; it < 5; ++it) inv *= 2 - (uint32_t)p * inv;
        D.pinv_neg = 0u - inv;
        D.r2 = (uint32_t)((((fhe::u128)1) << 64) % p);
        D.r1 = (uint32_t)((uint64_t(1) << 32) % p);
    }
}

// the team shape of the torus kernels per ring degree: 4 coefficients per lane from N = 1024 up (the state of a CMUX -- accumulator,
// difference, digit state, two unreduced sum pairs -- is heavier than FHEW's; measured at cfg5: 17.2-17.3 k gates/s against
// 16.6-16.9 k with 8 per lane, same session, and the better shape for small batches as well)
#ifndef FHE_TORUS_LOG_E
#define FHE_TORUS_LOG_E 2
#endif
template <int LN>
using TorusRing = fhe::WaveRing<LN, (LN <= 9 ? LN - 6 : FHE_TORUS_LOG_E)>;
// the 30-bit path carries half the registers per coefficient: 8 per lane again (22.9 k against 21.8 k gates/s at cfg5)
template <int LN>
using TorusRing30 = fhe::WaveRing<LN, (LN <= 9 ? LN - 6 : 3)>;

#define TORUS_DISPATCH(log_n, ...)                                         \
    switch (log_n) {                                                       \
        case 8: { constexpr int LN = 8; __VA_ARGS__; break; }              \
        case 9: { constexpr int LN = 9; __VA_ARGS__; break; }              \
        case 10: { constexpr int LN = 10; __VA_ARGS__; break; }            \
        case 11: { constexpr int LN = 11; __VA_ARGS__; break; }            \
        default: return FHE_ERR_UNSUPPORTED;                               \
    }

// fft64 mode: a team over the N / 2 complex slots of a polynomial, 4 slots per lane from N = 1024 up (two waves per SIMD)
constexpr int TF_LOG_CAP = 11;
template <int LN>
using TorusRingF = fhe::WaveRing<LN - 1, (LN - 1 <= 8 ? LN - 1 - 6 : 2)>;
constexpr int TF_MIN_WAVES = 2;

template <class K>
int set_lds(K kernel, size_t lds) {
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return FHE_OK;
}

int launch_cmux(const fhe_torus_ctx *t, const fhe_tggsw_key *key, size_t index, u64 *a, u64 *b, size_t batch, const u64 *rot,
                size_t rot_stride, hipStream_t st) {
    const size_t n = size_t(1) << key->log_n;
    const size_t per = size_t(2 * key->d) * 2 * n;
    if (key->d_rowsf) {  // fft64 mode
        const double2 *frows = key->d_rowsf + index * (per / 2);
        TORUS_DISPATCH(key->log_n, {
            typedef TorusRingF<LN> WR;
            hipLaunchKernelGGL((fhe::torusf_cmux_kernel<WR, TF_MIN_WAVES>), dim3((unsigned)((batch + WR::TEAMS - 1) / WR::TEAMS)), dim3(WR::THREADS),
                               fhe::TorusF<WR>::LDS_BYTES, st, a, b, (unsigned)batch, frows, key->P, rot, rot_stride, (const double2 *)t->d_twf);
        });
        HIP_TRY(hipGetLastError());
        return FHE_OK;
    }
    if (key->d_rowsx3) {  // exact, three key pieces through f64 transforms
        const double2 *xrows = key->d_rowsx3 + index * (per / 2) * 3;
        TORUS_DISPATCH(key->log_n, {
            typedef TorusRingF<LN> WR;
            const size_t lds = fhe::TorusX3<WR>::lds_bytes(2 * key->d);
            if (set_lds((fhe::torusx3_cmux_kernel<WR, TF_MIN_WAVES>), lds) != FHE_OK) return FHE_ERR_HIP;
            hipLaunchKernelGGL((fhe::torusx3_cmux_kernel<WR, TF_MIN_WAVES>), dim3((unsigned)((batch + WR::TEAMS - 1) / WR::TEAMS)), dim3(WR::THREADS), lds, st, a, b,
                               (unsigned)batch, xrows, key->P, rot, rot_stride, (const double2 *)t->d_twf);
        });
        HIP_TRY(hipGetLastError());
        return FHE_OK;
    }
    if (key->d_rows30) {  // three 30-bit primes
        const size_t plane = key->count * per;
        TORUS_DISPATCH(key->log_n, {
            typedef TorusRing30<LN> WR;
            const size_t lds = WR::TORUS_LDS_BYTES;
            if (lds > 64 * 1024)
                HIP_TRY(hipFuncSetAttribute((const void *)fhe::torus30_cmux_kernel<WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(fhe::torus30_cmux_kernel<WR>, dim3((unsigned)((batch + WR::TEAMS - 1) / WR::TEAMS)), dim3(WR::THREADS), lds, st, a, b,
                               (unsigned)batch, (const unsigned *)(key->d_rows30 + index * per), plane, key->P, rot, rot_stride, t->T30);
        });
        HIP_TRY(hipGetLastError());
        return FHE_OK;
    }
    const u64 *rows0 = key->d_rows[0] + index * per, *rows1 = key->d_rows[1] + index * per;
    TORUS_DISPATCH(key->log_n, {
        const size_t lds = TorusRing<LN>::TORUS_LDS_BYTES;
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)fhe::torus_cmux_kernel<TorusRing<LN>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(fhe::torus_cmux_kernel<TorusRing<LN>>, dim3((unsigned)((batch + TorusRing<LN>::TEAMS - 1) / TorusRing<LN>::TEAMS)), dim3(TorusRing<LN>::THREADS), lds, st, a, b, (unsigned)batch, rows0,
                           rows1, key->P, rot, rot_stride, t->T);
    });
    HIP_TRY(hipGetLastError());
    return FHE_OK;
}

}  // namespace

extern "C" {

void fhe_torus_ctx_destroy(fhe_torus_ctx *t) {
    if (!t) return;
    if (t->device >= 0) {
        DeviceGuard guard(t->device);
        if (t->d_descs) (void)hipFree(t->d_descs);
        if (t->d_blob30) (void)hipFree(t->d_blob30);
        if (t->d_twf) (void)hipFree(t->d_twf);
    }
    fhe_ctx_destroy(t->mods[0]);
    fhe_ctx_destroy(t->mods[1]);
    delete t;
}

int fhe_torus_ctx_create(int device, fhe_torus_ctx **out) {
    if (!out) return FHE_ERR_INVALID;
    *out = nullptr;
    if (device < 0) return FHE_ERR_NO_DEVICE;
    fhe_torus_ctx *t = new (std::nothrow) fhe_torus_ctx();
    if (!t) return FHE_ERR_INVALID;
    t->device = device;
    int rc = fhe_ctx_create(TORUS_P0, device, &t->mods[0]);
    if (rc == FHE_OK) rc = fhe_ctx_create(TORUS_P1, device, &t->mods[1]);
    if (rc != FHE_OK || t->mods[0]->pm_b != 60 || t->mods[1]->pm_b != 60) { fhe_torus_ctx_destroy(t); return rc != FHE_OK ? rc : FHE_ERR_UNSUPPORTED; }
    DeviceGuard guard(device);
    if (!guard.ok) { fhe_torus_ctx_destroy(t); return FHE_ERR_HIP; }
    fhe::ModDesc descs[2] = {t->mods[0]->h_desc, t->mods[1]->h_desc};
    hipError_t e = hipMalloc((void **)&t->d_descs, sizeof(descs));
    if (e == hipSuccess) e = hipMemcpy(t->d_descs, descs, sizeof(descs), hipMemcpyHostToDevice);
    if (e != hipSuccess) { g_last_hip = (int)e; fhe_torus_ctx_destroy(t); return FHE_ERR_HIP; }
    const fhe::u128 P = (fhe::u128)TORUS_P0 * TORUS_P1, Ph = P >> 1;
    t->T.descs = t->d_descs;
    t->T.p0 = TORUS_P0; t->T.p1 = TORUS_P1;
    t->T.inv01 = fhe::invmod(TORUS_P0 % TORUS_P1, TORUS_P1);
    t->T.inv01_s = fhe::shoup(t->T.inv01, TORUS_P1);
    t->T.P_lo = (uint64_t)P;
    t->T.Ph_hi = (uint64_t)(Ph >> 64); t->T.Ph_lo = (uint64_t)Ph;
    t->T.B0 = t->mods[0]->barrett; t->T.B1 = t->mods[1]->barrett;
    {   // three-prime 30-bit path: tables [3][tw | twi][2^11] uint2, then the three descriptors
        Host30 H;
        build_host30(H);
        const size_t cap = size_t(1) << T30_LOG_CAP, tbytes = cap * sizeof(uint2);
        const size_t blob = 6 * tbytes + 3 * sizeof(fhe::Mod30Desc);
        e = hipMalloc(&t->d_blob30, blob);
        if (e != hipSuccess) { g_last_hip = (int)e; fhe_torus_ctx_destroy(t); return FHE_ERR_HIP; }
        char *base = (char *)t->d_blob30;
        for (int i = 0; i < 3 && e == hipSuccess; ++i) {
            e = hipMemcpy(base + (2 * i) * tbytes, H.tw[i].data(), tbytes, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(base + (2 * i + 1) * tbytes, H.twi[i].data(), tbytes, hipMemcpyHostToDevice);
            H.desc[i].tw = (const uint2 *)(base + (2 * i) * tbytes);
            H.desc[i].twi = (const uint2 *)(base + (2 * i + 1) * tbytes);
        }
        if (e == hipSuccess) e = hipMemcpy(base + 6 * tbytes, H.desc, 3 * sizeof(fhe::Mod30Desc), hipMemcpyHostToDevice);
        if (e != hipSuccess) { g_last_hip = (int)e; fhe_torus_ctx_destroy(t); return FHE_ERR_HIP; }
        fhe::Torus30Consts &C = t->T30;
        C.descs = (const fhe::Mod30Desc *)(base + 6 * tbytes);
        const uint64_t p0 = H.p[0], p1 = H.p[1], p2 = H.p[2];
        C.p[0] = H.p[0]; C.p[1] = H.p[1]; C.p[2] = H.p[2];
        C.inv01 = (uint32_t)fhe::invmod(p0 % p1, p1); C.inv01_s = shoup32(C.inv01, p1);
        C.inv02 = (uint32_t)fhe::invmod(p0 % p2, p2); C.inv02_s = shoup32(C.inv02, p2);
        C.inv12 = (uint32_t)fhe::invmod(p1 % p2, p2); C.inv12_s = shoup32(C.inv12, p2);
        C.p01 = p0 * p1;
        const fhe::u128 P3 = (fhe::u128)C.p01 * p2, half = P3 >> 1;
        C.P_lo = (uint64_t)P3;
        C.h0 = (uint32_t)(half % p0);
        C.h1 = (uint32_t)((half / p0) % p1);
        C.h2 = (uint32_t)(half / p0 / p1);
    }
    {   // fft64 mode: the twiddle tree of torusf_kernels.hpp, node 2^l + i -> cis(pi (2 bitrev_l(i) + 1) / 2^(l+1))
        std::vector<double2> tw(size_t(1) << TF_LOG_CAP);
        tw[0] = double2{1.0, 0.0};
        for (int l = 0; l < TF_LOG_CAP; ++l)
            for (unsigned i = 0; i < (1u << l); ++i) {
                const double ang = M_PI * double(2 * (l ? fhe::bitrev(i, l) : 0u) + 1) / double(2u << l);
                tw[(size_t(1) << l) + i] = double2{cos(ang), sin(ang)};
            }
        e = hipMalloc((void **)&t->d_twf, tw.size() * sizeof(double2));
        if (e == hipSuccess) e = hipMemcpy(t->d_twf, tw.data(), tw.size() * sizeof(double2), hipMemcpyHostToDevice);
        if (e != hipSuccess) { g_last_hip = (int)e; fhe_torus_ctx_destroy(t); return FHE_ERR_HIP; }
    }
    *out = t;
    return FHE_OK;
}

int fhe_torus_decompose(int log_b, int d, const uint64_t *in, size_t n, size_t polys, uint64_t *out, fhe_mem mem, void *stream) {
    PtrDeviceGuard pguard(in, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    fhe::TDecomp P;
    int rc = make_tdecomp(log_b, d, &P);
    if (rc != FHE_OK) return rc;
    if ((!in || !out) && n * polys) return FHE_ERR_INVALID;
    if (n * polys == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    Mirror mi(in, n * polys, mem, true, st), mo(out, n * polys * d, mem, false, st);
    if (mi.rc | mo.rc) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::torus_decompose_kernel, dim3(grid_for(n * polys)), dim3(256), 0, st, mi.d, mo.d, n, polys, P);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

// exact a <- a * b in Z_{2^64}[X]/(X^n+1) (util/src/ring.rs:315-320 `Rt *= &Rt`), operands read as signed 64-bit integers;
// log_bound_b: |b_i| < 2^log_bound_b.  Exactness needs n * 2^(63 + log_bound_b) < p0 p1 / 2.
}  // extern "C"
namespace {
// a [batch][n] <- a * b (exact, mod 2^64) on device buffers; b [b_rows][n] cycled over the batch
int torus_mul_dev(const fhe_torus_ctx *t, u64 *a, const u64 *b, size_t b_rows, int log_n, size_t batch, hipStream_t st) {
    const size_t n = size_t(1) << log_n;
    StreamWs wsp(2 * n * (batch + b_rows) * sizeof(u64), st);
    if (wsp.rc != FHE_OK) return wsp.rc;
    u64 *ra = wsp.as<u64>(), *rb = ra + 2 * n * batch;  // adjacent: one forward launch over both
    hipLaunchKernelGGL(fhe::torus_residue2_kernel, dim3(grid_for(n * batch)), dim3(256), 0, st, (const u64 *)a, ra, n, batch, t->T.p0, t->T.p1);
    hipLaunchKernelGGL(fhe::torus_residue2_kernel, dim3(grid_for(n * b_rows)), dim3(256), 0, st, b, rb, n, b_rows, t->T.p0, t->T.p1);
    int rc = hipGetLastError() == hipSuccess ? FHE_OK : FHE_ERR_HIP;
    if (rc == FHE_OK) rc = fhe::ntt_fwd_multi(t->d_descs, 2, ra, log_n, 2 * (batch + b_rows), st, 60);
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::torus_pointwise_kernel, dim3(grid_for(2 * n * batch)), dim3(256), 0, st, ra, (const u64 *)rb, n, batch, b_rows, t->T.B0,
                           t->T.B1);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK) rc = fhe::ntt_inv_multi(t->d_descs, 2, ra, log_n, 2 * batch, st, 60);
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::torus_crt_kernel, dim3(grid_for(n * batch)), dim3(256), 0, st, (const u64 *)ra, a, n, batch, t->T);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    return rc;
}
}  // namespace
extern "C" {

int fhe_torus_mul(const fhe_torus_ctx *t, uint64_t *a, const uint64_t *b, int log_bound_b, size_t n, size_t batch, fhe_mem mem,
                  void *stream) {
    if (!t || !is_pow2(n) || ((!a || !b) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    const int log_n = ilog2(n);
    if (log_n > 15 || log_n < 1) return FHE_ERR_UNSUPPORTED;
    if (log_bound_b < 0 || 63 + log_bound_b + log_n + 1 > 118) return FHE_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    Mirror ma(a, n * batch, mem, true, st), mb(b, n * batch, mem, true, st);
    if (ma.rc | mb.rc) return FHE_ERR_HIP;
    int rc = torus_mul_dev(t, ma.d, mb.d, batch, log_n, batch, st);
    if (rc == FHE_OK) rc = ma.sync_out(st);
    return rc;
}

void fhe_tggsw_key_destroy(fhe_tggsw_key *k) {
    if (!k) return;
    if (k->t && k->t->device >= 0) {
        DeviceGuard guard(k->t->device);
        if (k->d_rows[0]) (void)hipFree(k->d_rows[0]);
        if (k->d_rows30) (void)hipFree(k->d_rows30);
        if (k->d_rowsf) (void)hipFree(k->d_rowsf);
        if (k->d_rowsx3) (void)hipFree(k->d_rowsx3);
    }
    delete k;
}

// `count` TGGSW ciphertexts with k = 1 (scheme/tfhe/src/tggsw.rs:44-88): rows_a / rows_b = the a / b polynomials of the 2d
// TGLWE rows of each, [count][2d][n] torus values.
int fhe_tggsw_prepare(const fhe_torus_ctx *t, int log_b, int d, const uint64_t *rows_a, const uint64_t *rows_b, size_t n, size_t count,
                      fhe_mem mem, fhe_tggsw_key **out) {
    if (!out) return FHE_ERR_INVALID;
    *out = nullptr;
    if (!t || !rows_a || !rows_b || !is_pow2(n) || count == 0) return FHE_ERR_INVALID;
    const int log_n = ilog2(n);
    if (log_n < 8 || log_n > 11) return FHE_ERR_UNSUPPORTED;
    fhe::TDecomp P;
    int rc = make_tdecomp(log_b, d, &P);
    if (rc != FHE_OK) return rc;
    // exactness: |coefficient| <= 2d * N * 2^(log_b - 1) * 2^63 must stay below P / 2
    const int bound_bits = ilog2((size_t)2 * d) + 1 + log_n + 62 + log_b;
    if (bound_bits > 118) return FHE_ERR_UNSUPPORTED;              // two 60-bit primes: P / 2 ~ 2^118.9
    const bool use30 = bound_bits <= 88 && log_b <= 28;            // three 30-bit primes: P / 2 ~ 2^88.9 (torus30_kernels.hpp)
    // exact through f64 transforms of three key pieces (torusf_kernels.hpp: TorusX3): 2d N 2^log_b <= 2^21 keeps the rounding error a factor
    // 12 inside 1/2; digits must be bytes (log_b <= 7) and their state a dword (log_b d <= 31), at most 16 limbs (the digit area in LDS);
    // N <= 1024 (four slots per lane at most)
    const bool usex3 = (size_t(2 * d) << (log_n + log_b)) <= (size_t(1) << 21) && log_b <= 7 && log_b * d <= 31 && 2 * d <= 16 && log_n <= 10 &&
                       fhe::opt(fhe::OPT_NO_F64_EXACT) == 0;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t rows = count * 2 * d, words = rows * n;
    hipStream_t st = nullptr;
    u64 *src = nullptr, *tmp = nullptr, *dst = nullptr;
    unsigned *dst30 = nullptr;
    HIP_TRY(hipMalloc((void **)&src, 2 * words * sizeof(u64)));
    hipMemcpyKind kind = mem == FHE_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    hipError_t e = hipMemcpyAsync(src, rows_a, words * sizeof(u64), kind, st);
    if (e == hipSuccess) e = hipMemcpyAsync(src + words, rows_b, words * sizeof(u64), kind, st);
    rc = e == hipSuccess ? FHE_OK : FHE_ERR_HIP;
    if (e != hipSuccess) g_last_hip = (int)e;
    double2 *dstx3 = nullptr;
    if (rc == FHE_OK && usex3) {
        if (hipMalloc((void **)&dstx3, 6 * rows * (n / 2) * sizeof(double2)) != hipSuccess) rc = FHE_ERR_HIP;
        if (rc == FHE_OK) {
            TORUS_DISPATCH(log_n, {
                typedef TorusRingF<LN> WR;
                hipLaunchKernelGGL(fhe::torusx3_key_prepare_kernel<WR>, dim3((unsigned)((6 * rows + WR::TEAMS - 1) / WR::TEAMS)), dim3(WR::THREADS),
                                   fhe::TorusF<WR>::LDS_BYTES, st, (const u64 *)src, (const u64 *)(src + words), rows, (const double2 *)t->d_twf, dstx3);
            });
            if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
        }
    } else if (rc == FHE_OK && use30) {
        if (hipMalloc((void **)&dst30, 3 * 2 * words * sizeof(unsigned)) != hipSuccess) rc = FHE_ERR_HIP;
        for (int pi = 0; pi < 3 && rc == FHE_OK; ++pi) {
            TORUS_DISPATCH(log_n, {
                typedef TorusRing30<LN> WR;
                if (WR::LDS_BYTES > 64 * 1024)
                    HIP_TRY(hipFuncSetAttribute((const void *)fhe::torus30_key_prepare_kernel<WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WR::LDS_BYTES));
                hipLaunchKernelGGL(fhe::torus30_key_prepare_kernel<WR>, dim3((unsigned)((2 * rows + WR::TEAMS - 1) / WR::TEAMS)), dim3(WR::THREADS),
                                   WR::LDS_BYTES, st, (const u64 *)src, (const u64 *)(src + words), rows, t->T30.descs + pi,
                                   dst30 + size_t(pi) * 2 * words);
            });
            if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
        }
    } else if (rc == FHE_OK) {
        e = hipMalloc((void **)&tmp, 2 * words * sizeof(u64));
        if (e == hipSuccess) e = hipMalloc((void **)&dst, 4 * words * sizeof(u64));  // both primes
        if (e != hipSuccess) { g_last_hip = (int)e; rc = FHE_ERR_HIP; }
        for (int pi = 0; pi < 2 && rc == FHE_OK; ++pi) {
            hipLaunchKernelGGL(fhe::torus_residue_kernel, dim3(grid_for(2 * words)), dim3(256), 0, st, (const u64 *)src, tmp, 2 * words,
                               pi ? t->T.p1 : t->T.p0);
            if (hipGetLastError() != hipSuccess) { rc = FHE_ERR_HIP; break; }
            rc = fhe::ntt_fwd_multi(t->d_descs + pi, 1, tmp, log_n, 2 * rows, st, 60);
            if (rc != FHE_OK) break;
            TORUS_DISPATCH(log_n, hipLaunchKernelGGL(fhe::key_permute_kernel<TorusRing<LN>>, dim3(grid_for(words)), dim3(256), 0, st, (const u64 *)tmp,
                                                     (const u64 *)(tmp + words), dst + size_t(pi) * 2 * words, rows, 60));
            if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
        }
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == FHE_OK) rc = FHE_ERR_HIP;
    (void)hipFree(src);
    if (tmp) (void)hipFree(tmp);
    if (rc != FHE_OK) { if (dst) (void)hipFree(dst); if (dst30) (void)hipFree(dst30); if (dstx3) (void)hipFree(dstx3); return rc; }
    fhe_tggsw_key *k = new (std::nothrow) fhe_tggsw_key();
    if (!k) { if (dst) (void)hipFree(dst); if (dst30) (void)hipFree(dst30); if (dstx3) (void)hipFree(dstx3); return FHE_ERR_INVALID; }
    k->t = t; k->log_n = log_n; k->log_b = log_b; k->d = d; k->count = count; k->P = P;
    k->d_rows[0] = dst; k->d_rows[1] = dst ? dst + 2 * words : nullptr; k->d_rows30 = dst30; k->d_rowsx3 = dstx3;
    *out = k;
    return FHE_OK;
}

// The same key prepared for the f64 FFT product the reference itself uses (util/src/ring/fft/c64.rs:11-56): every entry point that
// takes the key then runs torusf_kernels.hpp.  NOT exact: per product |result - exact| <= 2^(64 + log_b + log2 n - 53) (c64.rs:186-208).
int fhe_tggsw_prepare_fft64(const fhe_torus_ctx *t, int log_b, int d, const uint64_t *rows_a, const uint64_t *rows_b, size_t n, size_t count, fhe_mem mem,
                            fhe_tggsw_key **out) {
    if (!out) return FHE_ERR_INVALID;
    *out = nullptr;
    if (!t || !rows_a || !rows_b || !is_pow2(n) || count == 0) return FHE_ERR_INVALID;
    const int log_n = ilog2(n);
    if (log_n < 8 || log_n > 11) return FHE_ERR_UNSUPPORTED;
    fhe::TDecomp P;
    int rc = make_tdecomp(log_b, d, &P);
    if (rc != FHE_OK) return rc;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t rows = count * 2 * d, words = rows * n;
    hipStream_t st = nullptr;
    Mirror ma(rows_a, words, mem, true, st), mb(rows_b, words, mem, true, st);
    if (ma.rc | mb.rc) return FHE_ERR_HIP;
    double2 *dst = nullptr;
    HIP_TRY(hipMalloc((void **)&dst, 2 * rows * (n / 2) * sizeof(double2)));
    TORUS_DISPATCH(log_n, {
        typedef TorusRingF<LN> WR;
        hipLaunchKernelGGL(fhe::torusf_key_prepare_kernel<WR>, dim3((unsigned)((2 * rows + WR::TEAMS - 1) / WR::TEAMS)), dim3(WR::THREADS), fhe::TorusF<WR>::LDS_BYTES, st,
                           (const u64 *)ma.d, (const u64 *)mb.d, rows, (const double2 *)t->d_twf, dst);
    });
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipFree(dst); return FHE_ERR_HIP; }
    fhe_tggsw_key *k = new (std::nothrow) fhe_tggsw_key();
    if (!k) { (void)hipFree(dst); return FHE_ERR_INVALID; }
    k->t = t; k->log_n = log_n; k->log_b = log_b; k->d = d; k->count = count; k->P = P; k->d_rowsf = dst;
    *out = k;
    return FHE_OK;
}

// scheme/tfhe/src/tggsw.rs:100-112 `Tggsw::external_product(param, key[index], ct)`, in place on [batch][n] a / b
int fhe_tggsw_external_product(const fhe_torus_ctx *t, const fhe_tggsw_key *key, size_t index, uint64_t *ct_a, uint64_t *ct_b, size_t batch,
                               fhe_mem mem, void *stream) {
    if (!t || !key || key->t != t || index >= key->count || ((!ct_a || !ct_b) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t n = size_t(1) << key->log_n;
    Mirror ma(ct_a, n * batch, mem, true, st), mb(ct_b, n * batch, mem, true, st);
    if (ma.rc | mb.rc) return FHE_ERR_HIP;
    int rc = launch_cmux(t, key, index, ma.d, mb.d, batch, nullptr, 0, st);
    if (rc == FHE_OK) rc = ma.sync_out(st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    return rc;
}

// scheme/tfhe/src/tggsw.rs:114-121 `Tggsw::cmux(b, ct0, ct1)` = ct0 + external_product(b, ct1 - ct0) for `batch` pairs of TGLWE
// ciphertexts (k = 1) under key[index]: [batch][n] each; out may alias ct0 or ct1.
int fhe_tggsw_cmux(const fhe_torus_ctx *t, const fhe_tggsw_key *key, size_t index, const uint64_t *ct0_a, const uint64_t *ct0_b,
                   const uint64_t *ct1_a, const uint64_t *ct1_b, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream) {
    if (!t || !key || key->t != t || index >= key->count || ((!ct0_a || !ct0_b || !ct1_a || !ct1_b || !out_a || !out_b) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t n = size_t(1) << key->log_n, words = n * batch;
    Mirror m0a(ct0_a, words, mem, true, st), m0b(ct0_b, words, mem, true, st), m1a(ct1_a, words, mem, true, st), m1b(ct1_b, words, mem, true, st);
    Mirror moa(out_a, words, mem, false, st), mob(out_b, words, mem, false, st);
    if (m0a.rc | m0b.rc | m1a.rc | m1b.rc | moa.rc | mob.rc) return FHE_ERR_HIP;
    StreamWs ws(2 * words * sizeof(u64), st);  // the difference, then its external product (in place)
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *da = ws.as<u64>(), *db = da + words;
    const dim3 grid(grid_for(words));
    hipLaunchKernelGGL(fhe::torus_addsub_kernel, grid, dim3(256), 0, st, (const u64 *)m1a.d, (const u64 *)m0a.d, da, words, 1);
    hipLaunchKernelGGL(fhe::torus_addsub_kernel, grid, dim3(256), 0, st, (const u64 *)m1b.d, (const u64 *)m0b.d, db, words, 1);
    HIP_TRY(hipGetLastError());
    int rc = launch_cmux(t, key, index, da, db, batch, nullptr, 0, st);
    if (rc != FHE_OK) return rc;
    hipLaunchKernelGGL(fhe::torus_addsub_kernel, grid, dim3(256), 0, st, (const u64 *)m0a.d, (const u64 *)da, moa.d, words, 0);
    hipLaunchKernelGGL(fhe::torus_addsub_kernel, grid, dim3(256), 0, st, (const u64 *)m0b.d, (const u64 *)db, mob.d, words, 0);
    HIP_TRY(hipGetLastError());
    rc = moa.sync_out(st);
    return rc != FHE_OK ? rc : mob.sync_out(st);
}

// scheme/tfhe/src/tglwe.rs:61-66 `TglweCiphertext::rotate(i)`: both halves times X^i; ct, out [batch][n], out != ct
int fhe_tglwe_rotate(const uint64_t *ct_a, const uint64_t *ct_b, size_t n, int64_t i, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem,
                     void *stream) {
    PtrDeviceGuard pguard(ct_a, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    if (!is_pow2(n) || (n >> 30) || ((!ct_a || !ct_b || !out_a || !out_b) && batch) || (batch && (ct_a == out_a || ct_b == out_b))) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    const int64_t two_n = 2 * (int64_t)n;
    const unsigned k = (unsigned)(((i % two_n) + two_n) % two_n);  // `X ^ i`: i.rem_euclid(2n) (util/src/ring.rs:380-386)
    hipStream_t st = (hipStream_t)stream;
    const size_t words = n * batch;
    Mirror ma(ct_a, words, mem, true, st), mb(ct_b, words, mem, true, st), moa(out_a, words, mem, false, st), mob(out_b, words, mem, false, st);
    if (ma.rc | mb.rc | moa.rc | mob.rc) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::torus_monomial_kernel, dim3(grid_for(words)), dim3(256), 0, st, (const u64 *)ma.d, moa.d, (unsigned)n, batch, k);
    hipLaunchKernelGGL(fhe::torus_monomial_kernel, dim3(grid_for(words)), dim3(256), 0, st, (const u64 *)mb.d, mob.d, (unsigned)n, batch, k);
    HIP_TRY(hipGetLastError());
    int rc = moa.sync_out(st);
    return rc != FHE_OK ? rc : mob.sync_out(st);
}

// scheme/tfhe/src/bootstrapping.rs:99-104 `mod_switch`: v -> rounding_shr(v, 64 - log2(2 big_n)) for `count` torus values
int fhe_tfhe_mod_switch(const uint64_t *in, uint64_t *out, size_t count, size_t big_n, fhe_mem mem, void *stream) {
    PtrDeviceGuard pguard(in, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    if (!is_pow2(big_n) || ((!in || !out) && count)) return FHE_ERR_INVALID;
    if (count == 0) return FHE_OK;
    const int bits = 64 - (ilog2(big_n) + 1);
    hipStream_t st = (hipStream_t)stream;
    Mirror mi(in, count, mem, true, st), mo(out, count, mem, false, st);
    if (mi.rc | mo.rc) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::torus_rounding_shr_kernel, dim3(grid_for(count)), dim3(256), 0, st, (const u64 *)mi.d, mo.d, count, bits);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

// scheme/tfhe/src/bootstrapping.rs:84-96 `blind_rotate` for a batch (k = 1): brk = key (n_lwe TGGSW ciphertexts);
// a_tilde [batch][n_lwe], b_tilde [batch]: the mod-switched TLWE ciphertexts (values mod 2N); v: the ENCODED test polynomial
// (Tglwe::encode(v), [n] torus values, shared by the batch); out_a, out_b [batch][n].
int fhe_tfhe_blind_rotate(const fhe_torus_ctx *t, const fhe_tggsw_key *brk, const uint64_t *a_tilde, const uint64_t *b_tilde,
                          const uint64_t *v, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream) {
    if (!t || !brk || brk->t != t || ((!a_tilde || !b_tilde || !v || !out_a || !out_b) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t n = size_t(1) << brk->log_n, n_lwe = brk->count;
    Mirror ma(a_tilde, n_lwe * batch, mem, true, st), mb(b_tilde, batch, mem, true, st), mv(v, n, mem, true, st);
    Mirror moa(out_a, n * batch, mem, false, st), mob(out_b, n * batch, mem, false, st);
    if (ma.rc | mb.rc | mv.rc | moa.rc | mob.rc) return FHE_ERR_HIP;
    if (batch > 0x7fffffffull || n_lwe > 0x7fffffffull) return FHE_ERR_UNSUPPORTED;
    // acc = (0, v).rotate(-b), then fold cmux(brk_i, acc, acc.rotate(a_i)) (bootstrapping.rs:91-95): one launch, the
    // accumulator never leaves the registers of the team that owns the ciphertext
    int rc = FHE_OK;
    if (brk->d_rowsf) {  // fft64 mode
        TORUS_DISPATCH(brk->log_n, {
            typedef TorusRingF<LN> WR;
            hipLaunchKernelGGL((fhe::torusf_blind_rotate_kernel<WR, TF_MIN_WAVES>), dim3((unsigned)((batch + WR::TEAMS - 1) / WR::TEAMS)), dim3(WR::THREADS),
                               fhe::TorusF<WR>::LDS_BYTES, st, (const u64 *)mv.d, (const u64 *)ma.d, (const u64 *)mb.d, (unsigned)n_lwe, (unsigned)batch,
                               (const double2 *)brk->d_rowsf, brk->P, (const double2 *)t->d_twf, moa.d, mob.d);
        });
    } else if (brk->d_rowsx3) {  // exact, three key pieces through f64 transforms
        TORUS_DISPATCH(brk->log_n, {
            typedef TorusRingF<LN> WR;
            const size_t lds = fhe::TorusX3<WR>::lds_bytes(2 * brk->d);
            if (set_lds((fhe::torusx3_blind_rotate_kernel<WR, TF_MIN_WAVES>), lds) != FHE_OK) return FHE_ERR_HIP;
            hipLaunchKernelGGL((fhe::torusx3_blind_rotate_kernel<WR, TF_MIN_WAVES>), dim3((unsigned)((batch + WR::TEAMS - 1) / WR::TEAMS)), dim3(WR::THREADS), lds, st,
                               (const u64 *)mv.d, (const u64 *)ma.d, (const u64 *)mb.d, (unsigned)n_lwe, (unsigned)batch, (const double2 *)brk->d_rowsx3, brk->P,
                               (const double2 *)t->d_twf, moa.d, mob.d);
        });
    } else if (brk->d_rows30) {  // three 30-bit primes
        const size_t plane = brk->count * size_t(2 * brk->d) * 2 * n;
        // base <= 2^8 and at most 8 limbs (cfg5: base 2^7, d = 3): the CMUX digits are computed once and parked as bytes, the
        // multiply-accumulate runs unreduced (torus30_kernels.hpp); lab switch NO_PACKED_DIGITS keeps the older kernel (bit-identical)
        // (a digit of base 2^8 ranges over [-128, 128]: one value too many for a byte)
        const bool packed = brk->P.log_b <= 7 && 2 * brk->d <= 8 && brk->log_n >= 8 && fhe::opt(fhe::OPT_NO_PACKED_DIGITS) == 0;
        if (packed) {
            TORUS_DISPATCH(brk->log_n, {
                typedef TorusRing30<LN> WR;
                const size_t lds = WR::torus_pk_lds_bytes(2 * brk->d);
                if (lds > 64 * 1024) {
                    static std::atomic<int> done{0};
                    if (!done.load(std::memory_order_acquire)) {
                        HIP_TRY(hipFuncSetAttribute((const void *)fhe::torus30_blind_rotate_pk_kernel<WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                        done.store(1, std::memory_order_release);
                    }
                }
                hipLaunchKernelGGL(fhe::torus30_blind_rotate_pk_kernel<WR>, dim3((unsigned)((batch + WR::TEAMS - 1) / WR::TEAMS)), dim3(WR::THREADS), lds, st,
                                   (const u64 *)mv.d, (const u64 *)ma.d, (const u64 *)mb.d, (unsigned)n_lwe, (unsigned)batch,
                                   (const unsigned *)brk->d_rows30, plane, brk->P, t->T30, moa.d, mob.d);
            });
        } else
        TORUS_DISPATCH(brk->log_n, {
            typedef TorusRing30<LN> WR;
            const size_t lds = WR::TORUS_LDS_BYTES;
            if (lds > 64 * 1024)
                HIP_TRY(hipFuncSetAttribute((const void *)fhe::torus30_blind_rotate_kernel<WR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(fhe::torus30_blind_rotate_kernel<WR>, dim3((unsigned)((batch + WR::TEAMS - 1) / WR::TEAMS)), dim3(WR::THREADS), lds, st,
                               (const u64 *)mv.d, (const u64 *)ma.d, (const u64 *)mb.d, (unsigned)n_lwe, (unsigned)batch,
                               (const unsigned *)brk->d_rows30, plane, brk->P, t->T30, moa.d, mob.d);
        });
    } else {
    TORUS_DISPATCH(brk->log_n, {
            const size_t lds = TorusRing<LN>::TORUS_LDS_BYTES;
            if (lds > 64 * 1024)
                HIP_TRY(hipFuncSetAttribute((const void *)fhe::torus_blind_rotate_kernel<TorusRing<LN>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(fhe::torus_blind_rotate_kernel<TorusRing<LN>>, dim3((unsigned)((batch + TorusRing<LN>::TEAMS - 1) / TorusRing<LN>::TEAMS)),
                               dim3(TorusRing<LN>::THREADS), lds, st, (const u64 *)mv.d, (const u64 *)ma.d, (const u64 *)mb.d, (unsigned)n_lwe,
                               (unsigned)batch, (const u64 *)brk->d_rows[0], (const u64 *)brk->d_rows[1], brk->P, t->T, moa.d, mob.d);
        });
    }
    if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    if (rc == FHE_OK) rc = moa.sync_out(st);
    if (rc == FHE_OK) rc = mob.sync_out(st);
    return rc;
}

// scheme/tfhe/src/tglwe.rs:115-127 `sample_extract(ct, i)` (k = 1): ct_a, ct_b [batch][n] -> TLWE (out_a [batch][n], out_b [batch])
int fhe_tglwe_sample_extract(const uint64_t *ct_a, const uint64_t *ct_b, size_t n, size_t index, uint64_t *out_a, uint64_t *out_b,
                             size_t batch, fhe_mem mem, void *stream) {
    PtrDeviceGuard pguard(ct_a, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    if (!is_pow2(n) || index >= n || n > (1u << 30) || ((!ct_a || !ct_b || !out_a || !out_b) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    Mirror ma(ct_a, n * batch, mem, true, st), mb(ct_b, n * batch, mem, true, st), moa(out_a, n * batch, mem, false, st),
        mob(out_b, batch, mem, false, st);
    if (ma.rc | mb.rc | moa.rc | mob.rc) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::tglwe_sample_extract_kernel, dim3(grid_for(n * batch)), dim3(256), 0, st, (const u64 *)ma.d, (const u64 *)mb.d,
                       (unsigned)n, batch, (unsigned)index, moa.d, mob.d);
    HIP_TRY(hipGetLastError());
    int rc = moa.sync_out(st);
    return rc != FHE_OK ? rc : mob.sync_out(st);
}

// scheme/tfhe/src/tlwe.rs:144-153 `Tlwe::key_switch`: ksk_a [n_in * d][n_out], ksk_b [n_in * d] (row j * n_in + i = digit j of
// input coefficient i); ct_a [batch][n_in], ct_b [batch] -> out_a [batch][n_out], out_b [batch]
int fhe_tlwe_key_switch(int log_b, int d, const uint64_t *ksk_a, const uint64_t *ksk_b, const uint64_t *ct_a, const uint64_t *ct_b,
                        size_t n_in, size_t n_out, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream) {
    PtrDeviceGuard pguard(ct_a, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    fhe::TDecomp P;
    int rc = make_tdecomp(log_b, d, &P);
    if (rc != FHE_OK) return rc;
    if (!ksk_a || !ksk_b || n_in == 0 || n_out == 0 || ((!ct_a || !ct_b || !out_a || !out_b) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = n_in * d;
    Mirror mka(ksk_a, rows * n_out, mem, true, st), mkb(ksk_b, rows, mem, true, st), ma(ct_a, n_in * batch, mem, true, st),
        mb(ct_b, batch, mem, true, st), moa(out_a, n_out * batch, mem, false, st), mob(out_b, batch, mem, false, st);
    if (mka.rc | mkb.rc | ma.rc | mb.rc | moa.rc | mob.rc) return FHE_ERR_HIP;
    const int tile = P.log_b <= 31 ? fhe::ks_tile(batch, rows) : 0;  // the tiled kernel keeps digits as 32-bit words
    constexpr int WIDE_TILE = 16;
    if (P.log_b <= 7 && n_out >= 256 && batch >= 64 && batch < (size_t(1) << 31) && n_in < (size_t(1) << 31)) {
        // many output columns: ciphertext tiles x column blocks x input-coefficient chunks, partial sums added atomically (lwe_kernels.hpp)
        const unsigned tiles = (unsigned)((batch + WIDE_TILE - 1) / WIDE_TILE), colb = (unsigned)((n_out + 1 + fhe::KS_THREADS - 1) / fhe::KS_THREADS);
        // enough blocks for ~8 waves per SIMD, chunks of at least 64 coefficients
        unsigned z = (unsigned)((size_t(16) * fhe::current_cu_count() + size_t(tiles) * colb - 1) / (size_t(tiles) * colb));
        const unsigned zmax = (unsigned)((n_in + 63) / 64);
        const size_t chunk_cap = (48 * 1024) / (size_t(P.d) * WIDE_TILE);  // the digit image of a block stays below 48 KiB of LDS
        const unsigned zmin = (unsigned)((n_in + chunk_cap - 1) / chunk_cap);
        z = z > zmax ? zmax : z;
        z = z < zmin ? zmin : (z < 1 ? 1 : z);
        const unsigned chunk = (unsigned)((n_in + z - 1) / z);
        z = (unsigned)((n_in + chunk - 1) / chunk);
        HIP_TRY(hipMemsetAsync(moa.d, 0, n_out * batch * sizeof(u64), st));
        HIP_TRY(hipMemsetAsync(mob.d, 0, batch * sizeof(u64), st));
        const size_t lds = size_t(P.d) * chunk * WIDE_TILE;
        hipLaunchKernelGGL(fhe::tlwe_key_switch_split<WIDE_TILE>, dim3(tiles, colb, z), dim3(fhe::KS_THREADS), lds, st, (const u64 *)ma.d, (const u64 *)mb.d,
                           (unsigned)n_in, (unsigned)n_out, (unsigned)batch, (const u64 *)mka.d, (const u64 *)mkb.d, P, moa.d, mob.d, chunk);
    } else if (tile && batch < (size_t(1) << 31)) {
        if (fhe::launch_key_switch_tiled(fhe::KsTorus{P}, ma.d, mb.d, n_in, n_out, batch, mka.d, mkb.d, moa.d, mob.d, tile, st)) return FHE_ERR_HIP;
    } else {
        hipLaunchKernelGGL(fhe::tlwe_key_switch_kernel, dim3(grid_for((n_out + 1) * batch)), dim3(256), 0, st, (const u64 *)ma.d,
                           (const u64 *)mb.d, (unsigned)n_in, (unsigned)n_out, batch, (const u64 *)mka.d, (const u64 *)mkb.d, P, moa.d, mob.d);
    }
    HIP_TRY(hipGetLastError());
    rc = moa.sync_out(st);
    return rc != FHE_OK ? rc : mob.sync_out(st);
}

// scheme/tfhe/src/bootstrapping.rs:78-82 `Bootstrapping::bootstrap` for `batch` TLWE ciphertexts in ONE call: mod switch (99-104),
// blind rotation (84-96), sample_extract(0) (tglwe.rs:115-127), TLWE key switch (tlwe.rs:144-153), all on `stream`, intermediates
// in stream-ordered scratch.  Bit-identical to the four separate calls.
int fhe_tfhe_bootstrap(const fhe_torus_ctx *t, const fhe_tggsw_key *brk, int ks_log_b, int ks_d, const uint64_t *ksk_a, const uint64_t *ksk_b,
                       const uint64_t *v, const uint64_t *lwe_a, const uint64_t *lwe_b, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem,
                       void *stream) {
    fhe::TDecomp P;
    int rc = make_tdecomp(ks_log_b, ks_d, &P);
    if (rc != FHE_OK) return rc;
    if (!t || !brk || brk->t != t || !ksk_a || !ksk_b || !v || ((!lwe_a || !lwe_b || !out_a || !out_b) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t n = size_t(1) << brk->log_n, n_lwe = brk->count, rows = n * ks_d;
    Mirror mka(ksk_a, rows * n_lwe, mem, true, st), mkb(ksk_b, rows, mem, true, st), mv(v, n, mem, true, st),
        ma(lwe_a, n_lwe * batch, mem, true, st), mb(lwe_b, batch, mem, true, st), moa(out_a, n_lwe * batch, mem, false, st),
        mob(out_b, batch, mem, false, st);
    if (mka.rc | mkb.rc | mv.rc | ma.rc | mb.rc | moa.rc | mob.rc) return FHE_ERR_HIP;
    StreamWs ws((batch * n_lwe + batch + 3 * batch * n + batch) * sizeof(u64), st);  // a~ | b~ | acc.a | acc.b | extracted a | b
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *at = ws.as<u64>(), *bt = at + batch * n_lwe, *ra = bt + batch, *rb = ra + batch * n, *ea = rb + batch * n, *eb = ea + batch * n;
    typedef uint64_t U;
    rc = fhe_tfhe_mod_switch((const U *)ma.d, (U *)at, batch * n_lwe, n, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = fhe_tfhe_mod_switch((const U *)mb.d, (U *)bt, batch, n, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = fhe_tfhe_blind_rotate(t, brk, (const U *)at, (const U *)bt, (const U *)mv.d, (U *)ra, (U *)rb, batch, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = fhe_tglwe_sample_extract((const U *)ra, (const U *)rb, n, 0, (U *)ea, (U *)eb, batch, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK)
        rc = fhe_tlwe_key_switch(ks_log_b, ks_d, (const U *)mka.d, (const U *)mkb.d, (const U *)ea, (const U *)eb, n, n_lwe, (U *)moa.d, (U *)mob.d, batch,
                                 FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = moa.sync_out(st);
    if (rc == FHE_OK) rc = mob.sync_out(st);
    return rc;
}

// ---- TFHE key material on the device -----------------------------------------------------
}  // extern "C"
namespace {
inline unsigned long long tdg_blocks(size_t count) { return (count + 3) / 4; }
inline unsigned long long word_blocks(size_t count) { return (count + 7) / 8; }
int sample_tdg_dev(double std_dev, const fhe::ChaChaKey &K, unsigned long long first, u64 *out, size_t count, hipStream_t st) {
    hipLaunchKernelGGL(fhe::sample_tdg_kernel, dim3(grid_for(tdg_blocks(count))), dim3(256), 0, st, out, count, std_dev, K, first);
    return hipGetLastError() == hipSuccess ? FHE_OK : FHE_ERR_HIP;
}
// scheme/tfhe/src/tglwe.rs:91-103 (k = 1) for `rows` ciphertexts on device buffers: a uniform, e <- tdg, b = a s + e + pt
// (pt [pt_rows][n] cycled, or null = encryptions of zero); sk [n], binary
int tglwe_sk_encrypt_dev(const fhe_torus_ctx *t, const u64 *sk, const u64 *pt, size_t pt_rows, u64 *ct_a, u64 *ct_b, int log_n, size_t rows,
                         double std_dev, const fhe::ChaChaKey &K, unsigned long long *cursor, hipStream_t st) {
    const size_t n = size_t(1) << log_n, count = rows * n;
    StreamWs we(count * sizeof(u64), st);
    if (we.rc != FHE_OK) return we.rc;
    u64 *e = we.as<u64>();
    hipLaunchKernelGGL(fhe::sample_u64_kernel, dim3(grid_for(word_blocks(count))), dim3(256), 0, st, ct_a, count, K, *cursor);
    *cursor += word_blocks(count);
    int rc = hipGetLastError() == hipSuccess ? FHE_OK : FHE_ERR_HIP;
    if (rc == FHE_OK) rc = sample_tdg_dev(std_dev, K, *cursor, e, count, st);
    *cursor += tdg_blocks(count);
    if (rc == FHE_OK && hipMemcpyAsync(ct_b, ct_a, count * sizeof(u64), hipMemcpyDeviceToDevice, st) != hipSuccess) rc = FHE_ERR_HIP;
    if (rc == FHE_OK) rc = torus_mul_dev(t, ct_b, sk, 1, log_n, rows, st);  // a binary key: |s_i| <= 1, far inside the exact range
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::torus_add3_kernel, dim3(grid_for(count)), dim3(256), 0, st, ct_b, (const u64 *)e, pt, count, pt ? pt_rows * n : 1);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    return rc;
}
int torus_ring_ok(const fhe_torus_ctx *t, size_t n) {
    if (!t || !is_pow2(n)) return FHE_ERR_INVALID;
    if (t->device < 0) return FHE_ERR_NO_DEVICE;
    const int log_n = ilog2(n);
    return (log_n < 1 || log_n > 15) ? FHE_ERR_UNSUPPORTED : FHE_OK;
}
}  // namespace
extern "C" {

int fhe_sample_tdg(double std_dev, const fhe_rng *rng, uint64_t stream_id, uint64_t *out, size_t count, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    if (!(std_dev >= 0) || (!out && count)) return FHE_ERR_INVALID;
    if (count == 0) return FHE_OK;
    PtrDeviceGuard pguard(out, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    Mirror mo(out, count, mem, false, st);
    if (mo.rc != FHE_OK) return mo.rc;
    int rc = sample_tdg_dev(std_dev, fhe::call_key(rng, stream_id, fhe::RNG_SAMPLE_TDG), 0, mo.d, count, st);
    return rc == FHE_OK ? mo.sync_out(st) : rc;
}

int fhe_sample_binary(const fhe_rng *rng, uint64_t stream_id, uint64_t *out, size_t count, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    if (!out && count) return FHE_ERR_INVALID;
    if (count == 0) return FHE_OK;
    PtrDeviceGuard pguard(out, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    Mirror mo(out, count, mem, false, st);
    if (mo.rc != FHE_OK) return mo.rc;
    hipLaunchKernelGGL(fhe::sample_binary_kernel, dim3(grid_for((count + 511) / 512)), dim3(64), 0, st, mo.d, count, fhe::call_key(rng, stream_id, fhe::RNG_SAMPLE_BINARY), 0ull);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

// scheme/tfhe/src/tlwe.rs:122-132 for `rows` plaintexts: out_a [rows][n] uniform torus, out_b[r] = <a[r], sk> + e + pt[r]
int fhe_tlwe_sk_encrypt(const uint64_t *sk, const uint64_t *pt, size_t n, size_t rows, double std_dev, const fhe_rng *rng, uint64_t stream_id,
                        uint64_t *out_a, uint64_t *out_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    if (!sk || n == 0 || !(std_dev >= 0) || ((!out_a || !out_b) && rows)) return FHE_ERR_INVALID;
    if (rows == 0) return FHE_OK;
    PtrDeviceGuard pguard(out_a, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    Mirror msk(sk, n, mem, true, st), mpt(pt, pt ? rows : 0, mem, true, st), ma(out_a, rows * n, mem, false, st), mb(out_b, rows, mem, false, st);
    if (msk.rc | mpt.rc | ma.rc | mb.rc) return FHE_ERR_HIP;
    StreamWs we(rows * sizeof(u64), st);
    if (we.rc != FHE_OK) return we.rc;
    const fhe::ChaChaKey K = fhe::call_key(rng, stream_id, fhe::RNG_TLWE_ENC);
    hipLaunchKernelGGL(fhe::sample_u64_kernel, dim3(grid_for(word_blocks(rows * n))), dim3(256), 0, st, ma.d, rows * n, K, 0ull);
    int rc = hipGetLastError() == hipSuccess ? FHE_OK : FHE_ERR_HIP;
    if (rc == FHE_OK) rc = sample_tdg_dev(std_dev, K, word_blocks(rows * n), we.as<u64>(), rows, st);
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::tlwe_encrypt_kernel, dim3(grid_for(rows)), dim3(256), 0, st, (const u64 *)ma.d, (const u64 *)msk.d, (const u64 *)we.as<u64>(),
                           pt ? (const u64 *)mpt.d : nullptr, mb.d, n, rows, (const u64 *)nullptr, (size_t)1, 0, 0);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK) rc = ma.sync_out(st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    return rc;
}

// scheme/tfhe/src/tlwe.rs:100-111 `Tlwe::ksk_gen(param, sk0, sk1)`: rows r = j n1 + i encrypt -sk1[i] 2^(rb + j log_b) under sk0:
// ksk_a [n1 d][n0], ksk_b [n1 d], the layout fhe_tlwe_key_switch takes
int fhe_tlwe_ksk_gen(int log_b, int d, const uint64_t *sk0, size_t n0, const uint64_t *sk1, size_t n1, double std_dev, const fhe_rng *rng, uint64_t stream_id, uint64_t *ksk_a, uint64_t *ksk_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    if (log_b < 1 || d < 1 || log_b * d > 64 || !sk0 || !sk1 || n0 == 0 || n1 == 0 || !ksk_a || !ksk_b || !(std_dev >= 0)) return FHE_ERR_INVALID;
    PtrDeviceGuard pguard(ksk_a, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = n1 * d;
    Mirror m0(sk0, n0, mem, true, st), m1(sk1, n1, mem, true, st), ma(ksk_a, rows * n0, mem, false, st), mb(ksk_b, rows, mem, false, st);
    if (m0.rc | m1.rc | ma.rc | mb.rc) return FHE_ERR_HIP;
    StreamWs we(rows * sizeof(u64), st);
    if (we.rc != FHE_OK) return we.rc;
    const fhe::ChaChaKey K = fhe::call_key(rng, stream_id, fhe::RNG_TLWE_KSK);
    hipLaunchKernelGGL(fhe::sample_u64_kernel, dim3(grid_for(word_blocks(rows * n0))), dim3(256), 0, st, ma.d, rows * n0, K, 0ull);
    int rc = hipGetLastError() == hipSuccess ? FHE_OK : FHE_ERR_HIP;
    if (rc == FHE_OK) rc = sample_tdg_dev(std_dev, K, word_blocks(rows * n0), we.as<u64>(), rows, st);
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::tlwe_encrypt_kernel, dim3(grid_for(rows)), dim3(256), 0, st, (const u64 *)ma.d, (const u64 *)m0.d, (const u64 *)we.as<u64>(),
                           (const u64 *)nullptr, mb.d, n0, rows, (const u64 *)m1.d, n1, 64 - log_b * d, log_b);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK) rc = ma.sync_out(st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    return rc;
}

// scheme/tfhe/src/tglwe.rs:91-103 (k = 1): ct_a, ct_b [rows][n]; pt [rows][n] or NULL (zeros); sk [n] binary
int fhe_tglwe_sk_encrypt(const fhe_torus_ctx *t, const uint64_t *sk, const uint64_t *pt, size_t n, size_t rows, double std_dev, const fhe_rng *rng, uint64_t stream_id, uint64_t *ct_a, uint64_t *ct_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rc = torus_ring_ok(t, n);
    if (rc != FHE_OK) return rc;
    if (!sk || !(std_dev >= 0) || ((!ct_a || !ct_b) && rows)) return FHE_ERR_INVALID;
    if (rows == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    Mirror msk(sk, n, mem, true, st), mpt(pt, pt ? rows * n : 0, mem, true, st), ma(ct_a, rows * n, mem, false, st), mb(ct_b, rows * n, mem, false, st);
    if (msk.rc | mpt.rc | ma.rc | mb.rc) return FHE_ERR_HIP;
    unsigned long long cursor = 0;
    rc = tglwe_sk_encrypt_dev(t, msk.d, pt ? mpt.d : nullptr, rows, ma.d, mb.d, ilog2(n), rows, std_dev, fhe::call_key(rng, stream_id, fhe::RNG_TGLWE_ENC), &cursor, st);
    if (rc == FHE_OK) rc = ma.sync_out(st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    return rc;
}

// scheme/tfhe/src/tggsw.rs:73-88 (k = 1) for `count` plaintext polynomials pt [count][n]: rows_a, rows_b [count][2d][n], the
// layout fhe_tggsw_prepare takes
int fhe_tggsw_encrypt(const fhe_torus_ctx *t, int log_b, int d, const uint64_t *sk, const uint64_t *pt, size_t n, size_t count, double std_dev,
                      const fhe_rng *rng, uint64_t stream_id, uint64_t *rows_a, uint64_t *rows_b, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rc = torus_ring_ok(t, n);
    if (rc != FHE_OK) return rc;
    if (log_b < 1 || d < 1 || log_b * d > 64 || !sk || !(std_dev >= 0) || ((!pt || !rows_a || !rows_b) && count)) return FHE_ERR_INVALID;
    if (count == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t rows = count * 2 * d;
    Mirror msk(sk, n, mem, true, st), mpt(pt, count * n, mem, true, st), ma(rows_a, rows * n, mem, false, st), mb(rows_b, rows * n, mem, false, st);
    if (msk.rc | mpt.rc | ma.rc | mb.rc) return FHE_ERR_HIP;
    unsigned long long cursor = 0;
    rc = tglwe_sk_encrypt_dev(t, msk.d, nullptr, 0, ma.d, mb.d, ilog2(n), rows, std_dev, fhe::call_key(rng, stream_id, fhe::RNG_TGGSW_ENC), &cursor, st);
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::tggsw_add_gadget_kernel, dim3(grid_for(count * d * n)), dim3(256), 0, st, ma.d, mb.d, (const u64 *)mpt.d, n, count, d,
                           64 - log_b * d, log_b);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK) rc = ma.sync_out(st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    return rc;
}

}  // extern "C"
