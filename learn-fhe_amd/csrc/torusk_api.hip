// extern "C" entry points for SURVEY.md section 8(a) row T at ANY TGLWE rank k (the reference's `TglweParam::n`): its TGLWE / TGGSW
// code is generic in the rank (scheme/tfhe/src/tglwe.rs:11-35, tggsw.rs:100-121) and its own tests of both run at k = 2, N = 256
// (tglwe.rs:138-166, tggsw.rs:134-181).  torusk_kernels.hpp says how the path is composed; k = 1 through these entries is
// bit-identical to the fused k = 1 entries of torus_api.hip (both are exact).
#include <hip/hip_runtime.h>

#include <new>

#include "torus_ctx.hpp"
#include "torusk_kernels.hpp"
#include "keygen_kernels.hpp"

struct fhe_tggswk_key {
    const fhe_torus_ctx *t = nullptr;
    int k = 0, log_n = 0;
    size_t count = 0;
    u64 *d_eval = nullptr;  // [count][(k + 1) d][k + 1][2 primes][n], evaluation domain
    fhe::TDecomp P{};
};

namespace {

constexpr int MAX_RANK = 8;

int rank_ring_ok(const fhe_torus_ctx *t, int k, size_t n) {
    if (!t || !is_pow2(n) || k < 1) return FHE_ERR_INVALID;
    if (t->device < 0) return FHE_ERR_NO_DEVICE;
    const int log_n = ilog2(n);
    return (log_n < 1 || log_n > 15 || k > MAX_RANK) ? FHE_ERR_UNSUPPORTED : FHE_OK;
}

struct DotWs {  // scratch of one limb-by-key product: limbs [batch][rows][2][n] | sums [batch][cols][2][n]
    u64 *limbs, *sums;
    static size_t words(size_t batch, unsigned rows, unsigned cols, size_t n) { return 2 * n * batch * (size_t(rows) + cols); }
    DotWs(u64 *base, size_t batch, unsigned rows, size_t n) : limbs(base), sums(base + 2 * n * batch * rows) {}
};

struct DotSrc {
    const u64 *src = nullptr, *sub = nullptr, *rot = nullptr;
    size_t rot_stride = 0;
    unsigned polys = 0, src_polys = 0;  // polynomials used / present per ciphertext
};
struct DotDst {
    u64 *out = nullptr;
    unsigned out_polys = 0, out_off = 0;
    const u64 *same = nullptr, *e = nullptr, *pt = nullptr;
    size_t pt_rows = 1;
};

// out <- CRT( sum_r limbs(src)_r * key[r][.] ) (+ addends): the one product every entry below is made of
int limb_dot(const fhe_torus_ctx *t, const fhe::TDecomp &P, const DotSrc &S, const u64 *key_eval, unsigned cols, const DotDst &D, size_t batch, int log_n,
             const DotWs &W, hipStream_t st) {
    const unsigned n = 1u << log_n, rows = S.polys * (P.d > 0 ? (unsigned)P.d : 1u);
    hipLaunchKernelGGL(fhe::torusk_limbs_kernel, dim3(grid_for(size_t(n) * S.polys * batch)), dim3(256), 0, st, S.src, S.sub, W.limbs, n, S.polys, S.src_polys,
                       batch, P, t->T.p0, t->T.p1, S.rot, S.rot_stride);
    HIP_TRY(hipGetLastError());
    int rc = fhe::ntt_fwd_multi(t->d_descs, 2, W.limbs, log_n, 2 * batch * rows, st, 60);
    if (rc != FHE_OK) return rc;
    hipLaunchKernelGGL(fhe::torusk_mac_kernel, dim3(grid_for(size_t(2) * n * cols * batch)), dim3(256), 0, st, (const u64 *)W.limbs, key_eval, W.sums, n, rows,
                       cols, batch, t->T.B0, t->T.B1);
    HIP_TRY(hipGetLastError());
    rc = fhe::ntt_inv_multi(t->d_descs, 2, W.sums, log_n, 2 * batch * cols, st, 60);
    if (rc != FHE_OK) return rc;
    hipLaunchKernelGGL(fhe::torusk_crt_kernel, dim3(grid_for(size_t(n) * cols * batch)), dim3(256), 0, st, (const u64 *)W.sums, D.out, n, cols, D.out_polys,
                       D.out_off, batch, t->T, D.same, D.e, D.pt, D.pt_rows);
    HIP_TRY(hipGetLastError());
    return FHE_OK;
}

// signed polynomials [polys][n] -> evaluations [polys][2][n] (in `out`)
int eval_polys(const fhe_torus_ctx *t, const u64 *in, u64 *out, int log_n, size_t polys, hipStream_t st) {
    const size_t n = size_t(1) << log_n;
    hipLaunchKernelGGL(fhe::torus_residue2_kernel, dim3(grid_for(n * polys)), dim3(256), 0, st, in, out, n, polys, t->T.p0, t->T.p1);
    HIP_TRY(hipGetLastError());
    return fhe::ntt_fwd_multi(t->d_descs, 2, out, log_n, 2 * polys, st, 60);
}

bool key_matches(const fhe_torus_ctx *t, const fhe_tggswk_key *key, size_t index) { return t && key && key->t == t && index < key->count; }
const u64 *key_at(const fhe_tggswk_key *key, size_t index) {
    const size_t k1 = size_t(key->k) + 1;
    return key->d_eval + index * (k1 * key->P.d) * k1 * 2 * (size_t(1) << key->log_n);
}

inline unsigned long long tdg_blocks(size_t count) { return (count + 3) / 4; }
inline unsigned long long word_blocks(size_t count) { return (count + 7) / 8; }

// tglwe.rs:91-103 for `rows` ciphertexts on device buffers: every a_j uniform, e <- tdg, b = sum_j a_j s_j + e + pt
// (pt [pt_rows][n] cycled or null); sk_eval [k][2][n]; ct [rows][k + 1][n]
int tglwek_encrypt_dev(const fhe_torus_ctx *t, int k, const u64 *sk_eval, const u64 *pt, size_t pt_rows, u64 *ct, int log_n, size_t rows, double std_dev,
                       const fhe::ChaChaKey &K, hipStream_t st) {
    const size_t n = size_t(1) << log_n, words = rows * (k + 1) * n;
    StreamWs ws((rows * n + DotWs::words(rows, k, 1, n)) * sizeof(u64), st);
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *e = ws.as<u64>();
    hipLaunchKernelGGL(fhe::sample_u64_kernel, dim3(grid_for(word_blocks(words))), dim3(256), 0, st, ct, words, K, 0ull);  // the b slots are overwritten below
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(fhe::sample_tdg_kernel, dim3(grid_for(tdg_blocks(rows * n))), dim3(256), 0, st, e, rows * n, std_dev, K, word_blocks(words));
    HIP_TRY(hipGetLastError());
    fhe::TDecomp plain{};  // d = 0: the mask polynomials themselves
    DotSrc S; S.src = ct; S.polys = (unsigned)k; S.src_polys = (unsigned)k + 1;
    DotDst D; D.out = ct; D.out_polys = (unsigned)k + 1; D.out_off = (unsigned)k; D.e = e; D.pt = pt; D.pt_rows = pt ? pt_rows : 1;
    return limb_dot(t, plain, S, sk_eval, 1, D, rows, log_n, DotWs(e + rows * n, rows, (unsigned)k, n), st);
}

}  // namespace

extern "C" {

void fhe_tggswk_key_destroy(fhe_tggswk_key *key) {
    if (!key) return;
    if (key->t && key->t->device >= 0 && key->d_eval) {
        DeviceGuard guard(key->t->device);
        (void)hipFree(key->d_eval);
    }
    delete key;
}

// `count` TGGSW ciphertexts of rank k (tggsw.rs:44-88): rows [count][(k + 1) d][k + 1][n] -- (k + 1) d TGLWE ciphertexts each, in the
// order `sk_encrypt` builds them (the d rows carrying the message on a_0, .., on a_{k-1}, then on b)
int fhe_tggswk_prepare(const fhe_torus_ctx *t, int k, int log_b, int d, const uint64_t *rows, size_t n, size_t count, fhe_mem mem, fhe_tggswk_key **out) {
    if (!out) return FHE_ERR_INVALID;
    *out = nullptr;
    int rc = rank_ring_ok(t, k, n);
    if (rc != FHE_OK) return rc;
    if (!rows || count == 0) return FHE_ERR_INVALID;
    fhe::TDecomp P;
    rc = make_tdecomp(log_b, d, &P);
    if (rc != FHE_OK) return rc;
    const int log_n = ilog2(n);
    // exactness: |coefficient| <= (k + 1) d * N * 2^(log_b - 1) * 2^63 must stay below p0 p1 / 2 ~ 2^118.9
    if (ilog2((size_t)(k + 1) * d) + 1 + log_n + 62 + log_b > 118) return FHE_ERR_UNSUPPORTED;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t polys = count * (size_t)(k + 1) * d * (k + 1);
    hipStream_t st = nullptr;
    Mirror mr(rows, polys * n, mem, true, st);
    if (mr.rc != FHE_OK) return mr.rc;
    u64 *eval = nullptr;
    HIP_TRY(hipMalloc((void **)&eval, 2 * polys * n * sizeof(u64)));
    rc = eval_polys(t, mr.d, eval, log_n, polys, st);
    if (rc == FHE_OK && hipStreamSynchronize(st) != hipSuccess) rc = FHE_ERR_HIP;
    fhe_tggswk_key *key = rc == FHE_OK ? new (std::nothrow) fhe_tggswk_key() : nullptr;
    if (!key) { (void)hipFree(eval); return rc != FHE_OK ? rc : FHE_ERR_INVALID; }
    key->t = t; key->k = k; key->log_n = log_n; key->count = count; key->d_eval = eval; key->P = P;
    *out = key;
    return FHE_OK;
}

// tggsw.rs:100-112 `Tggsw::external_product(param, key[index], ct)`, in place on ct [batch][k + 1][n]
int fhe_tggswk_external_product(const fhe_torus_ctx *t, const fhe_tggswk_key *key, size_t index, uint64_t *ct, size_t batch, fhe_mem mem, void *stream) {
    if (!key_matches(t, key, index) || (!ct && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const unsigned k1 = (unsigned)key->k + 1, rows = k1 * (unsigned)key->P.d;
    const size_t n = size_t(1) << key->log_n;
    Mirror mc(ct, n * k1 * batch, mem, true, st);
    if (mc.rc != FHE_OK) return mc.rc;
    StreamWs ws(DotWs::words(batch, rows, k1, n) * sizeof(u64), st);
    if (ws.rc != FHE_OK) return ws.rc;
    DotSrc S; S.src = mc.d; S.polys = S.src_polys = k1;
    DotDst D; D.out = mc.d; D.out_polys = k1;
    int rc = limb_dot(t, key->P, S, key_at(key, index), k1, D, batch, key->log_n, DotWs(ws.as<u64>(), batch, rows, n), st);
    return rc == FHE_OK ? mc.sync_out(st) : rc;
}

// tggsw.rs:114-121 `Tggsw::cmux(key[index], ct0, ct1)` = ct0 + external_product(key[index], ct1 - ct0); [batch][k + 1][n] each, out may
// alias ct0 or ct1
int fhe_tggswk_cmux(const fhe_torus_ctx *t, const fhe_tggswk_key *key, size_t index, const uint64_t *ct0, const uint64_t *ct1, uint64_t *out, size_t batch,
                    fhe_mem mem, void *stream) {
    if (!key_matches(t, key, index) || ((!ct0 || !ct1 || !out) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const unsigned k1 = (unsigned)key->k + 1, rows = k1 * (unsigned)key->P.d;
    const size_t n = size_t(1) << key->log_n, words = n * k1 * batch;
    Mirror m0(ct0, words, mem, true, st), m1(ct1, words, mem, true, st), mo(out, words, mem, false, st);
    if (m0.rc | m1.rc | mo.rc) return FHE_ERR_HIP;
    StreamWs ws(DotWs::words(batch, rows, k1, n) * sizeof(u64), st);
    if (ws.rc != FHE_OK) return ws.rc;
    // the limbs are formed (from both inputs) before anything is written: `out` may be either input
    DotSrc S; S.src = m1.d; S.sub = m0.d; S.polys = S.src_polys = k1;
    DotDst D; D.out = mo.d; D.out_polys = k1; D.same = m0.d;
    int rc = limb_dot(t, key->P, S, key_at(key, index), k1, D, batch, key->log_n, DotWs(ws.as<u64>(), batch, rows, n), st);
    return rc == FHE_OK ? mo.sync_out(st) : rc;
}

// tglwe.rs:61-66 `TglweCiphertext::rotate(i)`: every polynomial times X^i; ct, out [batch][k + 1][n], out != ct
int fhe_tglwek_rotate(const uint64_t *ct, int k, size_t n, int64_t i, uint64_t *out, size_t batch, fhe_mem mem, void *stream) {
    PtrDeviceGuard pguard(ct, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    if (k < 1 || k > MAX_RANK || !is_pow2(n) || (n >> 30) || ((!ct || !out) && batch) || (batch && ct == out)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    const int64_t two_n = 2 * (int64_t)n;
    const unsigned r = (unsigned)(((i % two_n) + two_n) % two_n);
    hipStream_t st = (hipStream_t)stream;
    const size_t polys = batch * (size_t)(k + 1);
    Mirror mi(ct, polys * n, mem, true, st), mo(out, polys * n, mem, false, st);
    if (mi.rc | mo.rc) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::torus_monomial_kernel, dim3(grid_for(polys * n)), dim3(256), 0, st, (const u64 *)mi.d, mo.d, (unsigned)n, polys, r);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

// tglwe.rs:115-127 `Tglwe::sample_extract(ct, index)`: ct [batch][k + 1][n] -> TLWE of dimension k n: out_a [batch][k n], out_b [batch]
int fhe_tglwek_sample_extract(const uint64_t *ct, int k, size_t n, size_t index, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream) {
    PtrDeviceGuard pguard(ct, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    if (k < 1 || k > MAX_RANK || !is_pow2(n) || index >= n || n > (1u << 30) || ((!ct || !out_a || !out_b) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    Mirror mi(ct, n * (k + 1) * batch, mem, true, st), moa(out_a, n * k * batch, mem, false, st), mob(out_b, batch, mem, false, st);
    if (mi.rc | moa.rc | mob.rc) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::tglwek_sample_extract_kernel, dim3(grid_for(n * k * batch)), dim3(256), 0, st, (const u64 *)mi.d, (unsigned)n, (unsigned)k, batch,
                       (unsigned)index, moa.d, mob.d);
    HIP_TRY(hipGetLastError());
    int rc = moa.sync_out(st);
    return rc != FHE_OK ? rc : mob.sync_out(st);
}

// bootstrapping.rs:84-96 `blind_rotate` at rank k: brk = n_lwe TGGSW ciphertexts; a_tilde [batch][n_lwe], b_tilde [batch] mod-switched;
// v [n] the ENCODED test polynomial -> out [batch][k + 1][n].  One CMUX per key ciphertext, each five launches over the whole batch
// (the accumulator lives in `out`).
int fhe_tfhek_blind_rotate(const fhe_torus_ctx *t, const fhe_tggswk_key *brk, const uint64_t *a_tilde, const uint64_t *b_tilde, const uint64_t *v, uint64_t *out,
                           size_t batch, fhe_mem mem, void *stream) {
    if (!key_matches(t, brk, 0) || ((!a_tilde || !b_tilde || !v || !out) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const unsigned k1 = (unsigned)brk->k + 1, rows = k1 * (unsigned)brk->P.d;
    const size_t n = size_t(1) << brk->log_n, n_lwe = brk->count;
    Mirror ma(a_tilde, n_lwe * batch, mem, true, st), mb(b_tilde, batch, mem, true, st), mv(v, n, mem, true, st), mo(out, n * k1 * batch, mem, false, st);
    if (ma.rc | mb.rc | mv.rc | mo.rc) return FHE_ERR_HIP;
    StreamWs ws(DotWs::words(batch, rows, k1, n) * sizeof(u64), st);
    if (ws.rc != FHE_OK) return ws.rc;
    const DotWs W(ws.as<u64>(), batch, rows, n);
    hipLaunchKernelGGL(fhe::torusk_init_acc_kernel, dim3(grid_for(n * k1 * batch)), dim3(256), 0, st, (const u64 *)mv.d, (const u64 *)mb.d, mo.d, (unsigned)n, k1, batch);
    HIP_TRY(hipGetLastError());
    for (size_t i = 0; i < n_lwe; ++i) {  // acc <- acc + brk_i (.) (acc X^{a_i} - acc)   (tggsw.rs:114-121 with ct0 = acc, ct1 = acc.rotate(a_i))
        DotSrc S; S.src = mo.d; S.rot = ma.d + i; S.rot_stride = n_lwe; S.polys = S.src_polys = k1;
        DotDst D; D.out = mo.d; D.out_polys = k1; D.same = mo.d;
        int rc = limb_dot(t, brk->P, S, key_at(brk, i), k1, D, batch, brk->log_n, W, st);
        if (rc != FHE_OK) return rc;
    }
    return mo.sync_out(st);
}

// bootstrapping.rs:78-82 `Bootstrapping::bootstrap` at rank k for `batch` TLWE ciphertexts: mod switch -> blind rotation ->
// sample_extract(0) (a TLWE of dimension k n) -> key switch (ksk_a [k n ks_d][n_lwe], ksk_b [k n ks_d]); lwe_a, out_a [batch][n_lwe]
int fhe_tfhek_bootstrap(const fhe_torus_ctx *t, const fhe_tggswk_key *brk, int ks_log_b, int ks_d, const uint64_t *ksk_a, const uint64_t *ksk_b, const uint64_t *v,
                        const uint64_t *lwe_a, const uint64_t *lwe_b, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream) {
    if (!key_matches(t, brk, 0) || ks_log_b < 1 || ks_d < 1 || !ksk_a || !ksk_b || !v || ((!lwe_a || !lwe_b || !out_a || !out_b) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t k = (size_t)brk->k, n = size_t(1) << brk->log_n, n_lwe = brk->count, ks_rows = k * n * ks_d;
    Mirror mka(ksk_a, ks_rows * n_lwe, mem, true, st), mkb(ksk_b, ks_rows, mem, true, st), mv(v, n, mem, true, st), ma(lwe_a, n_lwe * batch, mem, true, st),
        mb(lwe_b, batch, mem, true, st), moa(out_a, n_lwe * batch, mem, false, st), mob(out_b, batch, mem, false, st);
    if (mka.rc | mkb.rc | mv.rc | ma.rc | mb.rc | moa.rc | mob.rc) return FHE_ERR_HIP;
    StreamWs ws((batch * n_lwe + batch + batch * (k + 1) * n + batch * k * n + batch) * sizeof(u64), st);  // a~ | b~ | acc | extracted a | b
    if (ws.rc != FHE_OK) return ws.rc;
    typedef uint64_t U;
    U *at = ws.as<U>(), *bt = at + batch * n_lwe, *acc = bt + batch, *ea = acc + batch * (k + 1) * n, *eb = ea + batch * k * n;
    int rc = fhe_tfhe_mod_switch((const U *)ma.d, at, batch * n_lwe, n, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = fhe_tfhe_mod_switch((const U *)mb.d, bt, batch, n, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = fhe_tfhek_blind_rotate(t, brk, at, bt, (const U *)mv.d, acc, batch, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = fhe_tglwek_sample_extract(acc, brk->k, n, 0, ea, eb, batch, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = fhe_tlwe_key_switch(ks_log_b, ks_d, (const U *)mka.d, (const U *)mkb.d, ea, eb, k * n, n_lwe, (U *)moa.d, (U *)mob.d, batch, FHE_MEM_DEVICE, stream);
    if (rc == FHE_OK) rc = moa.sync_out(st);
    if (rc == FHE_OK) rc = mob.sync_out(st);
    return rc;
}

// tglwe.rs:91-103 `Tglwe::sk_encrypt` at rank k for `rows` plaintexts: sk [k n] binary (the TLWE key `as_rings` cuts into k rings,
// tglwe.rs:40-44); pt [rows][n] or NULL (zeros); ct [rows][k + 1][n]
int fhe_tglwek_sk_encrypt(const fhe_torus_ctx *t, int k, const uint64_t *sk, const uint64_t *pt, size_t n, size_t rows, double std_dev, const fhe_rng *rng,
                          uint64_t stream_id, uint64_t *ct, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rc = rank_ring_ok(t, k, n);
    if (rc != FHE_OK) return rc;
    if (!sk || !(std_dev >= 0) || (!ct && rows)) return FHE_ERR_INVALID;
    if (rows == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const int log_n = ilog2(n);
    Mirror msk(sk, k * n, mem, true, st), mpt(pt, pt ? rows * n : 0, mem, true, st), mc(ct, rows * (k + 1) * n, mem, false, st);
    if (msk.rc | mpt.rc | mc.rc) return FHE_ERR_HIP;
    StreamWs ske(2 * k * n * sizeof(u64), st);
    if (ske.rc != FHE_OK) return ske.rc;
    rc = eval_polys(t, msk.d, ske.as<u64>(), log_n, k, st);
    if (rc == FHE_OK) rc = tglwek_encrypt_dev(t, k, ske.as<u64>(), pt ? mpt.d : nullptr, rows, mc.d, log_n, rows, std_dev, fhe::call_key(rng, stream_id, fhe::RNG_TGLWEK_ENC), st);
    return rc == FHE_OK ? mc.sync_out(st) : rc;
}

// tggsw.rs:73-88 `Tggsw::sk_encrypt` at rank k for `count` plaintext polynomials pt [count][n]: rows [count][(k + 1) d][k + 1][n], the
// layout fhe_tggswk_prepare takes
int fhe_tggswk_encrypt(const fhe_torus_ctx *t, int k, int log_b, int d, const uint64_t *sk, const uint64_t *pt, size_t n, size_t count, double std_dev,
                       const fhe_rng *rng, uint64_t stream_id, uint64_t *rows, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rc = rank_ring_ok(t, k, n);
    if (rc != FHE_OK) return rc;
    if (log_b < 1 || d < 1 || log_b * d > 64 || !sk || !(std_dev >= 0) || ((!pt || !rows) && count)) return FHE_ERR_INVALID;
    if (count == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(t->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const int log_n = ilog2(n);
    const size_t cts = count * (size_t)(k + 1) * d;
    Mirror msk(sk, k * n, mem, true, st), mpt(pt, count * n, mem, true, st), mr(rows, cts * (k + 1) * n, mem, false, st);
    if (msk.rc | mpt.rc | mr.rc) return FHE_ERR_HIP;
    StreamWs ske(2 * k * n * sizeof(u64), st);
    if (ske.rc != FHE_OK) return ske.rc;
    rc = eval_polys(t, msk.d, ske.as<u64>(), log_n, k, st);
    if (rc == FHE_OK) rc = tglwek_encrypt_dev(t, k, ske.as<u64>(), nullptr, 0, mr.d, log_n, cts, std_dev, fhe::call_key(rng, stream_id, fhe::RNG_TGGSWK_ENC), st);
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::tggswk_add_gadget_kernel, dim3(grid_for(count * (k + 1) * d * n)), dim3(256), 0, st, mr.d, (const u64 *)mpt.d, (unsigned)n,
                           (unsigned)k + 1, count, d, 64 - log_b * d, log_b);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    return rc == FHE_OK ? mr.sync_out(st) : rc;
}

}  // extern "C"
