// extern "C" entry points for SURVEY.md section 8(a) row a14: RNS base extension, rescale_k, CKKS key switch.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <set>
#include <vector>

#include "api_common.hpp"
#include "ctx.hpp"
#include "rns_kernels.hpp"
#include "keygen_kernels.hpp"

struct fhe_rns_ctx {
    int L = 0, K = 0, device = -1;
    std::vector<uint64_t> qs, ps;
    std::vector<fhe_ctx *> mods;       // L + K per-prime transform contexts (qs then ps)
    fhe::ModDesc *d_descs = nullptr;   // [L + K]
    fhe::Barrett *d_barrett = nullptr; // [L + K]
    void *d_blob = nullptr;            // all conversion tables
    fhe::BaseConv q2p{}, p2q{};
    fhe::RescaleConsts resc{};
    fhe::RescaleConsts resc_last{};    // `rescale()` = rescale_k(1) of a polynomial over qs: drops q_{L-1} (L >= 2)
    int max_log_n = 0;                 // largest ring degree every prime supports
    int all_pm = -1;                   // common pseudo-Mersenne bit length of all primes, 0 if none
    // every modulus a pseudo-Mersenne prime of ONE bit length: the conversions run as unreduced dot products (rns_kernels.hpp)
    bool pm = false;
    fhe::pd::Uni uni{};
    fhe::PmSrc s_q2p{}, s_p2q{}, s_p2q_sum{}, s_p2q_diff{}, s_last{}, s_p2q_plain{};
    fhe::PmRows r_q2p{}, r_q2p_w{}, r_resc{}, r_resc_edge{}, r_last{}, r_p2q_plain{};
};

struct fhe_ckks_key {
    const fhe_rns_ctx *rns = nullptr;
    int log_n = 0;
    u64 *d_kb = nullptr, *d_ka = nullptr;  // [L + K][n], evaluation domain
};

// one device's part of a limb-sharded key switch (include/fhe_ring.h: fhe_ckks_shard_create)
struct fhe_ckks_shard {
    const fhe_rns_ctx *rns = nullptr;
    int log_n = 0, q_lo = 0, q_hi = 0, p_lo = 0, p_hi = 0;
    bool edge = false;                 // N = 2^15: the transforms' outermost layer runs inside the extend / rescale kernels
    fhe::ModDesc *d_descs = nullptr;   // [nq + np]: the owned q-limbs, then the owned p-limbs
    u64 *d_key = nullptr;              // [2][nq + np][n]: ksk.b then ksk.a on the owned limbs, evaluation domain
};

namespace {

using fhe::mulmod;

uint64_t prod_mod(const std::vector<uint64_t> &xs, int skip, uint64_t m) {
    uint64_t r = 1 % m;
    for (int i = 0; i < (int)xs.size(); ++i)
        if (i != skip) r = mulmod(r, xs[i] % m, m);
    return r;
}

// lays the tables of one conversion A -> B into `blob` (host image), returns the device-pointer view
struct BlobBuilder {
    std::vector<uint64_t> words;  // doubles are bit-cast into words
    size_t put(const std::vector<uint64_t> &v) {
        size_t off = words.size();
        words.insert(words.end(), v.begin(), v.end());
        return off;
    }
    size_t put_f64(const std::vector<double> &v) {
        size_t off = words.size();
        for (double d : v) { uint64_t w; std::memcpy(&w, &d, 8); words.push_back(w); }
        return off;
    }
};

struct ConvOffsets { size_t a_mod, ahat_inv, ahat_inv_s, frac, b_mod, c, c_s, ua; int la, lb; };

ConvOffsets build_conv(BlobBuilder &bb, const std::vector<uint64_t> &A, const std::vector<uint64_t> &B) {
    const int la = (int)A.size(), lb = (int)B.size();
    std::vector<uint64_t> inv(la), inv_s(la), c(size_t(lb) * la), c_s(size_t(lb) * la), ua(size_t(lb) * (la + 1));
    std::vector<double> frac(la);
    for (int i = 0; i < la; ++i) {
        inv[i] = fhe::invmod(prod_mod(A, i, A[i]), A[i]);  // rns.rs:290-293
        inv_s[i] = fhe::shoup(inv[i], A[i]);
        frac[i] = 1.0 / (double)A[i];                      // rns.rs:294
    }
    for (int j = 0; j < lb; ++j) {
        for (int i = 0; i < la; ++i) {
            c[size_t(j) * la + i] = prod_mod(A, i, B[j]);  // rns.rs:305-313
            c_s[size_t(j) * la + i] = fhe::shoup(c[size_t(j) * la + i], B[j]);
        }
        const uint64_t amod = prod_mod(A, -1, B[j]);
        for (int u = 0; u <= la; ++u) ua[size_t(j) * (la + 1) + u] = mulmod(amod, (uint64_t)u % B[j], B[j]);  // rns.rs:315-320
    }
    ConvOffsets o;
    o.la = la; o.lb = lb;
    o.a_mod = bb.put(A); o.ahat_inv = bb.put(inv); o.ahat_inv_s = bb.put(inv_s); o.frac = bb.put_f64(frac);
    o.b_mod = bb.put(B); o.c = bb.put(c); o.c_s = bb.put(c_s); o.ua = bb.put(ua);
    return o;
}

fhe::BaseConv conv_view(const ConvOffsets &o, const uint64_t *base) {
    fhe::BaseConv C;
    C.la = o.la; C.lb = o.lb;
    C.a_mod = (const u64 *)base + o.a_mod; C.ahat_inv = (const u64 *)base + o.ahat_inv; C.ahat_inv_s = (const u64 *)base + o.ahat_inv_s;
    C.frac = (const double *)(base + o.frac);
    C.b_mod = (const u64 *)base + o.b_mod; C.c = (const u64 *)base + o.c; C.c_s = (const u64 *)base + o.c_s; C.ua = (const u64 *)base + o.ua;
    return C;
}

// ---- tables of the pseudo-Mersenne route (rns_kernels.hpp: PmSrc / PmRows; pm_dot.hpp) -----------------------------------------
// everything is laid into the same blob; 32-bit tables two per word, 16-byte tables on 16-byte boundaries
size_t put_u32(BlobBuilder &bb, const std::vector<uint32_t> &v) {
    std::vector<uint64_t> w((v.size() + 1) / 2, 0);
    for (size_t i = 0; i < v.size(); ++i) w[i / 2] |= (uint64_t)v[i] << (32 * (i & 1));
    return bb.put(w);
}
// {a0, a1, b0, b1} of the two-operand form (arith.hpp ArithDS) for a B-bit modulus q
void push_ds(std::vector<uint32_t> &out, uint64_t w, uint64_t q, int B) {
    const uint64_t w1 = (uint64_t)((((fhe::u128)w) << 32) % q), lo = (uint64_t(1) << (B - 31)) - 1;
    out.push_back((uint32_t)(w & lo)); out.push_back((uint32_t)(w >> (B - 31)));
    out.push_back((uint32_t)(w1 & lo)); out.push_back((uint32_t)(w1 >> (B - 31)));
}
// source-side records (rns_kernels.hpp PmSrc): vs_i = (v_i mult_i + hk_i) mod a_i; w_i = tw[1] of a_i (or 0)
size_t build_src(BlobBuilder &bb, const std::vector<uint64_t> &A, const std::vector<uint64_t> &mult, const std::vector<uint64_t> &hk,
                 const std::vector<uint64_t> &w, int B) {
    std::vector<uint32_t> rec;
    for (int i = 0; i < (int)A.size(); ++i) {
        const double frac = 1.0 / (double)A[i];  // rns.rs:294
        uint64_t fbits;
        std::memcpy(&fbits, &frac, 8);
        rec.push_back((uint32_t)A[i]); rec.push_back((uint32_t)(A[i] >> 32));
        rec.push_back((uint32_t)((uint64_t(1) << B) - A[i])); rec.push_back(0);
        push_ds(rec, mult[i], A[i], B);
        rec.push_back((uint32_t)hk[i]); rec.push_back((uint32_t)(hk[i] >> 32));
        rec.push_back((uint32_t)fbits); rec.push_back((uint32_t)(fbits >> 32));
        push_ds(rec, w[i], A[i], B);
    }
    if (bb.words.size() & 7) bb.words.resize((bb.words.size() + 7) & ~size_t(7), 0);  // 64-byte records on 64-byte boundaries
    return put_u32(bb, rec);
}
fhe::PmSrc src_view(size_t off, int la, const uint64_t *base) { return fhe::PmSrc{la, (const unsigned *)(base + off)}; }

struct RowOffsets { size_t tab; int stride; };
// rows over the moduli Bm (rns_kernels.hpp PmRows): M[j][i] (la columns), the multiplier U[j] of u, the multipliers X[j] / X2[j] of the
// limb's own value (pair sum / pair difference), the constant term KC[j]
RowOffsets build_rows(BlobBuilder &bb, const std::vector<uint64_t> &Bm, int la, const std::vector<uint64_t> &M, const std::vector<uint64_t> &Uv,
                      const std::vector<uint64_t> &X, const std::vector<uint64_t> &X2, const std::vector<uint64_t> &KC, int B) {
    const int rows = (int)Bm.size(), stride = (la + 7) / 8 * 8, row_dw = 3 * stride + fhe::PM_ROW_TAIL;
    const uint32_t m30 = (1u << 30) - 1;
    std::vector<uint32_t> tab(size_t(rows) * row_dw, 0);
    for (int j = 0; j < rows; ++j) {
        uint32_t *t = tab.data() + size_t(j) * row_dw;
        for (int i = 0; i < la; ++i) {
            const uint64_t m = M[size_t(j) * la + i];
            t[i] = (uint32_t)m & m30; t[stride + i] = (uint32_t)(m >> 30); t[2 * stride + i] = t[i] + t[stride + i];
        }
        uint32_t *k = t + 3 * stride;
        const uint64_t c = (uint64_t(1) << B) - Bm[j];
        k[0] = (uint32_t)Uv[j] & m30; k[1] = (uint32_t)(Uv[j] >> 30);
        k[2] = (uint32_t)X[j] & m30; k[3] = (uint32_t)(X[j] >> 30); k[4] = (uint32_t)X2[j] & m30; k[5] = (uint32_t)(X2[j] >> 30);
        k[6] = (uint32_t)c; k[7] = (uint32_t)(c << (60 - B));
        k[8] = (uint32_t)Bm[j]; k[9] = (uint32_t)(Bm[j] >> 32); k[10] = (uint32_t)KC[j]; k[11] = (uint32_t)(KC[j] >> 32);
    }
    if (bb.words.size() & 7) bb.words.resize((bb.words.size() + 7) & ~size_t(7), 0);
    return RowOffsets{put_u32(bb, tab), stride};
}
fhe::PmRows rows_view(const RowOffsets &o, const uint64_t *base) { return fhe::PmRows{o.stride, (const unsigned *)(base + o.tab)}; }

inline unsigned grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b ? b : 1));
}

}  // namespace

extern "C" {

void fhe_rns_ctx_destroy(fhe_rns_ctx *r) {
    if (!r) return;
    if (r->device >= 0) {
        DeviceGuard guard(r->device);
        if (r->d_descs) (void)hipFree(r->d_descs);
        if (r->d_barrett) (void)hipFree(r->d_barrett);
        if (r->d_blob) (void)hipFree(r->d_blob);
    }
    for (fhe_ctx *c : r->mods) fhe_ctx_destroy(c);
    delete r;
}

int fhe_rns_ctx_create(const uint64_t *qs, int L, const uint64_t *ps, int K, int device, fhe_rns_ctx **out) {
    if (!out) return FHE_ERR_INVALID;
    *out = nullptr;
    if (!qs || !ps || L < 1 || K < 1 || L > fhe::RNS_MAX_LIMBS || K > fhe::RNS_MAX_LIMBS) return FHE_ERR_INVALID;
    std::set<uint64_t> uniq(qs, qs + L);
    uniq.insert(ps, ps + K);
    if ((int)uniq.size() != L + K) return FHE_ERR_INVALID;  // `assert!(... all_unique())`, rns.rs:25, 84
    fhe_rns_ctx *r = new (std::nothrow) fhe_rns_ctx();
    if (!r) return FHE_ERR_INVALID;
    r->L = L; r->K = K; r->device = device;
    r->qs.assign(qs, qs + L);
    r->ps.assign(ps, ps + K);
    r->max_log_n = 64;
    for (int i = 0; i < L + K; ++i) {
        fhe_ctx *c = nullptr;
        int rc = fhe_ctx_create(i < L ? qs[i] : ps[i - L], device, &c);
        if (rc != FHE_OK) { fhe_rns_ctx_destroy(r); return rc; }
        r->mods.push_back(c);
        if (c->log_cap < r->max_log_n) r->max_log_n = c->log_cap;
        if (r->all_pm == -1) r->all_pm = c->pm_b;
        else if (r->all_pm != c->pm_b) r->all_pm = 0;
    }
    if (device < 0) { *out = r; return FHE_OK; }
    DeviceGuard guard(device);
    if (!guard.ok) { fhe_rns_ctx_destroy(r); return FHE_ERR_HIP; }
    // tables
    BlobBuilder bb;
    ConvOffsets oq2p = build_conv(bb, r->qs, r->ps), op2q = build_conv(bb, r->ps, r->qs);
    std::vector<uint64_t> half_q(L), half_p(K), pinv(L), pinv_s(L), red_mu(L);
    auto half_of_p = [&](uint64_t m) {  // floor(P/2) mod m = (P mod m - 1) * 2^-1 mod m   (P odd, m odd)
        return mulmod(fhe::submod(prod_mod(r->ps, -1, m), 1 % m, m), ((m + 1) / 2) % m, m);
    };
    for (int i = 0; i < L; ++i) {
        half_q[i] = half_of_p(qs[i]);
        pinv[i] = fhe::invmod(prod_mod(r->ps, -1, qs[i]), qs[i]);
        pinv_s[i] = fhe::shoup(pinv[i], qs[i]);
        red_mu[i] = (uint64_t)((((fhe::u128)1) << 64) / qs[i]);
    }
    for (int j = 0; j < K; ++j) half_p[j] = half_of_p(ps[j]);
    const size_t o_hq = bb.put(half_q), o_hp = bb.put(half_p), o_pi = bb.put(pinv), o_pis = bb.put(pinv_s), o_mu = bb.put(red_mu);
    // rns.rs:99-101 `rescale()`: the same formulas with P = the last q-limb (rns.rs:104-111, the K == 1 branch)
    std::vector<uint64_t> lhalf_q(L), lhalf_p(1), lpinv(L), lpinv_s(L);
    if (L >= 2) {
        const uint64_t pl = qs[L - 1];
        lhalf_p[0] = pl >> 1;
        for (int i = 0; i + 1 < L; ++i) {
            lhalf_q[i] = (pl >> 1) % qs[i];
            lpinv[i] = fhe::invmod(pl % qs[i], qs[i]);
            lpinv_s[i] = fhe::shoup(lpinv[i], qs[i]);
        }
    }
    const size_t o_lhq = bb.put(lhalf_q), o_lhp = bb.put(lhalf_p), o_lpi = bb.put(lpinv), o_lpis = bb.put(lpinv_s);
    // ---- the pseudo-Mersenne route (rns_kernels.hpp): every linear step folded into the constants of unreduced dot products ----
    r->pm = r->all_pm >= 34 && r->all_pm <= 60;
    size_t so_q2p = 0, so_p2q = 0, so_sum = 0, so_diff = 0, so_last = 0, so_plain = 0;
    RowOffsets ro_q2p{}, ro_q2p_w{}, ro_resc{}, ro_edge{}, ro_last{}, ro_plain{};
    if (r->pm) {
        const int B = r->all_pm;
        const std::vector<uint64_t> &Q = r->qs, &P = r->ps;
        auto hat_inv = [&](const std::vector<uint64_t> &A) {  // (A / a_i)^-1 mod a_i (rns.rs:290-293)
            std::vector<uint64_t> v(A.size());
            for (int i = 0; i < (int)A.size(); ++i) v[i] = fhe::invmod(prod_mod(A, i, A[i]), A[i]);
            return v;
        };
        // tw[1], n^-1 and n^-1 twi[1] at n = 2^15 of modulus i of qs ++ ps (the edge kernels; zero where the modulus has no such root)
        auto tw1 = [&](int i) { const fhe_ctx *c = r->mods[i]; return c->tw.size() > 1 ? c->tw[1] : 0; };
        auto nv = [&](int i) { return r->mods[i]->ninv[15]; };
        auto nw = [&](int i) { return r->mods[i]->ninv_w[15]; };
        // extend Q -> P: out_j = sum_i (Q/q_i mod p_j) vs_i + u (p_j - Q mod p_j)
        {
            const std::vector<uint64_t> inv = hat_inv(Q), zero(L, 0);
            std::vector<uint64_t> fw(L);
            for (int i = 0; i < L; ++i) fw[i] = tw1(i);
            so_q2p = build_src(bb, Q, inv, zero, fw, B);
            std::vector<uint64_t> M(size_t(K) * L), MW(size_t(K) * L), Uv(K), UW(K), z(K, 0);
            for (int j = 0; j < K; ++j) {
                const uint64_t b = P[j], w = tw1(L + j);
                for (int i = 0; i < L; ++i) {
                    M[size_t(j) * L + i] = prod_mod(Q, i, b);
                    MW[size_t(j) * L + i] = mulmod(M[size_t(j) * L + i], w, b);
                }
                Uv[j] = (b - prod_mod(Q, -1, b)) % b;
                UW[j] = mulmod(Uv[j], w, b);
            }
            ro_q2p = build_rows(bb, P, L, M, Uv, z, z, z, B);
            ro_q2p_w = build_rows(bb, P, L, MW, UW, z, z, z, B);
        }
        // rescale_k(K): vs_j = (v_j + half_j) inv_j; out_l = sum_j (-(P/p_j) P^-1) vs_j + u + x P^-1 + half_l P^-1  (mod q_l)
        {
            const std::vector<uint64_t> inv = hat_inv(P);
            std::vector<uint64_t> hk(K), inv_sum(K), inv_diff(K);
            for (int j = 0; j < K; ++j) {
                hk[j] = mulmod(half_p[j], inv[j], P[j]);
                inv_sum[j] = mulmod(nv(L + j), inv[j], P[j]);
                inv_diff[j] = mulmod(nw(L + j), inv[j], P[j]);
            }
            const std::vector<uint64_t> zk(K, 0);
            so_p2q = build_src(bb, P, inv, hk, zk, B);
            so_sum = build_src(bb, P, inv_sum, hk, zk, B);
            so_diff = build_src(bb, P, inv_diff, hk, zk, B);
            std::vector<uint64_t> M(size_t(L) * K), one(L, 1), X(L), XS(L), XD(L), KC(L), z(L, 0);
            for (int l = 0; l < L; ++l) {
                const uint64_t q = Q[l];
                for (int j = 0; j < K; ++j) M[size_t(l) * K + j] = mulmod((q - prod_mod(P, j, q)) % q, pinv[l], q);
                X[l] = pinv[l];
                XS[l] = mulmod(nv(l), pinv[l], q);
                XD[l] = mulmod(nw(l), pinv[l], q);
                KC[l] = mulmod(half_q[l], pinv[l], q);
            }
            ro_resc = build_rows(bb, Q, K, M, one, X, z, KC, B);
            ro_edge = build_rows(bb, Q, K, M, one, XS, XD, KC, B);
            // switch_bases P -> Q on its own (rns.rs:93-97): out_l = sum_j (P/p_j mod q_l) vs_j + u (q_l - P mod q_l)
            so_plain = build_src(bb, P, inv, zk, zk, B);
            std::vector<uint64_t> MP(size_t(L) * K), UP(L);
            for (int l = 0; l < L; ++l) {
                for (int j = 0; j < K; ++j) MP[size_t(l) * K + j] = prod_mod(P, j, Q[l]);
                UP[l] = (Q[l] - prod_mod(P, -1, Q[l])) % Q[l];
            }
            ro_plain = build_rows(bb, Q, K, MP, UP, z, z, z, B);
        }
        // rescale(): P = the last q-limb, K == 1 (no correction term)
        if (L >= 2) {
            const std::vector<uint64_t> last(1, Q[L - 1]), one1(1, 1), hk1(1, Q[L - 1] >> 1);
            so_last = build_src(bb, last, one1, hk1, std::vector<uint64_t>(1, 0), B);
            const std::vector<uint64_t> Ql(Q.begin(), Q.end() - 1);
            std::vector<uint64_t> M(L - 1), X(L - 1), KC(L - 1), z(L - 1, 0);
            for (int l = 0; l + 1 < L; ++l) {
                M[l] = (Ql[l] - lpinv[l]) % Ql[l];
                X[l] = lpinv[l];
                KC[l] = mulmod(lhalf_q[l], lpinv[l], Ql[l]);
            }
            ro_last = build_rows(bb, Ql, 1, M, z, X, z, KC, B);
        }
    }
    std::vector<fhe::ModDesc> descs(L + K);
    std::vector<fhe::Barrett> bar(L + K);
    for (int i = 0; i < L + K; ++i) { descs[i] = r->mods[i]->h_desc; bar[i] = r->mods[i]->barrett; }
    hipError_t e = hipMalloc(&r->d_blob, bb.words.size() * 8);
    if (e == hipSuccess) e = hipMemcpy(r->d_blob, bb.words.data(), bb.words.size() * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&r->d_descs, descs.size() * sizeof(fhe::ModDesc));
    if (e == hipSuccess) e = hipMemcpy(r->d_descs, descs.data(), descs.size() * sizeof(fhe::ModDesc), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&r->d_barrett, bar.size() * sizeof(fhe::Barrett));
    if (e == hipSuccess) e = hipMemcpy(r->d_barrett, bar.data(), bar.size() * sizeof(fhe::Barrett), hipMemcpyHostToDevice);
    if (e != hipSuccess) { g_last_hip = (int)e; fhe_rns_ctx_destroy(r); return FHE_ERR_HIP; }
    const uint64_t *base = (const uint64_t *)r->d_blob;
    r->q2p = conv_view(oq2p, base);
    r->p2q = conv_view(op2q, base);
    r->resc.L = L; r->resc.K = K;
    r->resc.q_mod = r->q2p.a_mod; r->resc.p_mod = r->q2p.b_mod;
    r->resc.half_q = (const u64 *)base + o_hq; r->resc.half_p = (const u64 *)base + o_hp;
    r->resc.pinv = (const u64 *)base + o_pi; r->resc.pinv_s = (const u64 *)base + o_pis;
    r->resc.red_mu = (const u64 *)base + o_mu;
    r->resc.p2q = r->p2q;
    r->resc_last = r->resc;  // p2q unused when K == 1
    r->resc_last.L = L - 1; r->resc_last.K = 1;
    r->resc_last.p_mod = r->q2p.a_mod + (L - 1);
    r->resc_last.half_q = (const u64 *)base + o_lhq; r->resc_last.half_p = (const u64 *)base + o_lhp;
    r->resc_last.pinv = (const u64 *)base + o_lpi; r->resc_last.pinv_s = (const u64 *)base + o_lpis;
    if (r->pm) {
        r->uni = fhe::pd::make_uni(r->all_pm);
        r->s_q2p = src_view(so_q2p, L, base); r->s_p2q = src_view(so_p2q, K, base);
        r->s_p2q_sum = src_view(so_sum, K, base); r->s_p2q_diff = src_view(so_diff, K, base);
        r->r_q2p = rows_view(ro_q2p, base); r->r_q2p_w = rows_view(ro_q2p_w, base);
        r->r_resc = rows_view(ro_resc, base); r->r_resc_edge = rows_view(ro_edge, base);
        r->s_p2q_plain = src_view(so_plain, K, base); r->r_p2q_plain = rows_view(ro_plain, base);
        if (L >= 2) { r->s_last = src_view(so_last, 1, base); r->r_last = rows_view(ro_last, base); }
    }
    *out = r;
    return FHE_OK;
}

namespace {
// the RNS kernels keep their limb vectors in registers: instantiate for the smallest bound that holds the source base
#define RNS_BOUND(la, CALL)                                              \
    do {                                                                 \
        if ((la) == 8) { CALL(8, true); } /* the BASELINE shape: no limb predicates at all */ \
        else if ((la) == 1) { CALL(1, true); } /* `rescale()` and K = 1 */ \
        else if ((la) <= 4) { CALL(4, false); }                          \
        else if ((la) <= 8) { CALL(8, false); }                          \
        else if ((la) <= 16) { CALL(16, false); }                        \
        else { CALL(32, false); }                                        \
    } while (0)
// grid of the limb-wise products: x over the coefficients of one polynomial, y over (ciphertext, limb) pairs
struct PointwiseGrid {
    dim3 g;
    PointwiseGrid(size_t n, size_t polys) {
        const size_t gx = (n + 255) / 256;
        g = dim3((unsigned)(gx > 64 ? 64 : gx), (unsigned)(polys > 65535 ? 65535 : polys));
    }
};
// rows [lo, ..) of a table of output rows (a limb-sharded key switch computes only the limbs it owns)
fhe::PmRows rows_from(const fhe::PmRows &R, int lo) { return fhe::PmRows{R.stride, R.tab + size_t(lo) * (3 * R.stride + fhe::PM_ROW_TAIL)}; }

// the new limbs of extend_bases: qs -> ps (in [batch][L][n] -> out [batch][K][n]) or, `to_qs`, ps -> qs; `copy` (optional) receives the
// source limbs [c_lo, c_hi); only the target rows [r_lo, r_hi) are produced (default: all)
void launch_extend(const fhe_rns_ctx *r, const u64 *in, size_t in_bs, u64 *out, size_t out_bs, size_t n, size_t batch, hipStream_t st,
                   u64 *copy = nullptr, size_t copy_bs = 0, bool to_qs = false, int c_lo = 0, int c_hi = -1, int r_lo = 0, int r_hi = -1) {
    const dim3 grid(grid_for(n * batch));
    const int la = to_qs ? r->K : r->L, lb = to_qs ? r->L : r->K;
    if (c_hi < 0) c_hi = la;
    if (r_hi < 0) r_hi = lb;
    if (r->pm) {
        const fhe::PmSrc &S = to_qs ? r->s_p2q_plain : r->s_q2p;
        const fhe::PmRows R = rows_from(to_qs ? r->r_p2q_plain : r->r_q2p, r_lo);
#define CALL(M, F) hipLaunchKernelGGL((fhe::rns_extend_pm_kernel<M, F>), grid, dim3(256), 0, st, in, in_bs, out, out_bs, n, batch, S, R, r_hi - r_lo, r->uni, copy, copy_bs, c_lo, c_hi)
        RNS_BOUND(la, CALL);
#undef CALL
    } else {  // (the Shoup route has no limb subsets: callers check r->pm first)
        const fhe::BaseConv &C = to_qs ? r->p2q : r->q2p;
#define CALL(M, F) hipLaunchKernelGGL((fhe::rns_extend_kernel<M, F>), grid, dim3(256), 0, st, in, in_bs, out, out_bs, n, batch, C, copy, copy_bs)
        RNS_BOUND(la, CALL);
#undef CALL
    }
}
// where a whole [batch][L+K][n] block keeps the limbs of a rescale (rns_kernels.hpp RescaleIn)
fhe::RescaleIn rescale_in_block(const u64 *in, size_t in_bs, int L, int K, size_t n) {
    fhe::RescaleIn I{};
    I.in_q = in; I.q_bs = in_bs; I.in_p = in + size_t(L) * n; I.p_bs = in_bs;
    for (int j = 0; j < K; ++j) I.off[j] = size_t(j) * n;
    return I;
}
// rescale_k(K) (last = false: L q-limbs + K p-limbs -> out [batch][L][n]) or rescale() (last = true: in [batch][L][n] -> [batch][L-1][n]);
// rows [q_lo, q_hi) of the q-limbs only (pseudo-Mersenne route; default: all), I.in_q then holds those limbs
void launch_rescale(const fhe_rns_ctx *r, bool last, const fhe::RescaleIn &I, u64 *out, size_t out_bs, const u64 *addend, size_t add_bs, size_t n,
                    size_t batch, hipStream_t st, int q_lo = 0, int q_hi = -1) {
    const dim3 grid(grid_for(n * batch));
    const int k = last ? 1 : r->K, L = last ? r->L - 1 : r->L;
    if (q_hi < 0) q_hi = L;
    if (r->pm) {
        const fhe::PmSrc &S = last ? r->s_last : r->s_p2q;
        const fhe::PmRows R = rows_from(last ? r->r_last : r->r_resc, q_lo);
#define CALL(M, F) hipLaunchKernelGGL((fhe::rns_rescale_pm_kernel<M, F>), grid, dim3(256), 0, st, I, out, out_bs, addend, add_bs, n, batch, q_hi - q_lo, S, R, r->uni)
        RNS_BOUND(k, CALL);
#undef CALL
    } else {
        const fhe::RescaleConsts &R = last ? r->resc_last : r->resc;
#define CALL(M, F) hipLaunchKernelGGL((fhe::rns_rescale_kernel<M, F>), grid, dim3(256), 0, st, I.in_q, I.q_bs, out, out_bs, addend, add_bs, n, batch, R)
        RNS_BOUND(k, CALL);
#undef CALL
    }
}
// the same two steps with the outermost transform layer of a 2^15 ring in them (rns_kernels.hpp); pseudo-Mersenne bases only
void launch_extend_edge(const fhe_rns_ctx *r, const u64 *in, size_t in_bs, u64 *out, size_t out_bs, size_t n, size_t batch, hipStream_t st,
                        int c_lo = 0, int c_hi = -1, int r_lo = 0, int r_hi = -1) {
    if (c_hi < 0) c_hi = r->L;
    if (r_hi < 0) r_hi = r->K;
    const fhe::PmRows R = rows_from(r->r_q2p, r_lo), RW = rows_from(r->r_q2p_w, r_lo);
#define CALL(M, F) hipLaunchKernelGGL((fhe::rns_extend_edge_pm_kernel<M, F>), dim3(grid_for(n / 2 * batch)), dim3(256), 0, st, in, in_bs, out, out_bs, n, batch, \
                                      r->s_q2p, R, RW, r_hi - r_lo, r->uni, c_lo, c_hi)
    RNS_BOUND(r->L, CALL);
#undef CALL
}
void launch_rescale_edge(const fhe_rns_ctx *r, const fhe::RescaleIn &I, u64 *out, size_t out_bs, const u64 *addend, size_t add_bs, size_t n,
                         size_t batch, hipStream_t st, int q_lo = 0, int q_hi = -1) {
    if (q_hi < 0) q_hi = r->L;
    const fhe::PmRows R = rows_from(r->r_resc_edge, q_lo);
#define CALL(M, F) hipLaunchKernelGGL((fhe::rns_rescale_edge_pm_kernel<M, F>), dim3(grid_for(n / 2 * batch)), dim3(256), 0, st, I, out, out_bs, addend, \
                                      add_bs, n, batch, q_hi - q_lo, r->s_p2q_sum, r->s_p2q_diff, R, r->uni)
    RNS_BOUND(r->K, CALL);
#undef CALL
}
// lab switch NO_EDGE: the key switch keeps whole 2^15 transforms (A/B runs and the test that both routes agree bit for bit)
bool edge_enabled() { return fhe::opt(fhe::OPT_NO_EDGE) == 0; }
}  // namespace

int fhe_rns_extend_bases(const fhe_rns_ctx *r, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem, void *stream) {
    if (!r || ((!in || !out) && n * batch)) return FHE_ERR_INVALID;
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    if (n * batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    Mirror mi(in, n * batch * r->L, mem, true, st), mo(out, n * batch * r->K, mem, false, st);
    if (mi.rc | mo.rc) return FHE_ERR_HIP;
    launch_extend(r, mi.d, size_t(r->L) * n, mo.d, size_t(r->K) * n, n, batch, st);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

// util/src/ring/rns.rs:93-97 `RnsRq::switch_bases`: the same polynomial over the OTHER base (extend_bases, old limbs dropped).
// to_qs = 0: in [batch][L][n] over qs -> out [batch][K][n] over ps; to_qs != 0: in [batch][K][n] over ps -> out [batch][L][n] over qs
int fhe_rns_switch_bases(const fhe_rns_ctx *r, int to_qs, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem, void *stream) {
    if (!r || ((!in || !out) && n * batch)) return FHE_ERR_INVALID;
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    if (n * batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t la = to_qs ? r->K : r->L, lb = to_qs ? r->L : r->K;
    Mirror mi(in, n * batch * la, mem, true, st), mo(out, n * batch * lb, mem, false, st);
    if (mi.rc | mo.rc) return FHE_ERR_HIP;
    launch_extend(r, mi.d, la * n, mo.d, lb * n, n, batch, st, nullptr, 0, to_qs != 0);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

int fhe_rns_rescale_k(const fhe_rns_ctx *r, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem, void *stream) {
    if (!r || ((!in || !out) && n * batch)) return FHE_ERR_INVALID;
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    if (n * batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t lk = size_t(r->L + r->K);
    Mirror mi(in, n * batch * lk, mem, true, st), mo(out, n * batch * r->L, mem, false, st);
    if (mi.rc | mo.rc) return FHE_ERR_HIP;
    launch_rescale(r, false, rescale_in_block(mi.d, lk * n, r->L, r->K, n), mo.d, size_t(r->L) * n, nullptr, 0, n, batch, st);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

namespace {
// `extended` = 0: polynomials over qs ([batch][L][n]); != 0: over qs ++ ps ([batch][L+K][n])
int rns_transform(const fhe_rns_ctx *r, int extended, bool inv, uint64_t *a, size_t n, size_t batch, fhe_mem mem, void *stream) {
    if (!r || (!a && n * batch)) return FHE_ERR_INVALID;
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    if (!is_pow2(n)) return FHE_ERR_INVALID;
    if (ilog2(n) > r->max_log_n) return FHE_ERR_NO_ROOT;
    if (n * batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t limbs = size_t(extended ? r->L + r->K : r->L);
    Mirror ma(a, n * batch * limbs, mem, true, st);
    if (ma.rc != FHE_OK) return ma.rc;
    int rc = FHE_OK;
    if (n > 1)
        rc = inv ? fhe::ntt_inv_multi(r->d_descs, (unsigned)limbs, ma.d, ilog2(n), batch * limbs, st, r->all_pm)
                 : fhe::ntt_fwd_multi(r->d_descs, (unsigned)limbs, ma.d, ilog2(n), batch * limbs, st, r->all_pm);
    return rc != FHE_OK ? rc : ma.sync_out(st);
}
}  // namespace

int fhe_rns_ntt_fwd(const fhe_rns_ctx *r, int extended, uint64_t *a, size_t n, size_t batch, fhe_mem mem, void *stream) {
    return rns_transform(r, extended, false, a, n, batch, mem, stream);
}
int fhe_rns_ntt_inv(const fhe_rns_ctx *r, int extended, uint64_t *a, size_t n, size_t batch, fhe_mem mem, void *stream) {
    return rns_transform(r, extended, true, a, n, batch, mem, stream);
}
int fhe_rns_pointwise_mul(const fhe_rns_ctx *r, int extended, uint64_t *a, const uint64_t *b, size_t n, size_t batch, fhe_mem mem,
                          void *stream) {
    if (!r || ((!a || !b) && n * batch)) return FHE_ERR_INVALID;
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    if (n * batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t limbs = size_t(extended ? r->L + r->K : r->L), count = n * batch * limbs;
    Mirror ma(a, count, mem, true, st), mb(b, count, mem, true, st);
    if (ma.rc | mb.rc) return FHE_ERR_HIP;
    if (n >> 31) return FHE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fhe::rns_pointwise_kernel, PointwiseGrid(n, batch * limbs).g, dim3(256), 0, st, ma.d, (const u64 *)mb.d, (unsigned)n,
                       (unsigned)limbs, batch * limbs, (const fhe::Barrett *)r->d_barrett);
    HIP_TRY(hipGetLastError());
    return ma.sync_out(st);
}

namespace {
int rns_addsub(const fhe_rns_ctx *r, int extended, uint64_t *a, const uint64_t *b, size_t n, size_t batch, fhe_mem mem, void *stream, int op) {
    if (!r || ((!a || (!b && op != 2)) && n * batch)) return FHE_ERR_INVALID;
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    if (n * batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t limbs = size_t(extended ? r->L + r->K : r->L), count = n * batch * limbs;
    Mirror ma(a, count, mem, true, st), mb(b, op != 2 ? count : 0, mem, true, st);
    if (ma.rc | mb.rc) return FHE_ERR_HIP;
    if (n >> 31) return FHE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fhe::rns_addsub_kernel, PointwiseGrid(n, batch * limbs).g, dim3(256), 0, st, ma.d, (const u64 *)mb.d, (unsigned)n, (unsigned)limbs,
                       batch * limbs, (const fhe::Barrett *)r->d_barrett, op);
    HIP_TRY(hipGetLastError());
    return ma.sync_out(st);
}
}  // namespace

// util/src/ring/rns.rs:254-270 `RnsRq += / -= RnsRq`, `-RnsRq` (either basis; CkksCiphertext + / -: ckks.rs add_sub): [batch][limbs][n]
int fhe_rns_add(const fhe_rns_ctx *r, int extended, uint64_t *a, const uint64_t *b, size_t n, size_t batch, fhe_mem mem, void *stream) {
    return rns_addsub(r, extended, a, b, n, batch, mem, stream, 0);
}
int fhe_rns_sub(const fhe_rns_ctx *r, int extended, uint64_t *a, const uint64_t *b, size_t n, size_t batch, fhe_mem mem, void *stream) {
    return rns_addsub(r, extended, a, b, n, batch, mem, stream, 1);
}
int fhe_rns_neg(const fhe_rns_ctx *r, int extended, uint64_t *a, size_t n, size_t batch, fhe_mem mem, void *stream) {
    return rns_addsub(r, extended, a, nullptr, n, batch, mem, stream, 2);
}

void fhe_ckks_key_destroy(fhe_ckks_key *k) {
    if (!k) return;
    if (k->rns && k->rns->device >= 0) {
        DeviceGuard guard(k->rns->device);
        if (k->d_kb) (void)hipFree(k->d_kb);
    }
    delete k;
}

int fhe_ckks_ksk_prepare(const fhe_rns_ctx *r, const uint64_t *ksk_b, const uint64_t *ksk_a, size_t n, fhe_mem mem, fhe_ckks_key **out) {
    if (!out) return FHE_ERR_INVALID;
    *out = nullptr;
    if (!r || !ksk_b || !ksk_a || !is_pow2(n)) return FHE_ERR_INVALID;
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    const int log_n = ilog2(n);
    for (const fhe_ctx *c : r->mods)
        if (log_n > c->s - 1) return FHE_ERR_NO_ROOT;
    if (log_n > r->max_log_n) return FHE_ERR_UNSUPPORTED;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t lk = size_t(r->L + r->K), words = lk * n;
    u64 *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 2 * words * sizeof(u64)));
    hipMemcpyKind kind = mem == FHE_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    hipError_t e = hipMemcpy(d, ksk_b, words * sizeof(u64), kind);
    if (e == hipSuccess) e = hipMemcpy(d + words, ksk_a, words * sizeof(u64), kind);
    int rc = e == hipSuccess ? FHE_OK : FHE_ERR_HIP;
    // both halves at once: polynomial p of the 2*lk uses descs[p % lk]
    if (rc == FHE_OK && n > 1) rc = fhe::ntt_fwd_multi(r->d_descs, (unsigned)lk, d, log_n, 2 * lk, nullptr, r->all_pm);
    if (hipDeviceSynchronize() != hipSuccess && rc == FHE_OK) rc = FHE_ERR_HIP;
    if (rc != FHE_OK) { (void)hipFree(d); return rc; }
    fhe_ckks_key *k = new (std::nothrow) fhe_ckks_key();
    if (!k) { (void)hipFree(d); return FHE_ERR_INVALID; }
    k->rns = r; k->log_n = log_n; k->d_kb = d; k->d_ka = d + words;
    *out = k;
    return FHE_OK;
}

namespace {
// scheme/ckks/src/ckks.rs:284-293 on device pointers: a_in, add_b (may be null = zero), add_a (may be null), out_b, out_a:
// [batch][L][n].  out_b = rescale_K(ksk.b * a~) + add_b, out_a = rescale_K(ksk.a * a~) + add_a.  Outputs may alias the addends
// and a_in (a_in is consumed before anything is written).
int key_switch_dev(const fhe_rns_ctx *r, const fhe_ckks_key *key, const u64 *a_in, const u64 *add_b, const u64 *add_a, u64 *out_b, u64 *out_a,
                   size_t batch, hipStream_t st) {
    const int log_n = key->log_n;
    const size_t n = size_t(1) << log_n, L = r->L, lk = size_t(r->L + r->K);
    const size_t blk = batch * lk * n;
    StreamWs wsp(3 * blk * sizeof(u64), st);  // ext | pb | pa, each [batch][lk][n]
    if (wsp.rc != FHE_OK) return wsp.rc;
    u64 *ws = wsp.as<u64>();
    u64 *ext = ws, *pb = ws + blk;
    int rc = FHE_OK;
    if (log_n == 15 && r->pm && r->L <= 8 && r->K <= 8 && edge_enabled()) {  // (wider bases: two limb vectors per thread would spill)
        // N = 2^15 (cfg4): layer 0 of the forward transforms runs inside the extend kernel, layer 0 of the inverse ones (and n^-1)
        // inside the rescales; the transform launches are 2^14 sub-transforms, two workgroups per CU (rns_kernels.hpp)
        launch_extend_edge(r, a_in, L * n, ext, lk * n, n, batch, st);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
        if (rc == FHE_OK) rc = fhe::ntt_fwd_inner15(r->d_descs, (unsigned)lk, ext, batch * lk, st, r->all_pm);
        if (rc == FHE_OK) {
            fhe::NttIo io;
            io.src = ext; io.src_mod = (unsigned)(batch * lk);
            io.mul = key->d_kb; io.mul_div = (unsigned)(batch * lk); io.mul_period = (unsigned)lk;
            rc = fhe::ntt_inv_inner15(r->d_descs, (unsigned)lk, pb, 2 * batch * lk, st, r->all_pm, io);
        }
        if (rc == FHE_OK) {
            launch_rescale_edge(r, rescale_in_block(pb, lk * n, r->L, r->K, n), out_b, L * n, add_b, L * n, n, batch, st);
            launch_rescale_edge(r, rescale_in_block(pb + blk, lk * n, r->L, r->K, n), out_a, L * n, add_a, L * n, n, batch, st);
            if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
        }
        return rc;
    }
    // ext[:, :L] = ct_a; ext[:, L:] = extend_bases(ct_a, ps): one kernel, the q-limbs written back out of the registers it read them into
    launch_extend(r, a_in, L * n, ext + L * n, lk * n, n, batch, st, ext, lk * n);
    if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    if (rc == FHE_OK && n > 1) rc = fhe::ntt_fwd_multi(r->d_descs, (unsigned)lk, ext, log_n, batch * lk, st, r->all_pm);
    // ksk.b * a~ and ksk.a * a~ (ring/rns.rs:148-158) ride on the load of ONE inverse launch over the 2 * batch * lk output
    // limbs: output limb s reads a~ limb s % (batch lk) and key limb (s / (batch lk)) lk + s % lk (d_kb and d_ka are adjacent)
    if (rc == FHE_OK && n > 1) {
        fhe::NttIo io;
        io.src = ext; io.src_mod = (unsigned)(batch * lk);
        io.mul = key->d_kb; io.mul_div = (unsigned)(batch * lk); io.mul_period = (unsigned)lk;
        rc = fhe::ntt_inv_multi(r->d_descs, (unsigned)lk, pb, log_n, 2 * batch * lk, st, r->all_pm, io);
    } else if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::rns_pointwise2_kernel, PointwiseGrid(n, batch * lk).g, dim3(256), 0, st, (const u64 *)ext, (const u64 *)key->d_kb,
                           (const u64 *)key->d_ka, pb, pb + blk, (unsigned)n, (unsigned)lk, batch * lk, (const fhe::Barrett *)r->d_barrett);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK) {
        launch_rescale(r, false, rescale_in_block(pb, lk * n, r->L, r->K, n), out_b, L * n, add_b, L * n, n, batch, st);
        launch_rescale(r, false, rescale_in_block(pb + blk, lk * n, r->L, r->K, n), out_a, L * n, add_a, L * n, n, batch, st);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    return rc;
}
}  // namespace

// scheme/ckks/src/ckks.rs:284-293 for `batch` ciphertexts sharing one key: ct_b, ct_a [batch][L][n], in place
int fhe_ckks_key_switch(const fhe_rns_ctx *r, const fhe_ckks_key *key, uint64_t *ct_b, uint64_t *ct_a, size_t batch, fhe_mem mem,
                        void *stream) {
    if (!r || !key || key->rns != r || ((!ct_b || !ct_a) && batch)) return FHE_ERR_INVALID;
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t n = size_t(1) << key->log_n, L = r->L;
    Mirror mb(ct_b, batch * L * n, mem, true, st), ma(ct_a, batch * L * n, mem, true, st);
    if (mb.rc | ma.rc) return FHE_ERR_HIP;
    int rc = key_switch_dev(r, key, ma.d, mb.d, nullptr, mb.d, ma.d, batch, st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    if (rc == FHE_OK) rc = ma.sync_out(st);
    return rc;
}

// ---- the key switch with its RNS limbs sharded over devices ------------------------------------
// `Ckks::key_switch` (scheme/ckks/src/ckks.rs:284-293) is limb-wise except for its two base conversions.  A device that owns the
// q-limbs [q_lo, q_hi) and the p-limbs [p_lo, p_hi):
//   stage 1  extends the replicated ct.a to ITS p-limbs (every source limb is needed for that, only the owned rows are computed),
//            transforms its nq + np limbs, multiplies by its key limbs on the inverse's load -- the same fused kernels as the
//            single-device key switch, over nl = nq + np limbs instead of L + K -- and leaves the q-limb products and the
//            p-limb products in two contiguous blocks (NttIo::dst2);
//   (the caller all-gathers the p-limb blocks: the ONE exchange of the path)
//   stage 2  rescales its q-limbs against all K gathered p-limbs (+ ct.b's owned limbs).
// All arithmetic is the single-device path's, so the outputs are the same bits (tests/test_rns_gpu.py).
void fhe_ckks_shard_destroy(fhe_ckks_shard *sh) {
    if (!sh) return;
    if (sh->rns && sh->rns->device >= 0) {
        DeviceGuard guard(sh->rns->device);
        if (sh->d_descs) (void)hipFree(sh->d_descs);
        if (sh->d_key) (void)hipFree(sh->d_key);
    }
    delete sh;
}

int fhe_ckks_shard_create(const fhe_rns_ctx *r, const fhe_ckks_key *key, int q_lo, int q_hi, int p_lo, int p_hi, fhe_ckks_shard **out) {
    if (!out) return FHE_ERR_INVALID;
    *out = nullptr;
    if (!r || !key || key->rns != r || q_lo < 0 || q_hi <= q_lo || q_hi > r->L || p_lo < 0 || p_hi <= p_lo || p_hi > r->K) return FHE_ERR_INVALID;
    if (r->K % (p_hi - p_lo)) return FHE_ERR_INVALID;  // the gathered p-limbs come as K / np equal contributions
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    if (!r->pm) return FHE_ERR_UNSUPPORTED;  // limb subsets exist on the pseudo-Mersenne route only (every CkksParam of the reference)
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    fhe_ckks_shard *sh = new (std::nothrow) fhe_ckks_shard();
    if (!sh) return FHE_ERR_INVALID;
    sh->rns = r; sh->log_n = key->log_n; sh->q_lo = q_lo; sh->q_hi = q_hi; sh->p_lo = p_lo; sh->p_hi = p_hi;
    sh->edge = key->log_n == 15 && r->L <= 8 && r->K <= 8;
    const int nq = q_hi - q_lo, np = p_hi - p_lo, nl = nq + np;
    const size_t n = size_t(1) << key->log_n;
    std::vector<fhe::ModDesc> descs(nl);
    for (int i = 0; i < nl; ++i) descs[i] = r->mods[i < nq ? q_lo + i : r->L + p_lo + (i - nq)]->h_desc;
    hipError_t e = hipMalloc((void **)&sh->d_descs, nl * sizeof(fhe::ModDesc));
    if (e == hipSuccess) e = hipMemcpy(sh->d_descs, descs.data(), nl * sizeof(fhe::ModDesc), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&sh->d_key, 2 * size_t(nl) * n * sizeof(u64));
    for (int part = 0; part < 2 && e == hipSuccess; ++part) {
        const u64 *full = part ? key->d_ka : key->d_kb;
        u64 *dst = sh->d_key + size_t(part) * nl * n;
        e = hipMemcpy(dst, full + size_t(q_lo) * n, size_t(nq) * n * sizeof(u64), hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(dst + size_t(nq) * n, full + size_t(r->L + p_lo) * n, size_t(np) * n * sizeof(u64), hipMemcpyDeviceToDevice);
    }
    if (e != hipSuccess) { g_last_hip = (int)e; fhe_ckks_shard_destroy(sh); return FHE_ERR_HIP; }
    *out = sh;
    return FHE_OK;
}

// stage 1: ct_a [batch][L][n] (ALL q-limbs: replicated) -> prod_q [2][batch][nq][n], prod_p [2][batch][np][n]: ksk.b * a~ (part 0) and
// ksk.a * a~ (part 1) on the owned limbs, coefficient domain (at N = 2^15 short of the inverse transforms' outermost layer and of
// n^-1, which stage 2 applies: an internal form, only fhe_ckks_shard_finish reads it)
int fhe_ckks_shard_products(const fhe_ckks_shard *sh, const uint64_t *ct_a, uint64_t *prod_q, uint64_t *prod_p, size_t batch, fhe_mem mem,
                            void *stream) {
    if (!sh || ((!ct_a || !prod_q || !prod_p) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    const fhe_rns_ctx *r = sh->rns;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const int nq = sh->q_hi - sh->q_lo, np = sh->p_hi - sh->p_lo, nl = nq + np, log_n = sh->log_n;
    const size_t n = size_t(1) << log_n, L = r->L;
    if ((2 * batch * nl) >> 30) return FHE_ERR_UNSUPPORTED;
    Mirror ma(ct_a, batch * L * n, mem, true, st), mq(prod_q, 2 * batch * nq * n, mem, false, st), mp(prod_p, 2 * batch * np * n, mem, false, st);
    if (ma.rc | mq.rc | mp.rc) return FHE_ERR_HIP;
    // the split store of the multiplying inverse exists in the wave-local kernels (2^12 .. 2^15); other ring sizes pack afterwards
    const bool split = log_n >= 14 && log_n <= 15 ? true : ((log_n == 12 || log_n == 13) && fhe::opt(fhe::OPT_NO_W12) == 0);
    StreamWs wsp((batch * nl * n + (split ? 0 : 2 * batch * nl * n)) * sizeof(u64), st);  // ext [batch][nl][n] (| prod [2][batch][nl][n])
    if (wsp.rc != FHE_OK) return wsp.rc;
    u64 *ext = wsp.as<u64>(), *tmp = ext + batch * nl * n;
    int rc = FHE_OK;
    if (sh->edge) launch_extend_edge(r, ma.d, L * n, ext, size_t(nl) * n, n, batch, st, sh->q_lo, sh->q_hi, sh->p_lo, sh->p_hi);
    else launch_extend(r, ma.d, L * n, ext + size_t(nq) * n, size_t(nl) * n, n, batch, st, ext, size_t(nl) * n, false, sh->q_lo, sh->q_hi, sh->p_lo, sh->p_hi);
    if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    if (rc == FHE_OK && n > 1)
        rc = sh->edge ? fhe::ntt_fwd_inner15(sh->d_descs, (unsigned)nl, ext, batch * nl, st, r->all_pm)
                      : fhe::ntt_fwd_multi(sh->d_descs, (unsigned)nl, ext, log_n, batch * nl, st, r->all_pm);
    if (rc == FHE_OK && n > 1) {
        fhe::NttIo io;
        io.src = ext; io.src_mod = (unsigned)(batch * nl);
        io.mul = sh->d_key; io.mul_div = (unsigned)(batch * nl); io.mul_period = (unsigned)nl;
        if (split) { io.dst2 = mp.d; io.dst_period = (unsigned)nl; io.dst_first2 = (unsigned)nq; }
        u64 *dst = split ? mq.d : tmp;
        rc = sh->edge ? fhe::ntt_inv_inner15(sh->d_descs, (unsigned)nl, dst, 2 * batch * nl, st, r->all_pm, io)
                      : fhe::ntt_inv_multi(sh->d_descs, (unsigned)nl, dst, log_n, 2 * batch * nl, st, r->all_pm, io);
    } else if (rc == FHE_OK) {  // n = 1: the ring product is the scalar product
        hipLaunchKernelGGL(fhe::rns_pointwise2_desc_kernel, PointwiseGrid(n, batch * nl).g, dim3(256), 0, st, (const u64 *)ext, (const u64 *)sh->d_key,
                           (const u64 *)(sh->d_key + size_t(nl) * n), tmp, tmp + batch * nl * n, (unsigned)n, (unsigned)nl, batch * nl, (const fhe::ModDesc *)sh->d_descs);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK && !split) {
        hipLaunchKernelGGL(fhe::rns_split_limbs_kernel, dim3(grid_for(2 * batch * nl * n)), dim3(256), 0, st, (const u64 *)tmp, mq.d, mp.d, n, 2 * batch, nl, nq);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK) rc = mq.sync_out(st);
    if (rc == FHE_OK) rc = mp.sync_out(st);
    return rc;
}

// stage 2: prod_q [2][batch][nq][n] (stage 1), gathered_p [K / np][2][batch][np][n] (every device's prod_p, in p-limb order), ct_b
// [batch][nq][n] (ct.b's owned limbs, or NULL) -> out_b, out_a [batch][nq][n]: the owned limbs of the switched ciphertext
int fhe_ckks_shard_finish(const fhe_ckks_shard *sh, const uint64_t *prod_q, const uint64_t *gathered_p, const uint64_t *ct_b, uint64_t *out_b,
                          uint64_t *out_a, size_t batch, fhe_mem mem, void *stream) {
    if (!sh || ((!prod_q || !gathered_p || !out_b || !out_a) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    const fhe_rns_ctx *r = sh->rns;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const int nq = sh->q_hi - sh->q_lo, np = sh->p_hi - sh->p_lo, K = r->K;
    const size_t n = size_t(1) << sh->log_n, qw = batch * nq * n, pw = batch * np * n;
    Mirror mq(prod_q, 2 * qw, mem, true, st), mg(gathered_p, size_t(K / np) * 2 * pw, mem, true, st), mb(ct_b, ct_b ? qw : 0, mem, true, st);
    Mirror mob(out_b, qw, mem, false, st), moa(out_a, qw, mem, false, st);
    if (mq.rc | mg.rc | mb.rc | mob.rc | moa.rc) return FHE_ERR_HIP;
    for (int part = 0; part < 2; ++part) {
        fhe::RescaleIn I{};
        I.in_q = mq.d + size_t(part) * qw; I.q_bs = size_t(nq) * n;
        I.in_p = mg.d + size_t(part) * pw; I.p_bs = size_t(np) * n;
        for (int j = 0; j < K; ++j) I.off[j] = size_t(j / np) * 2 * pw + size_t(j % np) * n;
        u64 *o = part ? moa.d : mob.d;
        const u64 *add = part ? nullptr : (ct_b ? mb.d : nullptr);
        if (sh->edge) launch_rescale_edge(r, I, o, size_t(nq) * n, add, size_t(nq) * n, n, batch, st, sh->q_lo, sh->q_hi);
        else launch_rescale(r, false, I, o, size_t(nq) * n, add, size_t(nq) * n, n, batch, st, sh->q_lo, sh->q_hi);
    }
    HIP_TRY(hipGetLastError());
    int rc = mob.sync_out(st);
    return rc != FHE_OK ? rc : moa.sync_out(st);
}

// util/src/ring/rns.rs:99-101 `RnsRq::rescale()` = rescale_k(1): in [batch][L][n] -> out [batch][L-1][n]
int fhe_rns_rescale(const fhe_rns_ctx *r, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem, void *stream) {
    if (!r || ((!in || !out) && n * batch)) return FHE_ERR_INVALID;
    if (r->L < 2) return FHE_ERR_INVALID;  // the reference would leave an RnsRq without limbs
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    if (n * batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t L = r->L;
    Mirror mi(in, n * batch * L, mem, true, st), mo(out, n * batch * (L - 1), mem, false, st);
    if (mi.rc | mo.rc) return FHE_ERR_HIP;
    launch_rescale(r, true, rescale_in_block(mi.d, L * n, r->L - 1, 1, n), mo.d, (L - 1) * n, nullptr, 0, n, batch, st);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

// scheme/ckks/src/ckks.rs:127-129 (util/src/avec.rs:34-50 on every limb): in, out [batch][L][n], t odd
int fhe_rns_automorphism(const fhe_rns_ctx *r, int64_t t, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem, void *stream) {
    if (!r || !is_pow2(n) || (n >> 30) || ((!in || !out) && batch) || in == out) return FHE_ERR_INVALID;
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    const int64_t two_n = 2 * (int64_t)n;
    const unsigned tt = (unsigned)(((t % two_n) + two_n) % two_n);  // t.rem_euclid(2n), avec.rs:38
    if (!(tt & 1) && n > 1) return FHE_ERR_UNSUPPORTED;  // CKKS only ever uses 5^j and -1; an even t is not a permutation
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t L = r->L;
    Mirror mi(in, n * batch * L, mem, true, st), mo(out, n * batch * L, mem, false, st);
    if (mi.rc | mo.rc) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::rns_automorphism_kernel, PointwiseGrid(n, batch * L).g, dim3(256), 0, st, (const u64 *)mi.d, mo.d, (unsigned)n, (unsigned)L,
                       batch * L, tt, (const fhe::Barrett *)r->d_barrett);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

// scheme/ckks/src/ckks.rs:274-282 `Ckks::rotate` (t = 5^j mod 2N) and `Ckks::conjugate` (t = -1): automorphism, then key switch
int fhe_ckks_rotate(const fhe_rns_ctx *r, const fhe_ckks_key *key, int64_t t, uint64_t *ct_b, uint64_t *ct_a, size_t batch, fhe_mem mem,
                    void *stream) {
    if (!r || !key || key->rns != r || ((!ct_b || !ct_a) && batch)) return FHE_ERR_INVALID;
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    const size_t n = size_t(1) << key->log_n, L = r->L;
    const int64_t two_n = 2 * (int64_t)n;
    const unsigned tt = (unsigned)(((t % two_n) + two_n) % two_n);
    if (!(tt & 1) && n > 1) return FHE_ERR_UNSUPPORTED;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    Mirror mb(ct_b, batch * L * n, mem, true, st), ma(ct_a, batch * L * n, mem, true, st);
    if (mb.rc | ma.rc) return FHE_ERR_HIP;
    const size_t words = batch * L * n;
    StreamWs rot(2 * words * sizeof(u64), st);
    if (rot.rc != FHE_OK) return rot.rc;
    u64 *rb = rot.as<u64>(), *ra = rb + words;
    hipLaunchKernelGGL(fhe::rns_automorphism_kernel, PointwiseGrid(n, batch * L).g, dim3(256), 0, st, (const u64 *)mb.d, rb, (unsigned)n, (unsigned)L,
                       batch * L, tt, (const fhe::Barrett *)r->d_barrett);
    hipLaunchKernelGGL(fhe::rns_automorphism_kernel, PointwiseGrid(n, batch * L).g, dim3(256), 0, st, (const u64 *)ma.d, ra, (unsigned)n, (unsigned)L,
                       batch * L, tt, (const fhe::Barrett *)r->d_barrett);
    HIP_TRY(hipGetLastError());
    int rc = key_switch_dev(r, key, ra, rb, nullptr, mb.d, ma.d, batch, st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    if (rc == FHE_OK) rc = ma.sync_out(st);
    return rc;
}

// scheme/ckks/src/ckks.rs:250-263 `Ckks::mul` with the relinearisation key `rlk` (ckks.rs:265-272) and the closing `rescale()`
// (ckks.rs:123-125): ct0, ct1 [batch][L][n] coefficient domain over qs -> out_b, out_a [batch][L-1][n].  The four input
// polynomials are transformed once each and stay in the evaluation domain through the tensor:
// 4 L forward + 3 L inverse transforms where the reference's four `Rq * Rq` take 12 L.
int fhe_ckks_mul(const fhe_rns_ctx *r, const fhe_ckks_key *rlk, const uint64_t *ct0_b, const uint64_t *ct0_a, const uint64_t *ct1_b,
                 const uint64_t *ct1_a, uint64_t *out_b, uint64_t *out_a, size_t batch, fhe_mem mem, void *stream) {
    if (!r || !rlk || rlk->rns != r || ((!ct0_b || !ct0_a || !ct1_b || !ct1_a || !out_b || !out_a) && batch)) return FHE_ERR_INVALID;
    if (r->L < 2) return FHE_ERR_INVALID;
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const int log_n = rlk->log_n;
    const size_t n = size_t(1) << log_n, L = r->L, words = batch * L * n;
    if (n >> 31) return FHE_ERR_UNSUPPORTED;
    const uint64_t *ins[4] = {ct0_b, ct0_a, ct1_b, ct1_a};
    Mirror m0(ins[0], words, mem, true, st), m1(ins[1], words, mem, true, st), m2(ins[2], words, mem, true, st), m3(ins[3], words, mem, true, st);
    Mirror mob(out_b, batch * (L - 1) * n, mem, false, st), moa(out_a, batch * (L - 1) * n, mem, false, st);
    if (m0.rc | m1.rc | m2.rc | m3.rc | mob.rc | moa.rc) return FHE_ERR_HIP;
    StreamWs wsp(7 * words * sizeof(u64), st);  // e [4][batch][L][n] | d [3][batch][L][n]
    if (wsp.rc != FHE_OK) return wsp.rc;
    u64 *e = wsp.as<u64>(), *d = e + 4 * words;
    const u64 *src[4] = {m0.d, m1.d, m2.d, m3.d};
    int rc = FHE_OK;
    if (n > 1 && log_n <= 15 && batch * L < (size_t(1) << 30)) {  // the four inputs in ONE launch (NttIo::src_group)
        fhe::NttIo io;
        io.src = src[0]; io.src2 = src[1]; io.src3 = src[2]; io.src4 = src[3]; io.src_group = (unsigned)(batch * L);
        rc = fhe::ntt_fwd_multi(r->d_descs, (unsigned)L, e, log_n, 4 * batch * L, st, r->all_pm, io);
    } else
    for (int i = 0; i < 4 && rc == FHE_OK; ++i) {
        if (n > 1) {
            fhe::NttIo io;
            io.src = src[i]; io.src_mod = (unsigned)(batch * L);
            rc = fhe::ntt_fwd_multi(r->d_descs, (unsigned)L, e + i * words, log_n, batch * L, st, r->all_pm, io);
        } else if (hipMemcpyAsync(e + i * words, src[i], words * sizeof(u64), hipMemcpyDeviceToDevice, st) != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::rns_tensor_kernel, PointwiseGrid(n, batch * L).g, dim3(256), 0, st, (const u64 *)e, d, (unsigned)n, (unsigned)L, batch * L,
                           (const fhe::Barrett *)r->d_barrett);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK && n > 1) rc = fhe::ntt_inv_multi(r->d_descs, (unsigned)L, d, log_n, 3 * batch * L, st, r->all_pm);
    // (d0, d1) + relinearize(d2) (ckks.rs:262, 265-272): the key switch adds d0 and d1 as it rescales; its outputs reuse e
    if (rc == FHE_OK) rc = key_switch_dev(r, rlk, d + 2 * words, d, d + words, e, e + words, batch, st);
    if (rc == FHE_OK) {
        launch_rescale(r, true, rescale_in_block(e, L * n, r->L - 1, 1, n), mob.d, (L - 1) * n, nullptr, 0, n, batch, st);
        launch_rescale(r, true, rescale_in_block(e + words, L * n, r->L - 1, 1, n), moa.d, (L - 1) * n, nullptr, 0, n, batch, st);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK) rc = mob.sync_out(st);
    if (rc == FHE_OK) rc = moa.sync_out(st);
    return rc;
}

// ---- CKKS key material on the device (scheme/ckks/src/ckks.rs:139-183, 215-225) ---------------------------------------------
}  // extern "C"
namespace {
// ckks.rs:215-225 on device buffers over the first `limbs` moduli of qs ++ ps: a uniform, e <- dg(3.2, 6), b = -(a s) + e + pt.
// sk [n] two's-complement i64; pt [pt_batch][limbs][n] or null; out_b, out_a [batch][limbs][n]
int ckks_sk_encrypt_dev(const fhe_rns_ctx *r, int limbs, const u64 *sk, const u64 *pt, size_t pt_batch, u64 *out_b, u64 *out_a, int log_n,
                        size_t batch, const fhe::ChaChaKey &K, unsigned long long *cursor, hipStream_t st) {
    const size_t n = size_t(1) << log_n;
    StreamWs ws((size_t(limbs) * n + batch * n) * sizeof(u64), st);
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *s_eval = ws.as<u64>(), *e = s_eval + size_t(limbs) * n;
    hipLaunchKernelGGL(fhe::rns_from_i64_kernel, dim3(grid_for(n * limbs)), dim3(256), 0, st, sk, s_eval, n, limbs, (const fhe::Barrett *)r->d_barrett,
                       (const u64 *)nullptr);
    int rc = hipGetLastError() == hipSuccess ? FHE_OK : FHE_ERR_HIP;
    if (rc == FHE_OK) rc = fhe::ntt_fwd_multi(r->d_descs, (unsigned)limbs, s_eval, log_n, limbs, st, r->all_pm);
    for (size_t c = 0; c < batch && rc == FHE_OK; ++c)
        for (int l = 0; l < limbs && rc == FHE_OK; ++l) {  // one modulus per launch: set-up code
            const uint64_t m = l < r->L ? r->qs[l] : r->ps[l - r->L];
            hipLaunchKernelGGL(fhe::sample_uniform_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, out_a + (c * limbs + l) * n, n, fhe::make_barrett(m), K,
                               *cursor);
            *cursor += (n + 3) / 4;
            if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
        }
    fhe::DgTable T;
    if (rc == FHE_OK && !fhe::make_dg_table(3.2, 6, &T)) rc = FHE_ERR_UNSUPPORTED;  // dg(3.2, 6): 39 entries
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::sample_dg_kernel, dim3(grid_for((batch * n + 7) / 8)), dim3(256), 0, st, e, batch * n, (u64)0, T, K, *cursor);
        *cursor += (batch * n + 7) / 8;
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK) {  // a s: forward of a out of place into b, inverse with the evaluation-domain key on its load (limb s % limbs)
        fhe::NttIo src;
        src.src = out_a; src.src_mod = (unsigned)(batch * limbs);
        rc = fhe::ntt_fwd_multi(r->d_descs, (unsigned)limbs, out_b, log_n, batch * limbs, st, r->all_pm, src);
    }
    if (rc == FHE_OK) {
        fhe::NttIo mul;
        mul.mul = s_eval; mul.mul_div = (unsigned)(batch * limbs); mul.mul_period = (unsigned)limbs;
        rc = fhe::ntt_inv_multi(r->d_descs, (unsigned)limbs, out_b, log_n, batch * limbs, st, r->all_pm, mul);
    }
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::ckks_finish_b_kernel, dim3(grid_for(batch * limbs * n)), dim3(256), 0, st, out_b, (const u64 *)e, pt, n, limbs, batch,
                           pt ? pt_batch : 1, (const fhe::Barrett *)r->d_barrett, 1);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    return rc;
}

// forward transforms of two caller buffers ([polys][n] each, polys a multiple of L) out of place into dst [2][polys][n]: one launch
// where the kernels take grouped sources (NttIo::src_group), two otherwise
int fwd_two_sources(const fhe_rns_ctx *r, const u64 *a, const u64 *b, u64 *dst, int log_n, size_t polys, hipStream_t st) {
    fhe::NttIo io;
    if (log_n >= 1 && log_n <= 15 && polys < (size_t(1) << 30)) {
        io.src = a; io.src2 = b; io.src_group = (unsigned)polys;
        return fhe::ntt_fwd_multi(r->d_descs, (unsigned)r->L, dst, log_n, 2 * polys, st, r->all_pm, io);
    }
    io.src = a; io.src_mod = (unsigned)polys;
    int rc = fhe::ntt_fwd_multi(r->d_descs, (unsigned)r->L, dst, log_n, polys, st, r->all_pm, io);
    io.src = b;
    return rc != FHE_OK ? rc : fhe::ntt_fwd_multi(r->d_descs, (unsigned)r->L, dst + (polys << log_n), log_n, polys, st, r->all_pm, io);
}

int ckks_ring_ok(const fhe_rns_ctx *r, size_t n) {
    if (!r || !is_pow2(n) || n < 2) return FHE_ERR_INVALID;
    if (r->device < 0) return FHE_ERR_NO_DEVICE;
    for (const fhe_ctx *c : r->mods)
        if (ilog2(n) > c->s - 1) return FHE_ERR_NO_ROOT;
    return ilog2(n) > r->max_log_n ? FHE_ERR_UNSUPPORTED : FHE_OK;
}
}  // namespace
extern "C" {

// util/src/misc/distribution.rs:10-21 `zo(rho)` as two's-complement i64 (ckks.rs:139-141 `Ckks::sk_gen`: rho = 0.5)
int fhe_sample_zo(double rho, const fhe_rng *rng, uint64_t stream_id, uint64_t *out, size_t count, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    if (!(rho >= 0 && rho <= 1.0) || (!out && count)) return FHE_ERR_INVALID;  // `assert!(rho <= 1.0)`
    if (count == 0) return FHE_OK;
    PtrDeviceGuard pguard(out, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    Mirror mo(out, count, mem, false, st);
    if (mo.rc != FHE_OK) return mo.rc;
    hipLaunchKernelGGL(fhe::sample_zo_kernel, dim3(grid_for((count + 7) / 8)), dim3(256), 0, st, mo.d, count, rho, fhe::call_key(rng, stream_id, fhe::RNG_SAMPLE_ZO), 0ull);
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}

// scheme/ckks/src/ckks.rs:215-225 `Ckks::sk_encrypt` for `batch` plaintexts over qs (extended = 0) or qs ++ ps: sk [n] i64;
// pt [batch][limbs][n] or NULL (zeros: ckks.rs:143-146 `pk_gen`); out_b, out_a [batch][limbs][n], coefficient domain
int fhe_ckks_sk_encrypt(const fhe_rns_ctx *r, int extended, const uint64_t *sk, const uint64_t *pt, size_t n, size_t batch, const fhe_rng *rng, uint64_t stream_id, uint64_t *out_b, uint64_t *out_a, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rc = ckks_ring_ok(r, n);
    if (rc != FHE_OK) return rc;
    if (!sk || ((!out_b || !out_a) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const int limbs = extended ? r->L + r->K : r->L;
    const size_t words = batch * limbs * n;
    Mirror msk(sk, n, mem, true, st), mpt(pt, pt ? words : 0, mem, true, st), mb(out_b, words, mem, false, st), ma(out_a, words, mem, false, st);
    if (msk.rc | mpt.rc | mb.rc | ma.rc) return FHE_ERR_HIP;
    unsigned long long cursor = 0;
    rc = ckks_sk_encrypt_dev(r, limbs, msk.d, pt ? mpt.d : nullptr, batch, mb.d, ma.d, ilog2(n), batch, fhe::call_key(rng, stream_id, fhe::RNG_CKKS_ENC), &cursor, st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    if (rc == FHE_OK) rc = ma.sync_out(st);
    return rc;
}

// scheme/ckks/src/ckks.rs:154-161 `Ckks::ksk_gen(param, sk, sk_prime)`: an encryption of sk' * P over qs ++ ps under sk.
// sk_prime NULL: sk' = sk^2 (ckks.rs:163-166 `rlk_gen`), the integer negacyclic square computed on the device.  For
// `cjk_gen` / `rtk_gen` (ckks.rs:168-183) pass sk(X^t), t = -1 / 5^j.  ksk_b, ksk_a [L+K][n], what fhe_ckks_ksk_prepare takes.
int fhe_ckks_ksk_gen(const fhe_rns_ctx *r, const uint64_t *sk, const uint64_t *sk_prime, size_t n, const fhe_rng *rng, uint64_t stream_id,
                     uint64_t *ksk_b, uint64_t *ksk_a, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rc = ckks_ring_ok(r, n);
    if (rc != FHE_OK) return rc;
    if (!sk || !ksk_b || !ksk_a) return FHE_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const int limbs = r->L + r->K, log_n = ilog2(n);
    const size_t words = size_t(limbs) * n;
    Mirror msk(sk, n, mem, true, st), msp(sk_prime, sk_prime ? n : 0, mem, true, st), mb(ksk_b, words, mem, false, st), ma(ksk_a, words, mem, false, st);
    if (msk.rc | msp.rc | mb.rc | ma.rc) return FHE_ERR_HIP;
    StreamWs ws((words + 2 * n + limbs + 1) * sizeof(u64), st);
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *pt = ws.as<u64>(), *sq = pt + words, *d_pm = sq + 2 * n, *d_max = d_pm + limbs;
    const u64 *spr = msp.d;
    unsigned long long sk_max = 0;
    if (!sk_prime) {
        // sk^2 over Z is computed mod q_0 and lifted back centred: exact only while n max|sk_i|^2 < q_0 / 2 (ternary keys from
        // fhe_sample_zo: |coefficient| <= n).  The bound is CHECKED on the device: any other key is an error, not a wrong key.
        if (hipMemsetAsync(d_max, 0, sizeof(u64), st) != hipSuccess) rc = FHE_ERR_HIP;
        if (rc == FHE_OK) {
            hipLaunchKernelGGL(fhe::max_abs_i64_kernel, dim3(grid_for(n)), dim3(256), 0, st, (const u64 *)msk.d, n, (unsigned long long *)d_max);
            if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
        }
        if (rc == FHE_OK && hipMemcpyAsync(&sk_max, d_max, sizeof(u64), hipMemcpyDeviceToHost, st) != hipSuccess) rc = FHE_ERR_HIP;
        hipLaunchKernelGGL(fhe::rns_from_i64_kernel, dim3(grid_for(n)), dim3(256), 0, st, (const u64 *)msk.d, sq, n, 1, (const fhe::Barrett *)r->d_barrett,
                           (const u64 *)nullptr);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
        if (rc == FHE_OK) rc = fhe::ntt_fwd_multi(r->d_descs, 1, sq, log_n, 1, st, r->mods[0]->pm_b);
        if (rc == FHE_OK) {
            fhe::NttIo io;
            io.mul = sq; io.mul_div = 1; io.mul_period = 1;
            if (hipMemcpyAsync(sq + n, sq, n * sizeof(u64), hipMemcpyDeviceToDevice, st) != hipSuccess) rc = FHE_ERR_HIP;
            if (rc == FHE_OK) rc = fhe::ntt_inv_multi(r->d_descs, 1, sq + n, log_n, 1, st, r->mods[0]->pm_b, io);
        }
        if (rc == FHE_OK) {
            hipLaunchKernelGGL(fhe::centre_to_i64_kernel, dim3(grid_for(n)), dim3(256), 0, st, (const u64 *)(sq + n), sq, n, (u64)r->qs[0]);
            if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
        }
        spr = sq;
    }
    // pt = RnsRq::from_i64(qps, sk') * P (ckks.rs:160): P mod p_j = 0, so the p-limbs of pt vanish
    std::vector<uint64_t> pm(limbs);
    for (int l = 0; l < limbs; ++l) pm[l] = prod_mod(r->ps, -1, l < r->L ? r->qs[l] : r->ps[l - r->L]);
    if (rc == FHE_OK && hipMemcpyAsync(d_pm, pm.data(), limbs * sizeof(u64), hipMemcpyHostToDevice, st) != hipSuccess) rc = FHE_ERR_HIP;
    if (rc == FHE_OK && hipStreamSynchronize(st) != hipSuccess) rc = FHE_ERR_HIP;  // pm is a stack-owned vector
    if (rc == FHE_OK && !sk_prime) {  // n max^2 < q_0 / 2, without overflow
        const fhe::u128 bound = (fhe::u128)n * sk_max * sk_max;
        if (sk_max >> 31 || bound >= (fhe::u128)(r->qs[0] / 2)) rc = FHE_ERR_INVALID;
    }
    if (rc == FHE_OK) {
        hipLaunchKernelGGL(fhe::rns_from_i64_kernel, dim3(grid_for(words)), dim3(256), 0, st, spr, pt, n, limbs, (const fhe::Barrett *)r->d_barrett,
                           (const u64 *)d_pm);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    unsigned long long cursor = 0;
    if (rc == FHE_OK) rc = ckks_sk_encrypt_dev(r, limbs, msk.d, pt, 1, mb.d, ma.d, log_n, 1, fhe::call_key(rng, stream_id, fhe::RNG_CKKS_KSK), &cursor, st);
    if (rc == FHE_OK) rc = mb.sync_out(st);
    if (rc == FHE_OK) rc = ma.sync_out(st);
    return rc;
}

// scheme/ckks/src/ckks.rs:240-248 `Ckks::decrypt`: pt = b + a * sk over qs.  sk [n] i64; ct_b, ct_a, pt [batch][L][n] (coefficient
// domain); pt may alias ct_b
int fhe_ckks_decrypt(const fhe_rns_ctx *r, const uint64_t *sk, const uint64_t *ct_b, const uint64_t *ct_a, size_t n, size_t batch, uint64_t *pt, fhe_mem mem,
                     void *stream) {
    int rc = ckks_ring_ok(r, n);
    if (rc != FHE_OK) return rc;
    if (!sk || ((!ct_b || !ct_a || !pt) && batch) || (batch && pt == ct_a)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const int L = r->L, log_n = ilog2(n);
    const size_t words = batch * L * n;
    if (batch * L >= (size_t(1) << 31)) return FHE_ERR_UNSUPPORTED;
    Mirror msk(sk, n, mem, true, st), mb(ct_b, words, mem, true, st), ma(ct_a, words, mem, true, st), mo(pt, words, mem, false, st);
    if (msk.rc | mb.rc | ma.rc | mo.rc) return FHE_ERR_HIP;
    StreamWs ws((size_t(L) * n + words) * sizeof(u64), st);  // the key's evaluations | a s
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *s_eval = ws.as<u64>(), *as = s_eval + size_t(L) * n;
    hipLaunchKernelGGL(fhe::rns_from_i64_kernel, dim3(grid_for(n * L)), dim3(256), 0, st, (const u64 *)msk.d, s_eval, n, L, (const fhe::Barrett *)r->d_barrett,
                       (const u64 *)nullptr, (size_t)1);
    HIP_TRY(hipGetLastError());
    rc = fhe::ntt_fwd_multi(r->d_descs, (unsigned)L, s_eval, log_n, L, st, r->all_pm);
    if (rc == FHE_OK) {
        fhe::NttIo src;
        src.src = ma.d; src.src_mod = (unsigned)(batch * L);
        rc = fhe::ntt_fwd_multi(r->d_descs, (unsigned)L, as, log_n, batch * L, st, r->all_pm, src);
    }
    if (rc == FHE_OK) {
        fhe::NttIo mul;
        mul.mul = s_eval; mul.mul_div = (unsigned)(batch * L); mul.mul_period = (unsigned)L;
        rc = fhe::ntt_inv_multi(r->d_descs, (unsigned)L, as, log_n, batch * L, st, r->all_pm, mul);
    }
    if (rc != FHE_OK) return rc;
    // pt = a s + b: formed in the scratch, then moved (pt may be ct_b itself)
    hipLaunchKernelGGL(fhe::ckks_finish_b_kernel, dim3(grid_for(words)), dim3(256), 0, st, as, (const u64 *)nullptr, (const u64 *)mb.d, n, L, batch, batch,
                       (const fhe::Barrett *)r->d_barrett, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(mo.d, as, words * sizeof(u64), hipMemcpyDeviceToDevice, st));
    return mo.sync_out(st);
}

// scheme/ckks/src/ckks.rs:227-238 `Ckks::pk_encrypt` for `batch` plaintexts over qs: u <- zo(0.5), e0, e1 <- dg(3.2, 6) per ciphertext;
// a = pk.a u + e0, b = pk.b u + e1 + pt.  pk_b, pk_a [L][n] (ckks.rs:143-146 `pk_gen` = fhe_ckks_sk_encrypt with pt NULL); pt
// [batch][L][n] or NULL; out_b, out_a [batch][L][n]
int fhe_ckks_pk_encrypt(const fhe_rns_ctx *r, const uint64_t *pk_b, const uint64_t *pk_a, const uint64_t *pt, size_t n, size_t batch, const fhe_rng *rng,
                        uint64_t stream_id, uint64_t *out_b, uint64_t *out_a, fhe_mem mem, void *stream) {
    if (!rng) return FHE_ERR_INVALID;
    int rc = ckks_ring_ok(r, n);
    if (rc != FHE_OK) return rc;
    if (!pk_b || !pk_a || ((!out_b || !out_a) && batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const int L = r->L, log_n = ilog2(n);
    const size_t words = batch * L * n, kw = size_t(L) * n;
    if (batch * L >= (size_t(1) << 31)) return FHE_ERR_UNSUPPORTED;
    Mirror mpb(pk_b, kw, mem, true, st), mpa(pk_a, kw, mem, true, st), mpt(pt, pt ? words : 0, mem, true, st), mb(out_b, words, mem, false, st),
        ma(out_a, words, mem, false, st);
    if (mpb.rc | mpa.rc | mpt.rc | mb.rc | ma.rc) return FHE_ERR_HIP;
    StreamWs ws((2 * kw + 3 * batch * n) * sizeof(u64), st);  // pk evaluations (b | a) | u | e0 | e1 as i64
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *pk_eval = ws.as<u64>(), *u = pk_eval + 2 * kw, *e0 = u + batch * n, *e1 = e0 + batch * n;
    const fhe::ChaChaKey K = fhe::call_key(rng, stream_id, fhe::RNG_CKKS_PK_ENC);
    unsigned long long cursor = 0;
    hipLaunchKernelGGL(fhe::sample_zo_kernel, dim3(grid_for((batch * n + 7) / 8)), dim3(256), 0, st, u, batch * n, 0.5, K, cursor);
    cursor += (batch * n + 7) / 8;
    HIP_TRY(hipGetLastError());
    fhe::DgTable T;
    if (!fhe::make_dg_table(3.2, 6, &T)) return FHE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fhe::sample_dg_kernel, dim3(grid_for((2 * batch * n + 7) / 8)), dim3(256), 0, st, e0, 2 * batch * n, (u64)0, T, K, cursor);  // e0 | e1
    HIP_TRY(hipGetLastError());
    rc = fwd_two_sources(r, mpb.d, mpa.d, pk_eval, log_n, L, st);  // pk.b, pk.a -> evaluation domain
    // u over every limb into both outputs' buffers, forward, then the inverse transforms multiply by pk.b / pk.a on their loads
    for (int half = 0; half < 2 && rc == FHE_OK; ++half) {
        u64 *o = half ? ma.d : mb.d;
        hipLaunchKernelGGL(fhe::rns_from_i64_kernel, dim3(grid_for(words)), dim3(256), 0, st, (const u64 *)u, o, n, L, (const fhe::Barrett *)r->d_barrett,
                           (const u64 *)nullptr, batch);
        if (hipGetLastError() != hipSuccess) { rc = FHE_ERR_HIP; break; }
        rc = fhe::ntt_fwd_multi(r->d_descs, (unsigned)L, o, log_n, batch * L, st, r->all_pm);
        if (rc != FHE_OK) break;
        fhe::NttIo mul;
        mul.mul = pk_eval + (half ? kw : 0); mul.mul_div = (unsigned)(batch * L); mul.mul_period = (unsigned)L;
        rc = fhe::ntt_inv_multi(r->d_descs, (unsigned)L, o, log_n, batch * L, st, r->all_pm, mul);
        if (rc != FHE_OK) break;
        hipLaunchKernelGGL(fhe::ckks_finish_b_kernel, dim3(grid_for(words)), dim3(256), 0, st, o, (const u64 *)(half ? e0 : e1),
                           (const u64 *)(half ? nullptr : (pt ? mpt.d : nullptr)), n, L, batch, batch, (const fhe::Barrett *)r->d_barrett, 0);
        if (hipGetLastError() != hipSuccess) rc = FHE_ERR_HIP;
    }
    if (rc == FHE_OK) rc = mb.sync_out(st);
    if (rc == FHE_OK) rc = ma.sync_out(st);
    return rc;
}

// scheme/ckks/src/ckks.rs:250-253 `Ckks::mul_constant` after its `encode`: (pt * b, pt * a).rescale().  pt [pt_batch][L][n] encoded
// plaintexts (pt_batch = 1: one constant for the whole batch, or pt_batch = batch); ct [batch][L][n] -> out [batch][L-1][n]
int fhe_ckks_mul_plain(const fhe_rns_ctx *r, const uint64_t *pt, size_t pt_batch, const uint64_t *ct_b, const uint64_t *ct_a, uint64_t *out_b, uint64_t *out_a,
                       size_t n, size_t batch, fhe_mem mem, void *stream) {
    int rc = ckks_ring_ok(r, n);
    if (rc != FHE_OK) return rc;
    if (r->L < 2 || ((!pt || !ct_b || !ct_a || !out_b || !out_a) && batch) || (batch && pt_batch != 1 && pt_batch != batch)) return FHE_ERR_INVALID;
    if (batch == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(r->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const int log_n = ilog2(n);
    const size_t L = r->L, words = batch * L * n, pw = pt_batch * L * n;
    if (2 * batch * L >= (size_t(1) << 31)) return FHE_ERR_UNSUPPORTED;
    Mirror mp(pt, pw, mem, true, st), mb(ct_b, words, mem, true, st), ma(ct_a, words, mem, true, st), mob(out_b, batch * (L - 1) * n, mem, false, st),
        moa(out_a, batch * (L - 1) * n, mem, false, st);
    if (mp.rc | mb.rc | ma.rc | mob.rc | moa.rc) return FHE_ERR_HIP;
    StreamWs ws((pw + 2 * words) * sizeof(u64), st);  // the plaintexts' evaluations | pt b | pt a
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *p_eval = ws.as<u64>(), *prod = p_eval + pw;
    {
        fhe::NttIo src;
        src.src = mp.d; src.src_mod = (unsigned)(pt_batch * L);
        rc = fhe::ntt_fwd_multi(r->d_descs, (unsigned)L, p_eval, log_n, pt_batch * L, st, r->all_pm, src);
    }
    if (rc == FHE_OK) rc = fwd_two_sources(r, mb.d, ma.d, prod, log_n, batch * L, st);
    if (rc == FHE_OK) {
        fhe::NttIo mul;
        mul.mul = p_eval; mul.mul_div = (unsigned)(2 * batch * L); mul.mul_period = (unsigned)(pt_batch * L);
        rc = fhe::ntt_inv_multi(r->d_descs, (unsigned)L, prod, log_n, 2 * batch * L, st, r->all_pm, mul);
    }
    if (rc != FHE_OK) return rc;
    launch_rescale(r, true, rescale_in_block(prod, L * n, r->L - 1, 1, n), mob.d, (L - 1) * n, nullptr, 0, n, batch, st);
    launch_rescale(r, true, rescale_in_block(prod + words, L * n, r->L - 1, 1, n), moa.d, (L - 1) * n, nullptr, 0, n, batch, st);
    HIP_TRY(hipGetLastError());
    rc = mob.sync_out(st);
    return rc != FHE_OK ? rc : moa.sync_out(st);
}

}  // extern "C"
