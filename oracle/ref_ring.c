/* CPU restatement (plain C, `unsigned __int128 % q`) of the han0110/learn-fhe ring hot path.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The product library (learn-fhe_amd/) never calls into this file.
 *
 * Same loop nests and the same arithmetic as the reference (three `u128 % u64` reductions per
 * butterfly, util/src/zq.rs:156-196 + util/src/ring/fft.rs:92-106), so it doubles as the
 * "restated CPU path" baseline (cpu_baseline.kind = "port").  Each function cites the reference
 * file:line it follows (paths relative to the reference tree).
 *
 * PARITY STATUS: the Rust reference cannot be built here and holds no fixed vectors.  Products,
 * external products, key switches and rescales are mathematically unique and pinned by the
 * reference's own test properties (schoolbook equality, round trip, CRT preservation).  The
 * forward-NTT output order/root choice and the digit values of `decompose` are PARITY UNPINNED
 * (restatement only; cross-checked against the independent Python restatement oracle/pyref.py).
 *
 * Build: oracle/Makefile -> oracle/_build/libref_ring.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef unsigned __int128 u128;

/* ---- Zq scalar: util/src/zq.rs:146-196 ------------------------------------------------ */
static inline u64 zq_add(u64 q, u64 a, u64 b) { return (u64)(((u128)a + (u128)b) % q); } /* 164-170 */
static inline u64 zq_neg(u64 q, u64 a) { return (q - a) % q; }                           /* 146-153 */
static inline u64 zq_sub(u64 q, u64 a, u64 b) { return zq_add(q, a, zq_neg(q, b)); }     /* 172-178 */
static inline u64 zq_mul(u64 q, u64 a, u64 b) { return (u64)(((u128)a * (u128)b) % q); } /* 180-186 */

static u64 zq_pow(u64 q, u64 v, u64 e) { /* zq.rs:111-117 (BigUint::modpow) */
    u64 r = 1 % q;
    v %= q;
    while (e) {
        if (e & 1) r = zq_mul(q, r, v);
        v = zq_mul(q, v, v);
        e >>= 1;
    }
    return r;
}

static u64 zq_inv(u64 q, u64 v) { /* zq.rs:123-126: extended gcd on i64, then rem_euclid */
    __int128 r0 = (__int128)q, r1 = (__int128)v, t0 = 0, t1 = 1;
    while (r1 != 0) {
        __int128 k = r0 / r1, tmp;
        tmp = r0 - k * r1; r0 = r1; r1 = tmp;
        tmp = t0 - k * t1; t0 = t1; t1 = tmp;
    }
    t0 %= (__int128)q;
    if (t0 < 0) t0 += q;
    return (u64)t0;
}

static inline u64 zq_to_center_u64(u64 q, u64 v) { /* zq.rs:83-89 */
    return v < (q >> 1) ? v : (~(q - v)) + 1;
}

int ref_is_prime(u64 n) { /* zq.rs:337-342 (probably_prime) == deterministic Miller-Rabin on u64 */
    static const u64 bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return 0;
    for (int i = 0; i < 12; i++)
        if (n % bases[i] == 0) return n == bases[i];
    u64 d = n - 1;
    int r = 0;
    while ((d & 1) == 0) { d >>= 1; r++; }
    for (int i = 0; i < 12; i++) {
        u64 x = zq_pow(n, bases[i], d);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int j = 1; j < r; j++) {
            x = zq_mul(n, x, x);
            if (x == n - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

u64 ref_generator(u64 q) { /* zq.rs:99-105 */
    u64 order = q - 1;
    for (u64 g = 1; g < order; g++)
        if (zq_pow(q, g, order >> 1) == order) return g;
    return 0;
}

/* zq.rs:325-329: write up to `count` primes k*2^log_n+1 scanning k downward from 2^(bits-log_n)-1 */
int ref_two_adic_primes(int bits, int log_n, int count, u64 *out) {
    u64 lo = 1ull << (bits - log_n - 1), hi = 1ull << (bits - log_n);
    int found = 0;
    for (u64 k = hi - 1; k >= lo && found < count; k--) {
        u64 c = (k << log_n) + 1;
        if (ref_is_prime(c)) out[found++] = c;
    }
    return found;
}

/* ---- bit reversal: util/src/misc.rs:29-42 --------------------------------------------- */
static void bit_reverse(u64 *v, size_t n) {
    if (n <= 2) return;
    int log_len = 0;
    while (((size_t)1 << log_len) < n) log_len++;
    for (size_t i = 0; i < n; i++) {
        size_t j = 0;
        for (int b = 0; b < log_len; b++) j |= ((i >> b) & 1) << (log_len - 1 - b);
        if (i < j) { u64 t = v[i]; v[i] = v[j]; v[j] = t; }
    }
}

/* ---- twiddles: util/src/ring/fft/zq.rs:58-67 ------------------------------------------ */
typedef struct {
    u64 q;
    int s;
    u64 g, omega;
    size_t len; /* 2^(s-1) */
    u64 *tw, *twi;
} ref_twiddle;

#define MAX_TW 4096
static ref_twiddle g_tw[MAX_TW];
static int g_ntw = 0;

static const ref_twiddle *twiddle_fill(u64 q);
static const ref_twiddle *twiddle(u64 q) { /* fft/zq.rs:49-56: per-modulus cache (the reference guards it with a Mutex) */
    int n;
#pragma omp atomic read
    n = g_ntw;
    for (int i = 0; i < n; i++)
        if (g_tw[i].q == q) return &g_tw[i];
    const ref_twiddle *t;
#pragma omp critical(ref_twiddle_cache)
    t = twiddle_fill(q);
    return t;
}
static const ref_twiddle *twiddle_fill(u64 q) { /* inside the critical section: look again, then build */
    for (int i = 0; i < g_ntw; i++)
        if (g_tw[i].q == q) return &g_tw[i];
    if (!ref_is_prime(q) || g_ntw == MAX_TW) return NULL;
    ref_twiddle *t = &g_tw[g_ntw];
    t->q = q;
    t->s = __builtin_ctzll(q - 1);
    t->g = ref_generator(q);
    t->omega = zq_pow(q, t->g, (q - 1) >> t->s); /* two_adic_generator, zq.rs:107-109 */
    t->len = (size_t)1 << (t->s - 1);
    t->tw = (u64 *)malloc(t->len * sizeof(u64));
    t->twi = (u64 *)malloc(t->len * sizeof(u64));
    u64 x = 1;
    for (size_t i = 0; i < t->len; i++) { t->tw[i] = x; x = zq_mul(q, x, t->omega); }
    for (size_t i = 0; i < t->len; i++) t->twi[i] = zq_inv(q, t->tw[i]);
    bit_reverse(t->tw, t->len);
    bit_reverse(t->twi, t->len);
#pragma omp atomic update
    g_ntw++; /* published last: readers outside the critical section only see complete entries */
    return t;
}

/* info for fixture checks: returns 0 on success */
int ref_twiddle_info(u64 q, int *s, u64 *g, u64 *omega, u64 *tw_prefix, u64 *twi_prefix, size_t count) {
    const ref_twiddle *t = twiddle(q);
    if (!t) return 1;
    *s = t->s; *g = t->g; *omega = t->omega;
    for (size_t i = 0; i < count && i < t->len; i++) { tw_prefix[i] = t->tw[i]; twi_prefix[i] = t->twi[i]; }
    return 0;
}

/* ---- NTT: util/src/ring/fft.rs:40-77, butterflies 92-106 ------------------------------ */
static void ntt_fwd_one(const ref_twiddle *t, u64 *a, size_t n) {
    u64 q = t->q;
    int log_n = __builtin_ctzll(n);
    for (int layer = 0; layer < log_n; layer++) {
        size_t m = (size_t)1 << layer, size = (size_t)1 << (log_n - layer - 1);
        for (size_t i = 0; i < m; i++) {
            u64 w = t->tw[m + i];
            u64 *u = a + 2 * size * i, *v = u + size;
            for (size_t k = 0; k < size; k++) { /* dit: tb = t*b; a' = a+tb; b' = a-tb */
                u64 tb = zq_mul(q, w, v[k]);
                u64 c = zq_add(q, u[k], tb), d = zq_sub(q, u[k], tb);
                u[k] = c; v[k] = d;
            }
        }
    }
}

static void ntt_inv_one(const ref_twiddle *t, u64 *a, size_t n) {
    u64 q = t->q;
    int log_n = __builtin_ctzll(n);
    u64 n_inv = zq_inv(q, (u64)n % q); /* fft/zq.rs:34 */
    for (int layer = log_n - 1; layer >= 0; layer--) {
        size_t m = (size_t)1 << layer, size = (size_t)1 << (log_n - layer - 1);
        for (size_t i = 0; i < m; i++) {
            u64 w = t->twi[m + i];
            u64 *u = a + 2 * size * i, *v = u + size;
            for (size_t k = 0; k < size; k++) { /* dif: a' = a+b; b' = (a-b)*t */
                u64 c = zq_add(q, u[k], v[k]), d = zq_mul(q, zq_sub(q, u[k], v[k]), w);
                u[k] = c; v[k] = d;
            }
        }
    }
    for (size_t i = 0; i < n; i++) a[i] = zq_mul(q, a[i], n_inv);
}

/* batch of contiguous polynomials [batch][n]; threads<=1 -> the reference's single-threaded model */
int ref_ntt_fwd(u64 q, u64 *a, size_t n, size_t batch, int threads) {
    const ref_twiddle *t = twiddle(q);
    if (!t || (n & (n - 1)) || n == 0 || (n > 1 && n > t->len)) return 1; /* n > 2^(s-1): slice OOB, fft.rs:45 */
    (void)threads;
#pragma omp parallel for num_threads(threads > 1 ? threads : 1) schedule(static)
    for (long b = 0; b < (long)batch; b++) ntt_fwd_one(t, a + (size_t)b * n, n);
    return 0;
}

int ref_ntt_inv(u64 q, u64 *a, size_t n, size_t batch, int threads) {
    const ref_twiddle *t = twiddle(q);
    if (!t || (n & (n - 1)) || n == 0 || (n > 1 && n > t->len)) return 1;
    (void)threads;
#pragma omp parallel for num_threads(threads > 1 ? threads : 1) schedule(static)
    for (long b = 0; b < (long)batch; b++) ntt_inv_one(t, a + (size_t)b * n, n);
    return 0;
}

/* fft/zq.rs:14-19: a <- a*b (negacyclic), coefficient domain */
static void ntt_mul_one(const ref_twiddle *t, u64 *a, const u64 *b, size_t n, u64 *scratch) {
    memcpy(scratch, b, n * sizeof(u64));
    ntt_fwd_one(t, a, n);
    ntt_fwd_one(t, scratch, n);
    for (size_t i = 0; i < n; i++) a[i] = zq_mul(t->q, a[i], scratch[i]);
    ntt_inv_one(t, a, n);
}

int ref_ntt_mul(u64 q, u64 *a, const u64 *b, size_t n, size_t batch) {
    const ref_twiddle *t = twiddle(q);
    if (!t || (n & (n - 1)) || n == 0 || (n > 1 && n > t->len)) return 1;
    u64 *scratch = (u64 *)malloc(n * sizeof(u64));
    for (size_t k = 0; k < batch; k++) ntt_mul_one(t, a + k * n, b + k * n, n, scratch);
    free(scratch);
    return 0;
}

void ref_pointwise_mul(u64 q, u64 *a, const u64 *b, size_t len) { /* ring.rs:266-270 */
    for (size_t i = 0; i < len; i++) a[i] = zq_mul(q, a[i], b[i]);
}

/* util/src/ring.rs:421-440: the reference tests' ground truth */
void ref_schoolbook_mul(u64 q, const u64 *a, const u64 *b, u64 *c, size_t n) {
    for (size_t i = 0; i < n; i++) c[i] = zq_mul(q, a[i], b[0]);
    for (size_t i = 0; i < n; i++)
        for (size_t j = 1; j < n; j++) {
            u64 p = zq_mul(q, a[i], b[j]);
            if (i + j < n) c[i + j] = zq_add(q, c[i + j], p);
            else c[i + j - n] = zq_sub(q, c[i + j - n], p);
        }
}

/* ---- automorphism / monomial: util/src/avec.rs:34-50, util/src/ring.rs:299-313 -------- */
void ref_automorphism(u64 q, int64_t t, const u64 *in, u64 *out, size_t n) {
    int64_t two_n = 2 * (int64_t)n;
    size_t tt = (size_t)(((t % two_n) + two_n) % two_n);
    memcpy(out, in, n * sizeof(u64));
    for (size_t i = 0; i < n; i++) {
        size_t it = (i * tt) % (2 * n);
        if (it < n) out[it] = in[i];
        else out[it - n] = zq_neg(q, in[i]);
    }
}

void ref_monomial_mul(u64 q, int64_t k, const u64 *in, u64 *out, size_t n) {
    int64_t two_n = 2 * (int64_t)n;
    size_t i = (size_t)(((k % two_n) + two_n) % two_n), r = i % n;
    for (size_t j = 0; j < n; j++) out[(j + r) % n] = in[j]; /* rotate_right(r) */
    if (i < n) { for (size_t j = 0; j < i; j++) out[j] = zq_neg(q, out[j]); }
    else { for (size_t j = i - n; j < n; j++) out[j] = zq_neg(q, out[j]); }
}

/* ---- gadget decomposition: util/src/misc/decompose.rs:49-64, 91-112, 137-155 ---------- */
static int dec_log_q(u64 q) { /* q.next_power_of_two().ilog2() */
    return q <= 1 ? 0 : 64 - __builtin_clzll(q - 1);
}

int ref_rounding_bits(u64 q, int log_b, int d) {
    int lq = dec_log_q(q), rb = lq - log_b * d;
    return rb > 0 ? rb : 0;
}

/* out is digit-major: out[j*n + i] = digit j of in[i] */
void ref_decompose(u64 q, int log_b, int d, const u64 *in, size_t n, u64 *out) {
    int bits = ref_rounding_bits(q, log_b, d);
    u64 b_by_2 = 1ull << (log_b - 1), mask = (1ull << log_b) - 1, neg_b = q - (1ull << log_b);
    for (size_t i = 0; i < n; i++) {
        u64 rounded = zq_add(q, in[i], ((1ull << bits) >> 1) % q); /* rounding_shr, 92-95 */
        u64 v = (rounded >> bits) % q;
        u64 c = zq_to_center_u64(q, v);
        for (int j = 0; j < d; j++) {
            u64 limb = c & mask;
            u64 carry = (limb + (c & 1)) > b_by_2;
            c >>= log_b;
            c += carry;
            out[(size_t)j * n + i] = (limb + carry * neg_b) % q;
        }
    }
}

/* ---- RLWE key switch / RGSW external product: rlwe.rs:177-191, rgsw.rs:116-128 -------- */
/* keys: rows x n, coefficient domain.  acc += sum_j keys[j]*limbs[j] done as in misc.rs:50-62 */
static void dot_rows(const ref_twiddle *t, const u64 *keys, const u64 *limbs, int rows, size_t n, u64 *out,
                     u64 *tmp, u64 *scratch) {
    for (int j = 0; j < rows; j++) {
        memcpy(tmp, keys + (size_t)j * n, n * sizeof(u64));
        ntt_mul_one(t, tmp, limbs + (size_t)j * n, n, scratch);
        if (j == 0) memcpy(out, tmp, n * sizeof(u64));
        else for (size_t i = 0; i < n; i++) out[i] = zq_add(t->q, out[i], tmp[i]);
    }
}

/* ksk_a, ksk_b: [d][n]; ct in/out: a[n], b[n] */
int ref_rlwe_key_switch(u64 q, int log_b, int d, const u64 *ksk_a, const u64 *ksk_b, u64 *ct_a, u64 *ct_b,
                        size_t n) {
    const ref_twiddle *t = twiddle(q);
    if (!t) return 1;
    u64 *limbs = (u64 *)malloc((size_t)d * n * sizeof(u64));
    u64 *oa = (u64 *)malloc(4 * n * sizeof(u64)), *ob = oa + n, *tmp = ob + n, *scr = tmp + n;
    ref_decompose(q, log_b, d, ct_a, n, limbs);
    dot_rows(t, ksk_a, limbs, d, n, oa, tmp, scr);
    dot_rows(t, ksk_b, limbs, d, n, ob, tmp, scr);
    for (size_t i = 0; i < n; i++) { ct_b[i] = zq_add(q, ob[i], ct_b[i]); ct_a[i] = oa[i]; }
    free(limbs); free(oa);
    return 0;
}

int ref_rlwe_automorphism(u64 q, int log_b, int d, int64_t tt, const u64 *ak_a, const u64 *ak_b, u64 *ct_a,
                          u64 *ct_b, size_t n) {
    u64 *ta = (u64 *)malloc(2 * n * sizeof(u64)), *tb = ta + n;
    ref_automorphism(q, tt, ct_a, ta, n);
    ref_automorphism(q, tt, ct_b, tb, n);
    memcpy(ct_a, ta, n * sizeof(u64)); memcpy(ct_b, tb, n * sizeof(u64));
    free(ta);
    return ref_rlwe_key_switch(q, log_b, d, ak_a, ak_b, ct_a, ct_b, n);
}

/* rgsw_a, rgsw_b: [2d][n] */
int ref_external_product(u64 q, int log_b, int d, const u64 *rgsw_a, const u64 *rgsw_b, u64 *ct_a, u64 *ct_b,
                         size_t n) {
    const ref_twiddle *t = twiddle(q);
    if (!t) return 1;
    u64 *limbs = (u64 *)malloc((size_t)2 * d * n * sizeof(u64));
    u64 *oa = (u64 *)malloc(4 * n * sizeof(u64)), *ob = oa + n, *tmp = ob + n, *scr = tmp + n;
    ref_decompose(q, log_b, d, ct_a, n, limbs);
    ref_decompose(q, log_b, d, ct_b, n, limbs + (size_t)d * n);
    dot_rows(t, rgsw_a, limbs, 2 * d, n, oa, tmp, scr);
    dot_rows(t, rgsw_b, limbs, 2 * d, n, ob, tmp, scr);
    memcpy(ct_a, oa, n * sizeof(u64)); memcpy(ct_b, ob, n * sizeof(u64));
    free(limbs); free(oa);
    return 0;
}

/* ---- LMKCDEY blind rotation: scheme/fhew/src/bootstrapping.rs:158-231 ----------------- */
/* schedule: ops[k] = (kind<<32)|idx, kind 0 = external product with brk[idx], 1 = automorphism ak[idx].
 * a: n_lwe values mod 2N.  returns number of ops (<= n_lwe + N + 2) */
size_t ref_blind_rotate_schedule(size_t n, int w, const u64 *a, size_t n_lwe, u64 *ops) {
    size_t q2 = 2 * n, half = n / 2, nops = 0;
    /* log_g_map (228-231): index -> (l, sign) */
    int32_t *lm = (int32_t *)malloc(q2 * sizeof(int32_t)), *lp = (int32_t *)malloc(q2 * sizeof(int32_t));
    for (size_t i = 0; i < q2; i++) lm[i] = lp[i] = -1;
    u64 x = 1;
    for (size_t l = 0; l < half; l++) {
        lp[x % q2] = (int32_t)l;
        lm[(q2 - x) % q2] = (int32_t)l;
        x = x * 5 % q2;
    }
    /* buckets i_minus/i_plus (212-226), preserving index order inside a bucket */
    size_t *cnt_m = (size_t *)calloc(2 * half + 2, sizeof(size_t)), *cnt_p = cnt_m + half + 1;
    for (size_t i = 0; i < n_lwe; i++) {
        u64 ai = a[i];
        if (lm[ai] >= 0 && lp[ai] < 0) cnt_m[lm[ai]]++;
        else if (lp[ai] >= 0 && lm[ai] < 0) cnt_p[lp[ai]]++;
    }
#define EMIT_BUCKET(map, l)                                                   \
    for (size_t i = 0; i < n_lwe; i++)                                        \
        if (map[a[i]] == (int32_t)(l) && (map == lm ? lp[a[i]] < 0 : lm[a[i]] < 0)) ops[nops++] = (u64)i;
    int v = 0;
    for (size_t l = half - 1; l >= 1; l--) { /* 181-190 */
        EMIT_BUCKET(lm, l)
        v++;
        if (cnt_m[l - 1] != 0 || v == w || l == 1) { ops[nops++] = (1ull << 32) | (u64)v; v = 0; }
    }
    EMIT_BUCKET(lm, 0)                       /* 191-193 */
    ops[nops++] = (1ull << 32) | 0;          /* 194 */
    for (size_t l = half - 1; l >= 1; l--) { /* 195-204 */
        EMIT_BUCKET(lp, l)
        v++;
        if (cnt_p[l - 1] != 0 || v == w || l == 1) { ops[nops++] = (1ull << 32) | (u64)v; v = 0; }
    }
    EMIT_BUCKET(lp, 0)                       /* 205-207 */
#undef EMIT_BUCKET
    free(lm); free(lp); free(cnt_m);
    return nops;
}

/* brk: [n_lwe][2][2d][n] (a rows then b rows); ak: [w+1][2][d_ks][n]; ak_t: [w+1] exponents.
 * f: LUT polynomial; lwe_a: n_lwe values mod 2N, lwe_b mod 2N.  out: acc a[n], b[n]. */
int ref_blind_rotate(u64 q, size_t n, int w, int log_b, int d, int ks_log_b, int ks_d, const u64 *brk,
                     const u64 *ak, const int64_t *ak_t, const u64 *f, const u64 *lwe_a, u64 lwe_b,
                     size_t n_lwe, u64 *out_a, u64 *out_b) {
    size_t q2 = 2 * n;
    u64 *tmp = (u64 *)malloc(n * sizeof(u64));
    ref_automorphism(q, -5, f, tmp, n); /* 165: f.automorphism(-g) */
    u64 bg = (lwe_b % q2) * (5 % q2) % q2;
    int64_t k = bg < (q2 >> 1) ? (int64_t)bg : (int64_t)bg - (int64_t)q2; /* X ^ Zq -> to_i64 */
    ref_monomial_mul(q, k, tmp, out_b, n);
    memset(out_a, 0, n * sizeof(u64));
    free(tmp);
    u64 *ops = (u64 *)malloc((n_lwe + n + 4) * sizeof(u64));
    size_t nops = ref_blind_rotate_schedule(n, w, lwe_a, n_lwe, ops);
    int rc = 0;
    for (size_t o = 0; o < nops && !rc; o++) {
        u64 idx = ops[o] & 0xffffffffull;
        if ((ops[o] >> 32) == 0) {
            const u64 *key = brk + idx * (size_t)4 * d * n;
            rc = ref_external_product(q, log_b, d, key, key + (size_t)2 * d * n, out_a, out_b, n);
        } else {
            const u64 *key = ak + idx * (size_t)2 * ks_d * n;
            rc = ref_rlwe_automorphism(q, ks_log_b, ks_d, ak_t[idx], key, key + (size_t)ks_d * n, out_a, out_b, n);
        }
    }
    free(ops);
    return rc;
}

/* rlwe.rs:193-202 */
void ref_sample_extract(u64 q, const u64 *ct_a, const u64 *ct_b, size_t n, size_t i, u64 *lwe_a, u64 *lwe_b) {
    size_t k = 0;
    for (size_t j = i + 1; j-- > 0;) lwe_a[k++] = ct_a[j];
    for (size_t j = n; j-- > i + 1;) lwe_a[k++] = zq_neg(q, ct_a[j]);
    *lwe_b = ct_b[i];
}

/* ---- f64 modulus switching: util/src/zq.rs:59-61, 128-140 ----------------------------- */
static u64 zq_from_f64(u64 q, double x) {
    double r = round(x); /* C round() == Rust f64::round: half away from zero */
    int64_t v = r >= 9.223372036854775807e18 ? INT64_MAX : (r <= -9.223372036854775808e18 ? INT64_MIN : (int64_t)r);
    int64_t m = v % (int64_t)q;
    if (m < 0) m += (int64_t)q;
    return (u64)m;
}

u64 ref_mod_switch(u64 q, u64 v, u64 q_prime) { return zq_from_f64(q_prime, ((double)v * (double)q_prime) / (double)q); }

u64 ref_mod_switch_odd(u64 q, u64 v, u64 q_prime) {
    double x = ((double)v * (double)q_prime) / (double)q, u = floor(x);
    if (u == 0.0) return ((u64)round(x)) % q_prime;
    return (((u64)u) | 1) % q_prime;
}

/* lwe.rs:151-160.  ksk_a: [d*n_in][n_out], ksk_b: [d*n_in]; limbs digit-major */
void ref_lwe_key_switch(u64 q, int log_b, int d, const u64 *ksk_a, const u64 *ksk_b, const u64 *ct_a, u64 ct_b,
                        size_t n_in, size_t n_out, u64 *out_a, u64 *out_b) {
    u64 *limbs = (u64 *)malloc((size_t)d * n_in * sizeof(u64));
    ref_decompose(q, log_b, d, ct_a, n_in, limbs);
    memset(out_a, 0, n_out * sizeof(u64));
    u64 b = 0;
    for (size_t r = 0; r < (size_t)d * n_in; r++) {
        for (size_t k = 0; k < n_out; k++) out_a[k] = zq_add(q, out_a[k], zq_mul(q, ksk_a[r * n_out + k], limbs[r]));
        b = zq_add(q, b, zq_mul(q, ksk_b[r], limbs[r]));
    }
    *out_b = zq_add(q, b, ct_b);
    free(limbs);
}

/* ---- RNS: util/src/ring/rns.rs:83-132, 278-346 ---------------------------------------- */
static u64 prod_mod(const u64 *xs, int cnt, int skip, u64 m) {
    u64 r = 1 % m;
    for (int i = 0; i < cnt; i++)
        if (i != skip) r = zq_mul(m, r, xs[i] % m);
    return r;
}

/* limbs_in: [L][n] over qs; limbs_out: [K][n] over ps */
int ref_rns_extend_bases(const u64 *qs, int L, const u64 *ps, int K, const u64 *in, u64 *out, size_t n) {
    if (L > 64 || K > 64) return 1;
    u64 q_hat_inv[64], q_hats_ps[64][64], uq_ps[64][65];
    double q_fracs[64];
    for (int i = 0; i < L; i++) {
        q_hat_inv[i] = zq_inv(qs[i], prod_mod(qs, L, i, qs[i])); /* 290-293 */
        q_fracs[i] = 1.0 / (double)qs[i];                        /* 294 */
    }
    for (int j = 0; j < K; j++) {
        for (int i = 0; i < L; i++) q_hats_ps[j][i] = prod_mod(qs, L, i, ps[j]); /* 305-313 */
        u64 qmod = prod_mod(qs, L, -1, ps[j]);
        for (int u = 0; u <= L; u++) uq_ps[j][u] = zq_mul(ps[j], qmod, (u64)u % ps[j]); /* 315-320 */
    }
    for (size_t c = 0; c < n; c++) { /* 331-345 */
        u64 vs[64];
        double acc = 0.0;
        for (int i = 0; i < L; i++) {
            vs[i] = zq_mul(qs[i], in[(size_t)i * n + c], q_hat_inv[i]);
            acc = acc + q_fracs[i] * (double)vs[i];
        }
        size_t u = (size_t)round(acc);
        for (int j = 0; j < K; j++) {
            u64 dot = 0;
            for (int i = 0; i < L; i++) {
                u64 term = zq_mul(ps[j], q_hats_ps[j][i], vs[i] % ps[j]);
                dot = i == 0 ? term : zq_add(ps[j], dot, term);
            }
            out[(size_t)j * n + c] = zq_sub(ps[j], dot, uq_ps[j][u]);
        }
    }
    return 0;
}

/* limbs: [L+K][n] over qps (in place on the first L limbs) */
int ref_rns_rescale_k(const u64 *qps, int LK, int K, u64 *limbs, size_t n) {
    int L = LK - K;
    const u64 *qs = qps, *ps = qps + L;
    if (K <= 0 || L <= 0) return 1;
    for (int i = 0; i < LK; i++) { /* round(), 120-125: += floor(P/2) mod q_i.  P is odd, so
                                      floor(P/2) = (P-1)/2 = (P mod m - 1) * 2^{-1} (mod m), m odd */
        u64 m = qps[i];
        u64 p_half = zq_mul(m, zq_sub(m, prod_mod(ps, K, -1, m), 1 % m), (m + 1) / 2 % m);
        for (size_t c = 0; c < n; c++) limbs[(size_t)i * n + c] = zq_add(m, limbs[(size_t)i * n + c], p_half);
    }
    if (K == 1) { /* 108-111 */
        const u64 *rp = limbs + (size_t)L * n;
        for (int i = 0; i < L; i++)
            for (size_t c = 0; c < n; c++)
                limbs[(size_t)i * n + c] = zq_sub(qs[i], limbs[(size_t)i * n + c], rp[c] % qs[i]);
    } else { /* 112-115 */
        u64 *sw = (u64 *)malloc((size_t)L * n * sizeof(u64));
        if (ref_rns_extend_bases(ps, K, qs, L, limbs + (size_t)L * n, sw, n)) { free(sw); return 1; }
        for (int i = 0; i < L; i++)
            for (size_t c = 0; c < n; c++)
                limbs[(size_t)i * n + c] = zq_sub(qs[i], limbs[(size_t)i * n + c], sw[(size_t)i * n + c]);
        free(sw);
    }
    for (int i = 0; i < L; i++) { /* div(), 127-132 */
        u64 p_inv = zq_inv(qs[i], prod_mod(ps, K, -1, qs[i]));
        for (size_t c = 0; c < n; c++) limbs[(size_t)i * n + c] = zq_mul(qs[i], limbs[(size_t)i * n + c], p_inv);
    }
    return 0;
}

/* scheme/ckks/src/ckks.rs:284-293.  ksk_b, ksk_a: [L+K][n]; ct_b, ct_a: [L][n] (in/out) */
int ref_ckks_key_switch(const u64 *qs, int L, const u64 *ps, int K, const u64 *ksk_b, const u64 *ksk_a, u64 *ct_b,
                        u64 *ct_a, size_t n) {
    int LK = L + K;
    u64 qps[128];
    if (LK > 128) return 1;
    memcpy(qps, qs, L * sizeof(u64)); memcpy(qps + L, ps, K * sizeof(u64));
    u64 *ext = (u64 *)malloc((size_t)3 * LK * n * sizeof(u64)), *pb = ext + (size_t)LK * n, *pa = pb + (size_t)LK * n;
    memcpy(ext, ct_a, (size_t)L * n * sizeof(u64));
    int rc = ref_rns_extend_bases(qs, L, ps, K, ct_a, ext + (size_t)L * n, n);
    memcpy(pb, ksk_b, (size_t)LK * n * sizeof(u64));
    memcpy(pa, ksk_a, (size_t)LK * n * sizeof(u64));
    for (int i = 0; i < LK && !rc; i++) {
        rc |= ref_ntt_mul(qps[i], pb + (size_t)i * n, ext + (size_t)i * n, n, 1);
        rc |= ref_ntt_mul(qps[i], pa + (size_t)i * n, ext + (size_t)i * n, n, 1);
    }
    if (!rc) rc = ref_rns_rescale_k(qps, LK, K, pb, n) | ref_rns_rescale_k(qps, LK, K, pa, n);
    if (!rc)
        for (int i = 0; i < L; i++)
            for (size_t c = 0; c < n; c++) {
                ct_b[(size_t)i * n + c] = zq_add(qs[i], pb[(size_t)i * n + c], ct_b[(size_t)i * n + c]);
                ct_a[(size_t)i * n + c] = pa[(size_t)i * n + c];
            }
    free(ext);
    return rc;
}

/* ==== TFHE torus path, k = 1 ======================================
 * Two products over Z_{2^64}[X]/(X^N+1):
 *   exact  -- wrapping u64 schoolbook == the signed-integer negacyclic product mod 2^64 (two's complement): the value
 *             row T's acceptance rule (ii) measures distances to; what the GPU path must equal bit for bit;
 *   fft64  -- the reference's own algorithm restated (util/src/ring/fft/c64.rs:11-108, util/src/ring/fft.rs:9-35):
 *             twist by cis(pi i / N), cyclic f64 FFT of size N/2, pointwise, inverse, untwist, f64_mod_u64.  Its low
 *             bits carry rounding noise (libm cis), so it is NOT a bit-exact checker: it is the CPU baseline
 *             (cpu_baseline.kind = "port") and the subject of the tolerance test against `exact`. */
void ref_torus_mul_exact(const u64 *a, const u64 *b, u64 *c, size_t n) { /* ring.rs:421-440 on T64 (torus.rs:48-100) */
    for (size_t k = 0; k < n; k++) c[k] = 0;
    for (size_t i = 0; i < n; i++) {
        const u64 x = a[i];
        if (!x) continue;
        for (size_t j = 0; j < n - i; j++) c[i + j] += x * b[j];
        for (size_t j = n - i; j < n; j++) c[i + j - n] -= x * b[j];
    }
}

void ref_torus_monomial_mul(int64_t k, const u64 *in, u64 *out, size_t n) { /* ring.rs:299-313, 380-406 on Rt */
    const int64_t two_n = 2 * (int64_t)n;
    const size_t i = (size_t)(((k % two_n) + two_n) % two_n), r = i % n;
    for (size_t j = 0; j < n; j++) out[(j + r) % n] = in[j]; /* rotate_right(r) */
    if (i < n) for (size_t j = 0; j < i; j++) out[j] = 0 - out[j];
    else for (size_t j = i - n; j < n; j++) out[j] = 0 - out[j];
}

/* decompose.rs:66-81 (new), 114-135: out digit-major [d][n] */
void ref_torus_decompose(int log_b, int d, const u64 *in, size_t n, u64 *out) {
    const int rb = 64 - log_b * d > 0 ? 64 - log_b * d : 0;
    const u64 mask = (log_b >= 64) ? ~(u64)0 : (((u64)1 << log_b) - 1);
    for (size_t i = 0; i < n; i++) {
        u64 v = in[i];
        if (rb) v = (v + (((u64)1 << rb) >> 1)) >> rb; /* rounding_shr, 115-118 (wrapping add) */
        for (int j = 0; j < d; j++) {
            const u64 limb = v & mask;
            v = log_b >= 64 ? 0 : v >> log_b;
            const u64 carry = (((limb - 1) | v) & limb) >> (log_b - 1);
            v += carry;
            out[(size_t)j * n + i] = limb - (carry << log_b);
        }
    }
}

/* c64.rs:69-85 */
static inline u64 f64_mod_u64(double v) {
    u64 bits;
    memcpy(&bits, &v, 8);
    const u64 sign = bits >> 63, exponent = (bits >> 52) & 0x7ff, mantissa = (bits << 11) | 0x8000000000000000ull;
    const int64_t shift = 1086 - (int64_t)exponent;
    u64 value;
    if (shift >= -63 && shift <= 0) value = mantissa << (-shift);
    else if (shift >= 1 && shift <= 64) value = ((shift - 1 >= 64 ? 0 : (mantissa >> (shift - 1))) + 1) >> 1;
    else value = 0;
    return sign ? 0 - value : value;
}

typedef struct { double re, im; } c64;
static inline c64 c_mul(c64 a, c64 b) { c64 r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
static inline c64 c_add(c64 a, c64 b) { c64 r = {a.re + b.re, a.im + b.im}; return r; }
static inline c64 c_sub(c64 a, c64 b) { c64 r = {a.re - b.re, a.im - b.im}; return r; }

typedef struct { size_t n; c64 *tw, *twi, *tw_bo, *twi_bo; } ref_tw64; /* c64.rs:88-108: [normal, inverse, both bit-reversed] */
static ref_tw64 g_tw64 = {0, 0, 0, 0, 0};
static const ref_tw64 *twiddle64(size_t n) {
    if (g_tw64.n >= n) return &g_tw64;
    free(g_tw64.tw); free(g_tw64.twi); free(g_tw64.tw_bo); free(g_tw64.twi_bo);
    g_tw64.n = n;
    g_tw64.tw = malloc(n * sizeof(c64)); g_tw64.twi = malloc(n * sizeof(c64));
    g_tw64.tw_bo = malloc(n * sizeof(c64)); g_tw64.twi_bo = malloc(n * sizeof(c64));
    int lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    for (size_t i = 0; i < n; i++) {
        const double th = ((double)i * M_PI) / (double)n;
        g_tw64.tw[i].re = cos(th); g_tw64.tw[i].im = sin(th);
        g_tw64.twi[i].re = g_tw64.tw[i].re; g_tw64.twi[i].im = -g_tw64.tw[i].im;
    }
    for (size_t i = 0; i < n; i++) {
        size_t r = 0;
        for (int b = 0; b < lg; b++) r |= ((i >> b) & 1) << (lg - 1 - b);
        if (n <= 2) r = i; /* misc.rs:29-42: no-op for len <= 2 */
        g_tw64.tw_bo[i] = g_tw64.tw[r]; g_tw64.twi_bo[i] = g_tw64.twi[r];
    }
    return &g_tw64;
}
/* must be called (single-threaded) before a threaded region uses the f64 path */
void ref_torus_fft64_prepare(size_t n) { (void)twiddle64(n); }

static void fft64_fwd(c64 *a, size_t len, const c64 *tw_bo) { /* fft.rs:9-17 */
    int lg = 0;
    while (((size_t)1 << lg) < len) lg++;
    for (int layer = lg - 1; layer >= 0; layer--) {
        const size_t size = (size_t)1 << layer;
        for (size_t ch = 0; ch * 2 * size < len; ch++) {
            const c64 t = tw_bo[ch];
            c64 *x = a + ch * 2 * size, *y = x + size;
            for (size_t k = 0; k < size; k++) { const c64 tb = c_mul(t, y[k]); const c64 u = x[k]; x[k] = c_add(u, tb); y[k] = c_sub(u, tb); }
        }
    }
}
static void fft64_inv(c64 *a, size_t len, const c64 *twi_bo) { /* fft.rs:21-35 */
    int lg = 0;
    while (((size_t)1 << lg) < len) lg++;
    for (int layer = 0; layer < lg; layer++) {
        const size_t size = (size_t)1 << layer;
        for (size_t ch = 0; ch * 2 * size < len; ch++) {
            const c64 t = twi_bo[ch];
            c64 *x = a + ch * 2 * size, *y = x + size;
            for (size_t k = 0; k < size; k++) { const c64 u = x[k], v = y[k]; x[k] = c_add(u, v); y[k] = c_mul(c_sub(u, v), t); }
        }
    }
    const double n_inv = 1.0 / (double)len;
    for (size_t k = 0; k < len; k++) { a[k].re *= n_inv; a[k].im *= n_inv; }
}
static void to_c64_twisted(const u64 *a, size_t n, const ref_tw64 *T, c64 *out) { /* c64.rs:19-29 */
    const size_t step = T->n / n;
    for (size_t i = 0; i < n / 2; i++) {
        const c64 v = {(double)(int64_t)a[i], (double)(int64_t)a[n / 2 + i]};
        out[i] = c_mul(v, T->tw[i * step]);
    }
}
/* a <- a * b, the reference's floating-point product (c64.rs:11-17, 43-56); scratch: 2 * (n/2) c64 */
static void torus_mul_fft64_ws(u64 *a, const u64 *b, size_t n, c64 *ws) {
    if (n == 1) { a[0] *= b[0]; return; }
    const ref_tw64 *T = twiddle64(n);
    c64 *ca = ws, *cb = ws + n / 2;
    to_c64_twisted(a, n, T, ca);
    to_c64_twisted(b, n, T, cb);
    fft64_fwd(ca, n / 2, T->tw_bo);
    fft64_fwd(cb, n / 2, T->tw_bo);
    for (size_t i = 0; i < n / 2; i++) ca[i] = c_mul(ca[i], cb[i]);
    fft64_inv(ca, n / 2, T->twi_bo);
    const size_t step = T->n / n;
    for (size_t i = 0; i < n / 2; i++) { /* c64.rs:31-41 */
        const c64 c = c_mul(ca[i], T->twi[i * step]);
        a[i] = f64_mod_u64(c.re);
        a[n / 2 + i] = f64_mod_u64(c.im);
    }
}
void ref_torus_mul_fft64(u64 *a, const u64 *b, size_t n) {
    c64 *ws = malloc((n ? n : 1) * sizeof(c64));
    torus_mul_fft64_ws(a, b, n, ws);
    free(ws);
}

/* tggsw.rs:100-112, k = 1: limbs = decompose(a) ++ decompose(b); a' = sum rows_a[l] * limb_l, b' = sum rows_b[l] * limb_l
 * (key polynomial is the lhs of each product).  rows_*: [2d][n].  In place on (ct_a, ct_b).  fft != 0: reference's f64 product. */
static void tggsw_ext_ws(int log_b, int d, const u64 *rows_a, const u64 *rows_b, u64 *ct_a, u64 *ct_b, size_t n, int fft, u64 *ws) {
    u64 *limbs = ws, *oa = ws + (size_t)2 * d * n, *ob = oa + n, *tmp = ob + n;
    c64 *cws = (c64 *)(tmp + n);
    ref_torus_decompose(log_b, d, ct_a, n, limbs);
    ref_torus_decompose(log_b, d, ct_b, n, limbs + (size_t)d * n);
    memset(oa, 0, 2 * n * sizeof(u64));
    for (int l = 0; l < 2 * d; l++) {
        const u64 *limb = limbs + (size_t)l * n;
        for (int c = 0; c < 2; c++) {
            const u64 *row = (c ? rows_b : rows_a) + (size_t)l * n;
            u64 *o = c ? ob : oa;
            if (fft) { memcpy(tmp, row, n * sizeof(u64)); torus_mul_fft64_ws(tmp, limb, n, cws); }
            else ref_torus_mul_exact(row, limb, tmp, n);
            for (size_t i = 0; i < n; i++) o[i] += tmp[i];
        }
    }
    memcpy(ct_a, oa, n * sizeof(u64));
    memcpy(ct_b, ob, n * sizeof(u64));
}
static size_t tggsw_ws_words(int d, size_t n) { return (size_t)2 * d * n + 3 * n + 2 * n /* c64 scratch: n c64 = 2n words */; }

int ref_tggsw_external_product(int log_b, int d, const u64 *rows_a, const u64 *rows_b, u64 *ct_a, u64 *ct_b, size_t n, int fft) {
    if (log_b < 1 || d < 1 || log_b * d > 64 + log_b - 1) return 1;
    u64 *ws = malloc(tggsw_ws_words(d, n) * sizeof(u64));
    if (fft) ref_torus_fft64_prepare(n);
    tggsw_ext_ws(log_b, d, rows_a, rows_b, ct_a, ct_b, n, fft, ws);
    free(ws);
    return 0;
}

/* tfhe/bootstrapping.rs:99-104 */
void ref_tfhe_mod_switch(const u64 *in, u64 *out, size_t count, size_t big_n) {
    int lg = 0;
    while (((size_t)1 << lg) < 2 * big_n) lg++;
    const int bits = 64 - lg;
    for (size_t i = 0; i < count; i++) out[i] = (in[i] + (((u64)1 << bits) >> 1)) >> bits;
}

/* tfhe/bootstrapping.rs:84-96, k = 1: acc = (0, v).rotate(-b); fold cmux(brk_i, acc, acc.rotate(a_i)) (tggsw.rs:114-121).
 * brk_a / brk_b: [n_lwe][2d][n]; a_tilde: [batch][n_lwe] (already mod-switched); out: [batch][n] each. */
int ref_tfhe_blind_rotate(int log_b, int d, const u64 *brk_a, const u64 *brk_b, size_t n_lwe, const u64 *v, const u64 *a_tilde,
                          const u64 *b_tilde, u64 *out_a, u64 *out_b, size_t n, size_t batch, int threads, int fft) {
    if (fft) ref_torus_fft64_prepare(n);
    (void)threads;
#pragma omp parallel for schedule(dynamic) num_threads(threads > 0 ? threads : 1)
    for (size_t c = 0; c < batch; c++) {
        u64 *ws = malloc((tggsw_ws_words(d, n) + 4 * n) * sizeof(u64));
        u64 *acc_a = ws, *acc_b = ws + n, *da = ws + 2 * n, *db = ws + 3 * n, *ext = ws + 4 * n;
        memset(acc_a, 0, n * sizeof(u64));
        ref_torus_monomial_mul(-(int64_t)b_tilde[c], v, acc_b, n);
        for (size_t i = 0; i < n_lwe; i++) {
            const int64_t ai = (int64_t)a_tilde[c * n_lwe + i];
            ref_torus_monomial_mul(ai, acc_a, da, n);
            ref_torus_monomial_mul(ai, acc_b, db, n);
            for (size_t k = 0; k < n; k++) { da[k] -= acc_a[k]; db[k] -= acc_b[k]; }
            tggsw_ext_ws(log_b, d, brk_a + i * 2 * d * n, brk_b + i * 2 * d * n, da, db, n, fft, ext);
            for (size_t k = 0; k < n; k++) { acc_a[k] += da[k]; acc_b[k] += db[k]; }
        }
        memcpy(out_a + c * n, acc_a, n * sizeof(u64));
        memcpy(out_b + c * n, acc_b, n * sizeof(u64));
        free(ws);
    }
    return 0;
}

/* tglwe.rs:115-127 (k = 1) */
void ref_tglwe_sample_extract(const u64 *ct_a, const u64 *ct_b, size_t n, size_t i, u64 *lwe_a, u64 *lwe_b) {
    for (size_t j = 0; j <= i; j++) lwe_a[j] = ct_a[i - j];
    for (size_t j = i + 1; j < n; j++) lwe_a[j] = 0 - ct_a[n + i - j];
    *lwe_b = ct_b[i];
}

/* tlwe.rs:144-153: limbs digit-major (decompose(a).flatten()); ksk_a: [n_in * d][n_out], ksk_b: [n_in * d] */
void ref_tlwe_key_switch(int log_b, int d, const u64 *ksk_a, const u64 *ksk_b, const u64 *ct_a, u64 ct_b, size_t n_in, size_t n_out,
                         u64 *out_a, u64 *out_b) {
    u64 *limbs = malloc((size_t)d * n_in * sizeof(u64));
    ref_torus_decompose(log_b, d, ct_a, n_in, limbs);
    memset(out_a, 0, n_out * sizeof(u64));
    u64 b = 0;
    for (size_t r = 0; r < (size_t)d * n_in; r++) {
        const u64 l = limbs[r];
        const u64 *row = ksk_a + r * n_out;
        for (size_t k = 0; k < n_out; k++) out_a[k] += row[k] * l;
        b += ksk_b[r] * l;
    }
    *out_b = b + ct_b;
    free(limbs);
}

/* tfhe/bootstrapping.rs:78-82: blind rotate -> sample_extract(0) -> key switch, for a batch of TLWE ciphertexts (a: [batch][n_lwe]) */
int ref_tfhe_bootstrap(int log_b, int d, int ks_log_b, int ks_d, const u64 *brk_a, const u64 *brk_b, const u64 *ksk_a, const u64 *ksk_b,
                       size_t n_lwe, const u64 *v, const u64 *lwe_a, const u64 *lwe_b, u64 *out_a, u64 *out_b, size_t n, size_t batch,
                       int threads, int fft) {
    u64 *at = malloc(batch * n_lwe * sizeof(u64)), *bt = malloc(batch * sizeof(u64));
    u64 *ra = malloc(batch * n * sizeof(u64)), *rb = malloc(batch * n * sizeof(u64));
    ref_tfhe_mod_switch(lwe_a, at, batch * n_lwe, n);
    ref_tfhe_mod_switch(lwe_b, bt, batch, n);
    ref_tfhe_blind_rotate(log_b, d, brk_a, brk_b, n_lwe, v, at, bt, ra, rb, n, batch, threads, fft);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
    for (size_t c = 0; c < batch; c++) {
        u64 *ea = malloc(n * sizeof(u64)), eb;
        ref_tglwe_sample_extract(ra + c * n, rb + c * n, n, 0, ea, &eb);
        ref_tlwe_key_switch(ks_log_b, ks_d, ksk_a, ksk_b, ea, eb, n, n_lwe, out_a + c * n_lwe, out_b + c);
        free(ea);
    }
    free(at); free(bt); free(ra); free(rb);
    return 0;
}

/* ==== row T at any TGLWE rank k (the reference's `TglweParam::n`; its own TGLWE / TGGSW tests run at k = 2: tglwe.rs:138-166,
 * tggsw.rs:134-181).  A ciphertext is one buffer [k + 1][n]: a_0 .. a_{k-1}, b (tglwe.rs:49-50); a TGGSW ciphertext is (k + 1) d of
 * them, rows in the order tggsw.rs:80-87 builds.  Exact products only (the checker). ======================================== */
/* tggsw.rs:100-112: limbs = flat_map(decompose) over a_0 .. a_{k-1}, b; out_c = sum_l rows[l][c] * limb_l.  In place on ct. */
static void tggswk_ext_ws(int k, int log_b, int d, const u64 *rows, u64 *ct, size_t n, u64 *ws) {
    const size_t k1 = (size_t)k + 1, R = k1 * d;
    u64 *limbs = ws, *out = ws + R * n, *tmp = out + k1 * n;
    for (size_t c = 0; c < k1; c++) ref_torus_decompose(log_b, d, ct + c * n, n, limbs + c * d * n);
    memset(out, 0, k1 * n * sizeof(u64));
    for (size_t l = 0; l < R; l++)
        for (size_t c = 0; c < k1; c++) {
            ref_torus_mul_exact(rows + (l * k1 + c) * n, limbs + l * n, tmp, n);
            for (size_t i = 0; i < n; i++) out[c * n + i] += tmp[i];
        }
    memcpy(ct, out, k1 * n * sizeof(u64));
}
static size_t tggswk_ws_words(int k, int d, size_t n) { return ((size_t)(k + 1) * d + (size_t)k + 2) * n; }

int ref_tggswk_external_product(int k, int log_b, int d, const u64 *rows, u64 *ct, size_t n) {
    if (k < 1 || log_b < 1 || d < 1 || log_b * d > 64 + log_b - 1) return 1;
    u64 *ws = malloc(tggswk_ws_words(k, d, n) * sizeof(u64));
    tggswk_ext_ws(k, log_b, d, rows, ct, n, ws);
    free(ws);
    return 0;
}

/* tggsw.rs:114-121: out = ct0 + external_product(rows, ct1 - ct0) */
int ref_tggswk_cmux(int k, int log_b, int d, const u64 *rows, const u64 *ct0, const u64 *ct1, u64 *out, size_t n) {
    const size_t w = ((size_t)k + 1) * n;
    u64 *diff = malloc(w * sizeof(u64));
    for (size_t i = 0; i < w; i++) diff[i] = ct1[i] - ct0[i];
    const int rc = ref_tggswk_external_product(k, log_b, d, rows, diff, n);
    for (size_t i = 0; i < w && !rc; i++) out[i] = ct0[i] + diff[i];
    free(diff);
    return rc;
}

/* bootstrapping.rs:84-96: acc = (0, .., 0, v).rotate(-b); fold cmux(brk_i, acc, acc.rotate(a_i)).  brk [n_lwe][(k+1)d][k+1][n],
 * a_tilde [batch][n_lwe] (mod-switched), out [batch][k+1][n] */
int ref_tfhek_blind_rotate(int k, int log_b, int d, const u64 *brk, size_t n_lwe, const u64 *v, const u64 *a_tilde, const u64 *b_tilde, u64 *out,
                           size_t n, size_t batch, int threads) {
    const size_t k1 = (size_t)k + 1, w = k1 * n, per = k1 * d * w;
    (void)threads;
#pragma omp parallel for schedule(dynamic) num_threads(threads > 0 ? threads : 1)
    for (size_t c = 0; c < batch; c++) {
        u64 *ws = malloc((tggswk_ws_words(k, d, n) + 2 * w) * sizeof(u64));
        u64 *acc = ws, *diff = ws + w, *ext = ws + 2 * w;
        memset(acc, 0, w * sizeof(u64));
        ref_torus_monomial_mul(-(int64_t)b_tilde[c], v, acc + (size_t)k * n, n);
        for (size_t i = 0; i < n_lwe; i++) {
            const int64_t ai = (int64_t)a_tilde[c * n_lwe + i];
            for (size_t j = 0; j < k1; j++) ref_torus_monomial_mul(ai, acc + j * n, diff + j * n, n);
            for (size_t x = 0; x < w; x++) diff[x] -= acc[x];
            tggswk_ext_ws(k, log_b, d, brk + i * per, diff, n, ext);
            for (size_t x = 0; x < w; x++) acc[x] += diff[x];
        }
        memcpy(out + c * w, acc, w * sizeof(u64));
        free(ws);
    }
    return 0;
}

/* tglwe.rs:115-127: lwe_a [k n] = concat_j (a_j[..=i].rev() ++ a_j[i+1..].rev().neg()), lwe_b = b[i] */
void ref_tglwek_sample_extract(int k, const u64 *ct, size_t n, size_t i, u64 *lwe_a, u64 *lwe_b) {
    for (int j = 0; j < k; j++) {
        u64 dummy;
        ref_tglwe_sample_extract(ct + (size_t)j * n, ct + (size_t)k * n, n, i, lwe_a + (size_t)j * n, &dummy);
    }
    *lwe_b = ct[(size_t)k * n + i];
}

/* bootstrapping.rs:78-82 at rank k: the key switch takes the TLWE of dimension k n (ksk_a [k n ks_d][n_lwe]) */
int ref_tfhek_bootstrap(int k, int log_b, int d, int ks_log_b, int ks_d, const u64 *brk, const u64 *ksk_a, const u64 *ksk_b, size_t n_lwe, const u64 *v,
                        const u64 *lwe_a, const u64 *lwe_b, u64 *out_a, u64 *out_b, size_t n, size_t batch, int threads) {
    const size_t w = ((size_t)k + 1) * n;
    u64 *at = malloc(batch * n_lwe * sizeof(u64)), *bt = malloc(batch * sizeof(u64)), *acc = malloc(batch * w * sizeof(u64));
    ref_tfhe_mod_switch(lwe_a, at, batch * n_lwe, n);
    ref_tfhe_mod_switch(lwe_b, bt, batch, n);
    ref_tfhek_blind_rotate(k, log_b, d, brk, n_lwe, v, at, bt, acc, n, batch, threads);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
    for (size_t c = 0; c < batch; c++) {
        u64 *ea = malloc((size_t)k * n * sizeof(u64)), eb;
        ref_tglwek_sample_extract(k, acc + c * w, n, 0, ea, &eb);
        ref_tlwe_key_switch(ks_log_b, ks_d, ksk_a, ksk_b, ea, eb, (size_t)k * n, n_lwe, out_a + c * n_lwe, out_b + c);
        free(ea);
    }
    free(at); free(bt); free(acc);
    return 0;
}

int ref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
