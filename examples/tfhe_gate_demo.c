/* Row T from plain C: include/fhe_ring.h and libfhe_ring.so only.  A TGGSW external product (scheme/tfhe/src/tggsw.rs:100-112) at
 * N = 256, base 2^7 x 3 checked against an exact schoolbook computed HERE (wrapping u64: the product mod 2^64), once in the exact mode
 * (bit for bit) and once in the fft64 mode (the reference's own f64 FFT product: within 2d x 2^(64 + log_b + log_n - 53), the bound of
 * util/src/ring/fft/c64.rs:186-208), then a whole gate bootstrap (bootstrapping.rs:78-82) in one call on host buffers.
 * build: gcc -std=c99 -O2 -I include examples/tfhe_gate_demo.c -L learn-fhe_amd/lib -lfhe_ring -Wl,-rpath,$PWD/learn-fhe_amd/lib \
 *            -Wl,--allow-shlib-undefined -o tfhe_gate_demo */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fhe_ring.h"

#define N 256
#define LOG_B 7
#define D 3
#define N_LWE 8
#define BATCH 3

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

/* util/src/misc/decompose.rs:114-135 for T64: signed digits, least significant first */
static void decompose(uint64_t v, int64_t *dig) {
    const int rb = 64 - LOG_B * D;
    uint64_t c = (v + ((1ull << rb) >> 1)) >> rb;
    for (int j = 0; j < D; ++j) {
        const uint64_t limb = c & ((1ull << LOG_B) - 1);
        c >>= LOG_B;
        const uint64_t carry = (((limb - 1) | c) & limb) >> (LOG_B - 1);
        c += carry;
        dig[j] = (int64_t)(limb - (carry << LOG_B));
    }
}

/* out += row * digits (negacyclic, wrapping) */
static void mac(uint64_t *out, const uint64_t *row, const int64_t *dig) {
    for (int i = 0; i < N; ++i) {
        if (!dig[i]) continue;
        for (int j = 0; j < N; ++j) {
            const uint64_t p = row[j] * (uint64_t)dig[i];
            if (i + j < N) out[i + j] += p; else out[i + j - N] -= p;
        }
    }
}

int main(void) {
    fhe_torus_ctx *t = NULL;
    int rc = fhe_torus_ctx_create(0, &t);
    if (rc != FHE_OK) { fprintf(stderr, "fhe_torus_ctx_create: %d (hip %d)\n", rc, fhe_last_hip_error()); return 1; }
    const size_t rows = (size_t)N_LWE * 2 * D * N;
    uint64_t *ra = malloc(rows * 8), *rb = malloc(rows * 8);
    for (size_t i = 0; i < rows; ++i) { ra[i] = rnd(); rb[i] = rnd(); }
    uint64_t ca[N], cb[N], ea[N], eb[N];
    for (int i = 0; i < N; ++i) { ca[i] = rnd(); cb[i] = rnd(); ea[i] = eb[i] = 0; }
    /* the exact product, here: limbs = decompose(a) ++ decompose(b); a' = sum rows_a[l] * limb_l, b' = sum rows_b[l] * limb_l */
    static int64_t dig[2 * D][N];
    for (int i = 0; i < N; ++i) {
        int64_t d0[D], d1[D];
        decompose(ca[i], d0); decompose(cb[i], d1);
        for (int j = 0; j < D; ++j) { dig[j][i] = d0[j]; dig[D + j][i] = d1[j]; }
    }
    for (int l = 0; l < 2 * D; ++l) { mac(ea, ra + (size_t)l * N, dig[l]); mac(eb, rb + (size_t)l * N, dig[l]); }

    fhe_tggsw_key *exact = NULL, *fft = NULL;
    rc = fhe_tggsw_prepare(t, LOG_B, D, ra, rb, N, N_LWE, FHE_MEM_HOST, &exact);
    if (rc == FHE_OK) rc = fhe_tggsw_prepare_fft64(t, LOG_B, D, ra, rb, N, N_LWE, FHE_MEM_HOST, &fft);
    if (rc != FHE_OK) { fprintf(stderr, "prepare: %d\n", rc); return 1; }
    uint64_t xa[N], xb[N];
    memcpy(xa, ca, sizeof xa); memcpy(xb, cb, sizeof xb);
    rc = fhe_tggsw_external_product(t, exact, 0, xa, xb, 1, FHE_MEM_HOST, NULL);
    if (rc != FHE_OK || memcmp(xa, ea, sizeof xa) || memcmp(xb, eb, sizeof xb)) { fprintf(stderr, "exact external product != schoolbook (rc %d)\n", rc); return 1; }
    memcpy(xa, ca, sizeof xa); memcpy(xb, cb, sizeof xb);
    rc = fhe_tggsw_external_product(t, fft, 0, xa, xb, 1, FHE_MEM_HOST, NULL);
    const int64_t bound = 2 * D * (1ll << (64 + LOG_B + 8 - 53));
    int64_t worst = 0;
    for (int i = 0; i < N; ++i) {
        int64_t e0 = (int64_t)(xa[i] - ea[i]), e1 = (int64_t)(xb[i] - eb[i]);
        if (e0 < 0) e0 = -e0;
        if (e1 < 0) e1 = -e1;
        if (e0 > worst) worst = e0;
        if (e1 > worst) worst = e1;
    }
    if (rc != FHE_OK || worst > bound) { fprintf(stderr, "fft64 external product: rc %d, error %lld > bound %lld\n", rc, (long long)worst, (long long)bound); return 1; }

    /* the whole gate on host buffers: mod switch, N_LWE CMUXes, sample extract, key switch (base 2^4 x 5) */
    const int ks_lb = 4, ks_d = 5;
    uint64_t *ksa = malloc((size_t)N * ks_d * N_LWE * 8), *ksb = malloc((size_t)N * ks_d * 8);
    for (size_t i = 0; i < (size_t)N * ks_d * N_LWE; ++i) ksa[i] = rnd();
    for (size_t i = 0; i < (size_t)N * ks_d; ++i) ksb[i] = rnd();
    uint64_t v[N], la[BATCH * N_LWE], lb[BATCH], oa[BATCH * N_LWE], ob[BATCH], oa2[BATCH * N_LWE], ob2[BATCH];
    for (int i = 0; i < N; ++i) v[i] = rnd();
    for (int i = 0; i < BATCH * N_LWE; ++i) la[i] = rnd();
    for (int i = 0; i < BATCH; ++i) lb[i] = rnd();
    rc = fhe_tfhe_bootstrap(t, exact, ks_lb, ks_d, ksa, ksb, v, la, lb, oa, ob, BATCH, FHE_MEM_HOST, NULL);
    if (rc == FHE_OK) rc = fhe_tfhe_bootstrap(t, exact, ks_lb, ks_d, ksa, ksb, v, la, lb, oa2, ob2, BATCH, FHE_MEM_HOST, NULL);
    if (rc != FHE_OK || memcmp(oa, oa2, sizeof oa) || memcmp(ob, ob2, sizeof ob)) { fprintf(stderr, "gate bootstrap: rc %d or not reproducible\n", rc); return 1; }
    rc = fhe_tfhe_bootstrap(t, fft, ks_lb, ks_d, ksa, ksb, v, la, lb, oa2, ob2, BATCH, FHE_MEM_HOST, NULL);
    if (rc != FHE_OK) { fprintf(stderr, "fft64 gate bootstrap: %d\n", rc); return 1; }
    /* a key of another context is refused, as is an index past the key */
    fhe_torus_ctx *t2 = NULL;
    if (fhe_torus_ctx_create(0, &t2) != FHE_OK || fhe_tggsw_external_product(t2, exact, 0, xa, xb, 1, FHE_MEM_HOST, NULL) != FHE_ERR_INVALID ||
        fhe_tggsw_external_product(t, exact, N_LWE, xa, xb, 1, FHE_MEM_HOST, NULL) != FHE_ERR_INVALID) { fprintf(stderr, "status codes\n"); return 1; }
    printf("tfhe_gate_demo ok: exact external product == schoolbook; fft64 within %lld of it (bound %lld); gate bootstrap on host buffers\n", (long long)worst,
           (long long)bound);
    fhe_tggsw_key_destroy(exact); fhe_tggsw_key_destroy(fft);
    fhe_torus_ctx_destroy(t2); fhe_torus_ctx_destroy(t);
    free(ra); free(rb); free(ksa); free(ksb);
    return 0;
}
