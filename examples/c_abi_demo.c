/* The drop-in boundary from plain C: no Python, no torch -- include/fhe_ring.h and libfhe_ring.so only.
 * BASELINE config 1 (one forward NTT, N = 1024, q = 1073707009, the "plumbing" case) plus a ring product checked against a
 * schoolbook product computed here, through host-memory calls (FHE_MEM_HOST).
 * build: gcc -std=c99 -O2 -I include examples/c_abi_demo.c -L learn-fhe_amd/lib -lfhe_ring -Wl,-rpath,$PWD/learn-fhe_amd/lib \
 *            -Wl,--allow-shlib-undefined -o c_abi_demo */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "fhe_ring.h"

#define N 1024

static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((unsigned __int128)a * b % q); }

int main(void) {
    const uint64_t q = 1073707009ull; /* two_adic_primes(30, 11).next() */
    fhe_ctx *ctx = NULL;
    int rc = fhe_ctx_create(q, 0, &ctx);
    if (rc != FHE_OK) { fprintf(stderr, "fhe_ctx_create: %d (hip %d)\n", rc, fhe_last_hip_error()); return 1; }
    uint64_t *a = malloc(N * 8), *b = malloc(N * 8), *c = malloc(N * 8), *ref = calloc(N, 8), *saved = malloc(N * 8);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < N; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; a[i] = s % q;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; b[i] = s % q;
        c[i] = a[i]; saved[i] = a[i];
    }
    /* negacyclic schoolbook product (util/src/ring.rs:421-440 is the reference's own check) */
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            const uint64_t p = mulmod(a[i], b[j], q);
            const int k = (i + j) % N;
            ref[k] = (i + j < N) ? (ref[k] + p) % q : (ref[k] + q - p) % q;
        }
    rc = fhe_ntt_mul(ctx, c, b, N, 1, FHE_MEM_HOST, NULL);
    if (rc != FHE_OK) { fprintf(stderr, "fhe_ntt_mul: %d\n", rc); return 1; }
    for (int i = 0; i < N; ++i)
        if (c[i] != ref[i]) { fprintf(stderr, "ring product mismatch at %d\n", i); return 1; }
    rc = fhe_ntt_fwd(ctx, a, N, 1, FHE_MEM_HOST, NULL);
    if (rc == FHE_OK) rc = fhe_ntt_inv(ctx, a, N, 1, FHE_MEM_HOST, NULL);
    if (rc != FHE_OK) { fprintf(stderr, "transform: %d\n", rc); return 1; }
    for (int i = 0; i < N; ++i)
        if (a[i] != saved[i]) { fprintf(stderr, "round trip mismatch at %d\n", i); return 1; }
    /* the reference panics on a non-prime modulus (fft/zq.rs:44); the ABI returns a status */
    fhe_ctx *bad = NULL;
    if (fhe_ctx_create(1073707011ull, 0, &bad) == FHE_OK) { fprintf(stderr, "non-prime modulus accepted\n"); return 1; }
    fhe_ctx_destroy(ctx);
    printf("c_abi_demo ok: %s, N=%d q=%llu ring product == schoolbook, round trip == identity\n", fhe_version(), N, (unsigned long long)q);
    return 0;
}
