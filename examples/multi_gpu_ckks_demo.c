/* SURVEY.md section 8(e) row 3 at the C boundary: BASELINE config 4's key switch (N = 2^15, 8 + 8 sixty-bit primes) with its RNS
 * limbs sharded over the visible GPUs, driven by ONE process.  Every shard holds the context and the prepared key (replicated)
 * and owns K / S q-limbs and K / S p-limbs; per batch of ciphertexts
 *     stage 1  fhe_ckks_shard_products   local, asynchronous on the shard's stream
 *     gather   every shard's p-limb products into every shard's gather buffer: hipMemcpyPeerAsync on the PRODUCER's stream, an
 *              event per producer, hipStreamWaitEvent on every consumer -- the one exchange of the path.  (A multi-process
 *              deployment replaces exactly this block by one RCCL all-gather; the library links neither.)
 *     stage 2  fhe_ckks_shard_finish     local
 * and no host synchronisation in between.  On a one-GPU box the shards are streams of that device (still S contexts' worth of
 * limbs, S streams, the same copies).  Checked bit for bit against fhe_ckks_key_switch on device 0.
 * build: gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/multi_gpu_ckks_demo.c -L learn-fhe_amd/lib \
 *            -lfhe_ring -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/learn-fhe_amd/lib -Wl,-rpath,/opt/rocm/lib -o multi_gpu_ckks_demo */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fhe_ring.h"

#define LOG_N 15
#define N (1u << LOG_N)
#define BIG_L 8
#define BIG_K 8
#define BATCH 4
#define MAX_SHARDS 8
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_FHE(x) do { int rc_ = (x); if (rc_ != FHE_OK) { fprintf(stderr, "%s: status %d (hip %d)\n", #x, rc_, fhe_last_hip_error()); return 1; } } while (0)

static uint64_t rng_s = 88172645463325252ull;
static uint64_t rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }

int main(void) {
    int n_dev = 0;
    CHECK_HIP(hipGetDeviceCount(&n_dev));
    if (n_dev < 1) { fprintf(stderr, "no GPU\n"); return 1; }
    /* shards: a power of two that divides K; one per device, or two streams of the one device */
    int shards = n_dev == 1 ? 2 : (n_dev >= 8 ? 8 : n_dev >= 4 ? 4 : 2);
    const int per_dev = n_dev == 1 ? shards : 1, nq = BIG_L / shards, np = BIG_K / shards;
    uint64_t primes[BIG_L + BIG_K];
    if (fhe_two_adic_primes(60, LOG_N + 1, BIG_L + BIG_K, primes) != BIG_L + BIG_K) { fprintf(stderr, "primes\n"); return 1; }
    const uint64_t *qs = primes, *ps = primes + BIG_L;
    const size_t limb = (size_t)N, ct_words = (size_t)BATCH * BIG_L * limb, key_words = (size_t)(BIG_L + BIG_K) * limb;
    uint64_t *ksk_b = malloc(key_words * 8), *ksk_a = malloc(key_words * 8), *ct_b = malloc(ct_words * 8), *ct_a = malloc(ct_words * 8);
    uint64_t *ref_b = malloc(ct_words * 8), *ref_a = malloc(ct_words * 8), *out_b = malloc(ct_words * 8), *out_a = malloc(ct_words * 8);
    for (int l = 0; l < BIG_L + BIG_K; ++l)
        for (size_t i = 0; i < limb; ++i) { ksk_b[l * limb + i] = rnd() % primes[l]; ksk_a[l * limb + i] = rnd() % primes[l]; }
    for (int c = 0; c < BATCH; ++c)
        for (int l = 0; l < BIG_L; ++l)
            for (size_t i = 0; i < limb; ++i) { ct_b[(c * BIG_L + l) * limb + i] = rnd() % qs[l]; ct_a[(c * BIG_L + l) * limb + i] = rnd() % qs[l]; }

    /* reference: the whole key switch on device 0 (host-memory call, in place on copies) */
    fhe_rns_ctx *rns0 = NULL;
    fhe_ckks_key *key0 = NULL;
    CHECK_HIP(hipSetDevice(0));
    CHECK_FHE(fhe_rns_ctx_create(qs, BIG_L, ps, BIG_K, 0, &rns0));
    CHECK_FHE(fhe_ckks_ksk_prepare(rns0, ksk_b, ksk_a, N, FHE_MEM_HOST, &key0));
    memcpy(ref_b, ct_b, ct_words * 8); memcpy(ref_a, ct_a, ct_words * 8);
    CHECK_FHE(fhe_ckks_key_switch(rns0, key0, ref_b, ref_a, BATCH, FHE_MEM_HOST, NULL));

    fhe_rns_ctx *rns[MAX_SHARDS];
    fhe_ckks_key *key[MAX_SHARDS];
    fhe_ckks_shard *sh[MAX_SHARDS];
    hipStream_t st[MAX_SHARDS];
    hipEvent_t produced[MAX_SHARDS];
    uint64_t *d_a[MAX_SHARDS], *d_b[MAX_SHARDS], *d_pq[MAX_SHARDS], *d_pp[MAX_SHARDS], *d_gather[MAX_SHARDS], *d_ob[MAX_SHARDS], *d_oa[MAX_SHARDS];
    const size_t pq_words = 2 * (size_t)BATCH * nq * limb, pp_words = 2 * (size_t)BATCH * np * limb, own_words = (size_t)BATCH * nq * limb;
    for (int r = 0; r < shards; ++r) {
        const int dev = r / per_dev;
        CHECK_HIP(hipSetDevice(dev));
        CHECK_HIP(hipStreamCreateWithFlags(&st[r], hipStreamNonBlocking));
        CHECK_HIP(hipEventCreateWithFlags(&produced[r], hipEventDisableTiming));
        CHECK_FHE(fhe_rns_ctx_create(qs, BIG_L, ps, BIG_K, dev, &rns[r]));
        CHECK_FHE(fhe_ckks_ksk_prepare(rns[r], ksk_b, ksk_a, N, FHE_MEM_HOST, &key[r]));
        CHECK_FHE(fhe_ckks_shard_create(rns[r], key[r], r * nq, (r + 1) * nq, r * np, (r + 1) * np, &sh[r]));
        CHECK_HIP(hipMalloc((void **)&d_a[r], ct_words * 8)); CHECK_HIP(hipMalloc((void **)&d_b[r], own_words * 8));
        CHECK_HIP(hipMalloc((void **)&d_pq[r], pq_words * 8)); CHECK_HIP(hipMalloc((void **)&d_pp[r], pp_words * 8));
        CHECK_HIP(hipMalloc((void **)&d_gather[r], (size_t)shards * pp_words * 8));
        CHECK_HIP(hipMalloc((void **)&d_ob[r], own_words * 8)); CHECK_HIP(hipMalloc((void **)&d_oa[r], own_words * 8));
        /* staging: ct.a replicated, ct.b's owned limbs */
        CHECK_HIP(hipMemcpyAsync(d_a[r], ct_a, ct_words * 8, hipMemcpyHostToDevice, st[r]));
        for (int c = 0; c < BATCH; ++c)
            CHECK_HIP(hipMemcpyAsync(d_b[r] + (size_t)c * nq * limb, ct_b + ((size_t)c * BIG_L + r * nq) * limb, (size_t)nq * limb * 8, hipMemcpyHostToDevice, st[r]));
    }
    /* stage 1 on every shard, then each producer pushes its p-limb products into slot r of EVERY shard's gather buffer */
    for (int r = 0; r < shards; ++r) {
        CHECK_HIP(hipSetDevice(r / per_dev));
        CHECK_FHE(fhe_ckks_shard_products(sh[r], d_a[r], d_pq[r], d_pp[r], BATCH, FHE_MEM_DEVICE, st[r]));
        for (int t = 0; t < shards; ++t)
            CHECK_HIP(hipMemcpyPeerAsync(d_gather[t] + (size_t)r * pp_words, t / per_dev, d_pp[r], r / per_dev, pp_words * 8, st[r]));
        CHECK_HIP(hipEventRecord(produced[r], st[r]));
    }
    /* stage 2: a consumer waits for every producer's event (device side; the host does not block), then rescales its limbs */
    for (int t = 0; t < shards; ++t) {
        CHECK_HIP(hipSetDevice(t / per_dev));
        for (int r = 0; r < shards; ++r) CHECK_HIP(hipStreamWaitEvent(st[t], produced[r], 0));
        CHECK_FHE(fhe_ckks_shard_finish(sh[t], d_pq[t], d_gather[t], d_b[t], d_ob[t], d_oa[t], BATCH, FHE_MEM_DEVICE, st[t]));
        for (int c = 0; c < BATCH; ++c) {  /* the final gather of the output limbs is the consumer's choice: here back to the host */
            CHECK_HIP(hipMemcpyAsync(out_b + ((size_t)c * BIG_L + t * nq) * limb, d_ob[t] + (size_t)c * nq * limb, (size_t)nq * limb * 8, hipMemcpyDeviceToHost, st[t]));
            CHECK_HIP(hipMemcpyAsync(out_a + ((size_t)c * BIG_L + t * nq) * limb, d_oa[t] + (size_t)c * nq * limb, (size_t)nq * limb * 8, hipMemcpyDeviceToHost, st[t]));
        }
    }
    for (int r = 0; r < shards; ++r) {
        CHECK_HIP(hipSetDevice(r / per_dev));
        CHECK_HIP(hipStreamSynchronize(st[r]));
    }
    size_t bad = 0;
    for (size_t i = 0; i < ct_words; ++i) bad += (out_b[i] != ref_b[i]) + (out_a[i] != ref_a[i]);
    for (int r = 0; r < shards; ++r) {
        CHECK_HIP(hipSetDevice(r / per_dev));
        fhe_ckks_shard_destroy(sh[r]); fhe_ckks_key_destroy(key[r]); fhe_rns_ctx_destroy(rns[r]);
        CHECK_HIP(hipFree(d_a[r])); CHECK_HIP(hipFree(d_b[r])); CHECK_HIP(hipFree(d_pq[r])); CHECK_HIP(hipFree(d_pp[r]));
        CHECK_HIP(hipFree(d_gather[r])); CHECK_HIP(hipFree(d_ob[r])); CHECK_HIP(hipFree(d_oa[r]));
        CHECK_HIP(hipEventDestroy(produced[r])); CHECK_HIP(hipStreamDestroy(st[r]));
    }
    fhe_ckks_key_destroy(key0); fhe_rns_ctx_destroy(rns0);
    if (bad) { fprintf(stderr, "multi_gpu_ckks_demo: %zu mismatches\n", bad); return 1; }
    printf("multi_gpu_ckks_demo ok: %d device(s), %d shard(s) x (%d q-limb(s) + %d p-limb(s)), batch %d at N=2^%d: limb-sharded key switch == single-device key switch\n",
           n_dev, shards, nq, np, BATCH, LOG_N);
    return 0;
}
