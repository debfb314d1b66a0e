/* SURVEY.md section 8(e) at the C boundary: ONE process drives every visible GPU.  A batch of independent polynomials
 * (BASELINE config 2's ring: N = 2^14, q = 1152921504606748673) is split contiguously over the devices; each device gets its
 * own fhe_ctx (twiddles replicated, 2 x 16 B x 2^14 per direction), its own HIP stream and its own shard in its own HBM; the
 * forward transforms run concurrently and there is NO data-path collective.  The final gather is the CALLER's: here plain
 * hipMemcpyAsync into one host buffer (a device-resident consumer would use hipMemcpyPeerAsync / an RCCL all-gather in its own
 * process group -- the library deliberately links neither: SURVEY.md 8(e), "collective: none during compute").
 * On a one-GPU box the same code runs with the shards on two streams of that device (SHARDS_PER_DEVICE), which still exercises
 * contexts, streams and asynchronous entry points from one host thread.  Checked bit for bit against device 0 transforming the
 * whole batch in one call.
 * build: gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/multi_gpu_demo.c -L learn-fhe_amd/lib -lfhe_ring \
 *            -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/learn-fhe_amd/lib -Wl,-rpath,/opt/rocm/lib -o multi_gpu_demo */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fhe_ring.h"

#define N 16384
#define BATCH 64
#define MAX_SHARDS 16
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_FHE(x) do { int rc_ = (x); if (rc_ != FHE_OK) { fprintf(stderr, "%s: status %d (hip %d)\n", #x, rc_, fhe_last_hip_error()); return 1; } } while (0)

int main(void) {
    const uint64_t q = 1152921504606748673ull; /* two_adic_primes(60, 15).next() */
    int n_dev = 0;
    CHECK_HIP(hipGetDeviceCount(&n_dev));
    if (n_dev < 1) { fprintf(stderr, "no GPU\n"); return 1; }
    const int per_dev = n_dev == 1 ? 2 : 1, shards = n_dev * per_dev > MAX_SHARDS ? MAX_SHARDS : n_dev * per_dev;
    uint64_t *in = malloc((size_t)BATCH * N * 8), *out = malloc((size_t)BATCH * N * 8), *ref = malloc((size_t)BATCH * N * 8);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < (size_t)BATCH * N; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; in[i] = s % q; }

    /* reference: the whole batch on device 0, host-memory call */
    fhe_ctx *ctx0 = NULL;
    CHECK_FHE(fhe_ctx_create(q, 0, &ctx0));
    memcpy(ref, in, (size_t)BATCH * N * 8);
    CHECK_FHE(fhe_ntt_fwd(ctx0, ref, N, BATCH, FHE_MEM_HOST, NULL));

    fhe_ctx *ctx[MAX_SHARDS];
    hipStream_t st[MAX_SHARDS];
    uint64_t *d[MAX_SHARDS];
    size_t lo[MAX_SHARDS + 1];
    for (int r = 0; r <= shards; ++r) lo[r] = (size_t)BATCH * r / shards; /* contiguous shards (learn-fhe_amd/shard.py: shard_range) */
    for (int r = 0; r < shards; ++r) {
        const int dev = r / per_dev;
        const size_t cnt = lo[r + 1] - lo[r];
        CHECK_HIP(hipSetDevice(dev));
        CHECK_HIP(hipStreamCreateWithFlags(&st[r], hipStreamNonBlocking));
        CHECK_FHE(fhe_ctx_create(q, dev, &ctx[r]));
        CHECK_HIP(hipMalloc((void **)&d[r], cnt * N * 8));
        /* everything below is asynchronous on the shard's stream: upload, transform, gather */
        CHECK_HIP(hipMemcpyAsync(d[r], in + lo[r] * N, cnt * N * 8, hipMemcpyHostToDevice, st[r]));
        CHECK_FHE(fhe_ntt_fwd(ctx[r], d[r], N, cnt, FHE_MEM_DEVICE, st[r]));
        CHECK_HIP(hipMemcpyAsync(out + lo[r] * N, d[r], cnt * N * 8, hipMemcpyDeviceToHost, st[r])); /* the final gather */
    }
    for (int r = 0; r < shards; ++r) {
        CHECK_HIP(hipSetDevice(r / per_dev));
        CHECK_HIP(hipStreamSynchronize(st[r]));
    }
    size_t bad = 0;
    for (size_t i = 0; i < (size_t)BATCH * N; ++i) bad += out[i] != ref[i];
    for (int r = 0; r < shards; ++r) {
        CHECK_HIP(hipSetDevice(r / per_dev));
        CHECK_HIP(hipFree(d[r]));
        CHECK_HIP(hipStreamDestroy(st[r]));
        fhe_ctx_destroy(ctx[r]);
    }
    fhe_ctx_destroy(ctx0);
    if (bad) { fprintf(stderr, "multi_gpu_demo: %zu mismatches\n", bad); return 1; }
    printf("multi_gpu_demo ok: %d device(s), %d shard(s) of %d polynomials (N=%d), sharded result == single-device result\n", n_dev, shards, BATCH, N);
    return 0;
}
