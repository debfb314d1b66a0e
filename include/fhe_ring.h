/* fhe_ring.h -- C ABI of the MI355X (gfx950) polynomial-ring backend.
 *
 * Drop-in boundary for the ring hot path of han0110/learn-fhe's `util` crate.  The reference has no
 * FFI layer; the seam these entry points replace is the three crate-private functions in
 * util/src/ring/fft/zq.rs:14-36 (called only from util/src/ring.rs:140-144, 180-184, 256-264) plus the
 * pure helpers listed per function below.  A Rust `util` fork would bind them with one `extern "C"`
 * block (INTEGRATION.md shows it).
 *
 * Conventions
 *   - all ring data are flat little-endian uint64_t arrays of fully reduced values in [0, q);
 *     a batch of polynomials is contiguous, [batch][n];
 *   - `mem` says where every data pointer of the call lives: FHE_MEM_DEVICE = HBM of the context's device
 *     (the measured path: nothing is copied), FHE_MEM_HOST = pageable host memory (the library stages
 *     through its own device buffers and synchronises before returning);
 *   - `stream` is a hipStream_t (NULL = the default stream); device calls are asynchronous on it;
 *   - every function returns an int status (FHE_OK == 0) where the reference would panic; nothing aborts;
 *   - contexts and key handles are immutable after creation and may be shared between host threads.
 *   - there is NO CPU compute fallback: a context created with device < 0 only serves the host-side
 *     queries (primes, generator, twiddles) and every compute call on it returns FHE_ERR_NO_DEVICE.
 */
#ifndef FHE_RING_H
#define FHE_RING_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    FHE_OK = 0,
    FHE_ERR_INVALID = 1,      /* null pointer, n not a power of two, bad parameter (ring.rs:62 assert) */
    FHE_ERR_NOT_PRIME = 2,    /* NTT requested for a non-prime modulus (fft/zq.rs:44: missing map key) */
    FHE_ERR_NO_ROOT = 3,      /* n > 2^(s-1): no primitive 2n-th root (fft.rs:45 slice out of bounds) */
    FHE_ERR_MODULUS = 4,      /* operands over different moduli (zq.rs:145/177/185/193 assert_eq) */
    FHE_ERR_HIP = 5,          /* a HIP runtime call failed; fhe_last_hip_error() has the code */
    FHE_ERR_UNSUPPORTED = 6,  /* valid in the reference, outside what this build implements (q >= 2^62, n > 2^17) */
    FHE_ERR_NO_DEVICE = 7     /* compute call on a host-only context */
};

typedef enum { FHE_MEM_HOST = 0, FHE_MEM_DEVICE = 1 } fhe_mem;

typedef struct fhe_ctx fhe_ctx; /* one prime modulus: s, g, omega, twiddle tables (host + HBM) */

const char *fhe_version(void);
int fhe_last_hip_error(void);

/* ---- scalar / setup (host only) -------------------------------------------------------------- */
/* util/src/zq.rs:337-342 `is_prime` */
int fhe_is_prime(uint64_t q);
/* util/src/zq.rs:325-329 `two_adic_primes(bits, log_n).take(count)`; returns how many were written */
int fhe_two_adic_primes(int bits, int log_n, int count, uint64_t *out);

/* util/src/ring/fft/zq.rs:49-67 `twiddle(q)` / `compute_twiddle`: validates q prime, derives
 * s = trailing_zeros(q-1), g = smallest quadratic non-residue (zq.rs:99-105), omega = g^((q-1)>>s),
 * and the bit-reversed tables.  device >= 0: tables are uploaded to that GPU; device < 0: host-only. */
int fhe_ctx_create(uint64_t q, int device, fhe_ctx **out);
void fhe_ctx_destroy(fhe_ctx *ctx);
int fhe_ctx_info(const fhe_ctx *ctx, uint64_t *q, int *s, uint64_t *g, uint64_t *omega);
/* first `count` entries of the reference's twiddle table (inverse != 0: the inverse table) */
int fhe_ctx_twiddles(const fhe_ctx *ctx, int inverse, uint64_t *out, size_t count);

/* ---- transforms ------------------------------------------------------------------------------ */
/* util/src/ring/fft/zq.rs:27-30 `nega_cyclic_ntt_in_place` (= ring.rs:140-144 `to_evaluation`):
 * natural-order coefficients -> bit-reversed-order evaluations, in place, for `batch` polynomials. */
int fhe_ntt_fwd(const fhe_ctx *ctx, uint64_t *a, size_t n, size_t batch, fhe_mem mem, void *stream);
/* util/src/ring/fft/zq.rs:32-36 `nega_cyclic_intt_in_place` (= ring.rs:180-184 `to_coefficient`). */
int fhe_ntt_inv(const fhe_ctx *ctx, uint64_t *a, size_t n, size_t batch, fhe_mem mem, void *stream);
/* util/src/ring/fft/zq.rs:14-19 `nega_cyclic_ntt_mul_assign` (= ring.rs:256-264 `Rq *= &Rq`):
 * a[k] <- a[k] * b[k] in Z_q[X]/(X^n+1), coefficient domain in and out. */
int fhe_ntt_mul(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, size_t n, size_t batch, fhe_mem mem,
                void *stream);
/* util/src/ring.rs:266-270 evaluation-domain `MulAssign`: a[i] <- a[i] * b[i] mod q, len elements. */
int fhe_pointwise_mul(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, size_t len, fhe_mem mem, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FHE_RING_H */
