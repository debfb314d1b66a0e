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
/* Device scratch of the entry points comes from a private stream-ordered pool per device (never the device's default pool);
 * freed blocks are kept for the next call up to 4 GiB.  fhe_trim() returns everything that is not in use to the driver. */
int fhe_trim(void);
/* Lab switches.  Routes that a faster one replaced stay in the library so that tests can compare the two bit for bit; NONE changes a
 * result.  Each is read ONCE from the environment (FHE_RING_<NAME>) and afterwards only through this call: the library never calls
 * getenv on a call path.  Names: "NO_EDGE" (key switch at N = 2^15 on whole transforms), "NO_LIMB_MAJOR" (linear dispatch order over
 * several moduli), "NO_W12" (2^12 / 2^13 rings on the generic kernels), "NO_FUSED_MUL" (ring product as forward + multiplying
 * inverse), "SMALL_BATCH" (FHEW: 4 coefficients per lane up to this batch, 8 above; -1 = the library's rule), "NO_F64_EXACT" (TFHE: eligible keys on the three-prime integer path instead of the three-piece f64 one), "NO_PACKED_DIGITS" (TFHE blind
 * rotation: digits decomposed once per prime instead of once per CMUX).  Unknown name:
 * FHE_ERR_INVALID. */
int fhe_set_option(const char *name, long value);

/* Entry points that take a modulus instead of a context (fhe_rq_*, fhe_decompose, fhe_automorphism, fhe_monomial_mul,
 * fhe_lwe_*, fhe_rlwe_sample_extract, fhe_torus_decompose, fhe_tfhe_mod_switch, fhe_tglwe_sample_extract, fhe_tlwe_key_switch)
 * run on the device their FHE_MEM_DEVICE operands live on, whatever the caller's current device is; `stream` must belong to
 * that device.  With FHE_MEM_HOST they run on the current device. */

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
/* util/src/ring.rs:328-358 `Rq` + / - / unary - and 359-366 scalar `*= Zq` (zq.rs:156-196), element-wise over `len`
 * values of any modulus q < 2^62 (either basis: the operations are the same on coefficients and on evaluations);
 * `out` may alias an input.  fhe_rq_from_i64: util/src/zq.rs:63-69 `Zq::from_i64` (rem_euclid), the conversion
 * `Rq *= &AVec<i64>` (ring.rs:272-282) applies to its right-hand side before the product. */
int fhe_rq_add(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t len, fhe_mem mem, void *stream);
int fhe_rq_sub(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t len, fhe_mem mem, void *stream);
int fhe_rq_neg(uint64_t q, const uint64_t *a, uint64_t *out, size_t len, fhe_mem mem, void *stream);
int fhe_rq_scalar_mul(uint64_t q, const uint64_t *a, uint64_t scalar, uint64_t *out, size_t len, fhe_mem mem, void *stream);
int fhe_rq_from_i64(uint64_t q, const int64_t *in, uint64_t *out, size_t len, fhe_mem mem, void *stream);
/* util/src/ring.rs:266-270 evaluation-domain `MulAssign`: a[i] <- a[i] * b[i] mod q, len elements. */
int fhe_pointwise_mul(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, size_t len, fhe_mem mem, void *stream);

/* ---- gadget decomposition, automorphism, monomial (any modulus q < 2^62, on the current HIP device) ---- */
/* util/src/misc/decompose.rs:42-46 `Base2Decomposor::<Zq>::new(q, log_b, d).decompose(poly)`:
 * in [polys][n] -> out [polys][d][n], digit j of coefficient i of polynomial p at out[(p*d + j)*n + i],
 * least-significant digit first, digits in [-B/2, B/2] mod q exactly as decompose.rs:91-112 produces them. */
int fhe_decompose(uint64_t q, int log_b, int d, const uint64_t *in, size_t n, size_t polys, uint64_t *out, fhe_mem mem,
                  void *stream);
/* util/src/avec.rs:34-50 `automorphism(t)`: X -> X^t, t taken mod 2n; out must not alias in. */
int fhe_automorphism(uint64_t q, int64_t t, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem,
                     void *stream);
/* util/src/ring.rs:299-313 `Rq *= X^k`, k taken mod 2n; out must not alias in. */
int fhe_monomial_mul(uint64_t q, int64_t k, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem,
                     void *stream);

/* ---- prepared gadget keys (device resident, evaluation domain) ---------------------------------------- */
typedef struct fhe_key fhe_key;
/* `count` RGSW ciphertexts (scheme/fhew/src/rgsw.rs:37-47, 84-105): rows_a / rows_b = the a / b polynomials of the
 * 2d RLWE rows of each, [count][2d][n], coefficient domain.  n = 128 .. 2048. */
int fhe_rgsw_prepare(const fhe_ctx *ctx, int log_b, int d, const uint64_t *rows_a, const uint64_t *rows_b, size_t n,
                     size_t count, fhe_mem mem, fhe_key **out);
/* `count` RLWE key-switching / automorphism keys (scheme/fhew/src/rlwe.rs:43-66, 109-132): [count][d][n]. */
int fhe_ksk_prepare(const fhe_ctx *ctx, int log_b, int d, const uint64_t *rows_a, const uint64_t *rows_b, size_t n,
                    size_t count, fhe_mem mem, fhe_key **out);
void fhe_key_destroy(fhe_key *key);

/* scheme/fhew/src/rgsw.rs:116-128 `Rgsw::external_product(param, rgsw[index], ct)` for `batch` RLWE ciphertexts
 * (ct_a, ct_b: [batch][n], coefficient domain, in place). */
int fhe_external_product(const fhe_ctx *ctx, const fhe_key *rgsw, size_t index, uint64_t *ct_a, uint64_t *ct_b,
                         size_t batch, fhe_mem mem, void *stream);
/* scheme/fhew/src/rgsw.rs:130-150 `Rgsw::internal_product(param, ct0 = rgsw[index], ct1)`: RGSW x RGSW -> RGSW.  The reference
 * transforms ct0's rows once and takes, for every one of ct1's 2d RLWE rows, the evaluation-domain dot product with that row's
 * digits -- i.e. the external product of ct0 with each row of ct1.  ct0 is a prepared key (its rows already sit in the
 * evaluation domain, exactly what lines 136-137 compute); ct1_a / ct1_b: [count][2d][n] (coefficient domain, the layout
 * fhe_rgsw_prepare takes), replaced in place by the rows of the product, for `count` right-hand RGSW ciphertexts. */
int fhe_rgsw_internal_product(const fhe_ctx *ctx, const fhe_key *rgsw, size_t index, uint64_t *ct1_a, uint64_t *ct1_b,
                              size_t count, fhe_mem mem, void *stream);
/* scheme/fhew/src/rlwe.rs:177-186 `Rlwe::key_switch(param, ksk[index], ct)`. */
int fhe_rlwe_key_switch(const fhe_ctx *ctx, const fhe_key *ksk, size_t index, uint64_t *ct_a, uint64_t *ct_b,
                        size_t batch, fhe_mem mem, void *stream);
/* scheme/fhew/src/rlwe.rs:188-191 `Rlwe::automorphism(param, ak[index] (exponent t), ct)`. */
int fhe_rlwe_automorphism(const fhe_ctx *ctx, const fhe_key *ak, size_t index, int64_t t, uint64_t *ct_a,
                          uint64_t *ct_b, size_t batch, fhe_mem mem, void *stream);

/* ---- LMKCDEY blind rotation --------------------------------------------------------------------------- */
typedef struct fhe_bootstrap_key fhe_bootstrap_key;
/* scheme/fhew/src/bootstrapping.rs:93-113 `BootstrappingKey{brk, ak}` (the LWE ksk belongs to the "next" row):
 * brk = n_lwe prepared RGSW ciphertexts, ak = w+1 prepared automorphism keys with exponents ak_t[0..w]
 * (= [-g, g, g^2, .., g^w] mod 2n, bootstrapping.rs:86-89).  The handles must outlive the bootstrap key. */
int fhe_bootstrap_key_create(const fhe_ctx *ctx, const fhe_key *brk, const fhe_key *ak, const int64_t *ak_t, int w,
                             fhe_bootstrap_key **out);
void fhe_bootstrap_key_destroy(fhe_bootstrap_key *bk);
/* Waits for everything enqueued on `stream`, then FHE_ERR_INVALID if an asynchronous (device-memory) fhe_blind_rotate /
 * fhe_fhew_bootstrap on this key met an LWE coefficient that is not an odd residue mod 2n since the word was last cleared, else
 * FHE_OK; clear != 0 resets the word. */
int fhe_bootstrap_key_status(const fhe_bootstrap_key *bk, void *stream, int clear);
/* scheme/fhew/src/bootstrapping.rs:158-209 `blind_rotate(param, brk, ak, f, LweCiphertext(a, b))` for a batch:
 * lwe_a [batch][n_lwe] and lwe_b [batch] are taken mod 2n (after mod_switch_odd); f = LUT polynomial(s),
 * f_stride = 0 (one f) or n (one per ciphertext); out_a/out_b [batch][n] = the rotated accumulator.
 * ops_out/nops_out (host pointers, may be NULL): the walk of blind_rotate_core per ciphertext,
 * [batch][n_lwe + n + 2] entries, bit 31 set = automorphism ak[idx], clear = external product brk[idx].
 * lwe_b is reduced mod 2n; every lwe_a entry must already be an ODD residue below 2n (what `mod_switch_odd`, lwe.rs:94-99,
 * produces): anything else makes the reference index out of its log map and panic (bootstrapping.rs:221), here FHE_ERR_INVALID.
 * That check is data dependent.  Host-memory calls (and calls that ask for the walk) report it in their return value; DEVICE-memory
 * calls stay asynchronous like every other entry point -- they return once the work is enqueued and record the condition in a
 * sticky status word of the bootstrap key, which fhe_bootstrap_key_status reads (the same holds for fhe_fhew_bootstrap, which
 * ends with this call). */
int fhe_blind_rotate(const fhe_bootstrap_key *bk, const uint64_t *lwe_a, const uint64_t *lwe_b, const uint64_t *f,
                     size_t f_stride, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream,
                     uint32_t *ops_out, uint32_t *nops_out);

/* ---- the LWE side of the FHEW gate ------------------------------------------- */
/* util/src/zq.rs:128-140 via scheme/fhew/src/lwe.rs:90-99: `mod_switch(q_prime)` (odd = 0) / `mod_switch_odd` (odd != 0)
 * for `count` values over q: the reference's f64 arithmetic and rounding reproduced with IEEE double operations. */
int fhe_lwe_mod_switch(uint64_t q, uint64_t q_prime, const uint64_t *in, uint64_t *out, size_t count, int odd, fhe_mem mem,
                       void *stream);
/* scheme/fhew/src/lwe.rs:151-160 `Lwe::key_switch` over q < 2^32: ksk_a [d*n_in][n_out], ksk_b [d*n_in], rows digit-major. */
int fhe_lwe_key_switch(uint64_t q, int log_b, int d, const uint64_t *ksk_a, const uint64_t *ksk_b, const uint64_t *ct_a,
                       const uint64_t *ct_b, size_t n_in, size_t n_out, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem,
                       void *stream);
/* scheme/fhew/src/lwe.rs:22-75 (Add / Sub / Neg / double on LweCiphertext): out = sum_{t<k} coef[t] * in[t] + addend over
 * q < 2^62, k <= 4: the linear part of the gates (scheme/fhew/src/fhew.rs:27-29 `not`, 61-69 Table 1). */
int fhe_lwe_lincomb(uint64_t q, int k, const int64_t *coef, const uint64_t *const *in, uint64_t addend, uint64_t *out, size_t count,
                    fhe_mem mem, void *stream);
/* scheme/fhew/src/rlwe.rs:193-202 `Rlwe::sample_extract(ct, index)`; `addend` (< q) is added to b. */
int fhe_rlwe_sample_extract(uint64_t q, const uint64_t *ct_a, const uint64_t *ct_b, size_t n, size_t index, uint64_t addend,
                            uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream);
/* scheme/fhew/src/bootstrapping.rs:149-155 `Bootstrapping::bootstrap(bk, f, ct)` for a batch of LWE ciphertexts under the
 * ring key (ct_a [batch][N], ct_b [batch] over Q): mod_switch(q_ks) -> Lwe::key_switch -> mod_switch_odd(2N) ->
 * blind_rotate -> sample_extract(0) (+ addend on b: Fhew::op's Q/8, scheme/fhew/src/fhew.rs:39).  One gate bootstrap. */
int fhe_fhew_bootstrap(const fhe_bootstrap_key *bk, uint64_t q_ks, int ks_log_b, int ks_d, const uint64_t *lwe_ksk_a,
                       const uint64_t *lwe_ksk_b, const uint64_t *f, size_t f_stride, uint64_t addend, const uint64_t *ct_a,
                       const uint64_t *ct_b, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream);

/* ---- RNS rings (CKKS) ------------------------------------------------------------------------------------ */
typedef struct fhe_rns_ctx fhe_rns_ctx;   /* bases qs (L primes) and ps (K primes): util/src/ring/rns.rs:278-322 `Rns` */
typedef struct fhe_ckks_key fhe_ckks_key; /* a key-switching key over qs ++ ps, evaluation domain, device resident */
/* `Rns::new(qs).with_ps(ps)` both ways (Q->P for extend_bases, P->Q for rescale_k) plus one transform context per
 * prime.  All L+K primes must be distinct (rns.rs:25, 84 asserts).  L, K <= 32. */
int fhe_rns_ctx_create(const uint64_t *qs, int L, const uint64_t *ps, int K, int device, fhe_rns_ctx **out);
void fhe_rns_ctx_destroy(fhe_rns_ctx *rns);
/* Evaluation-domain residency for `RnsRq`: util/src/ring/rns.rs:40-49 `to_evaluation` /
 * `to_coefficient` limb by limb in ONE launch, and rns.rs:148-158 the evaluation-basis product.  `extended` = 0:
 * polynomials over qs, [batch][L][n]; != 0: over qs ++ ps, [batch][L+K][n].  A caller keeps operands in the evaluation
 * basis on the device across mul -> relinearize -> rescale instead of transforming per operation. */
int fhe_rns_ntt_fwd(const fhe_rns_ctx *rns, int extended, uint64_t *a, size_t n, size_t batch, fhe_mem mem, void *stream);
int fhe_rns_ntt_inv(const fhe_rns_ctx *rns, int extended, uint64_t *a, size_t n, size_t batch, fhe_mem mem, void *stream);
int fhe_rns_pointwise_mul(const fhe_rns_ctx *rns, int extended, uint64_t *a, const uint64_t *b, size_t n, size_t batch, fhe_mem mem,
                          void *stream);
/* util/src/ring/rns.rs:83-91 `RnsRq::extend_bases(ps)`: in [batch][L][n] over qs -> out [batch][K][n], the new
 * p-limbs (fast base conversion with the reference's f64 rounding correction, rns.rs:331-345). */
int fhe_rns_extend_bases(const fhe_rns_ctx *rns, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem,
                         void *stream);
/* util/src/ring/rns.rs:93-97 `RnsRq::switch_bases`: the same polynomial over the OTHER base (extend_bases, old limbs dropped).
 * to_qs = 0: in [batch][L][n] over qs -> out [batch][K][n] over ps; to_qs != 0: in [batch][K][n] over ps -> out [batch][L][n] over qs. */
int fhe_rns_switch_bases(const fhe_rns_ctx *rns, int to_qs, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem,
                         void *stream);
/* util/src/ring/rns.rs:103-118 `rescale_k(K)` of a polynomial over qs ++ ps: in [batch][L+K][n] -> out [batch][L][n]. */
int fhe_rns_rescale_k(const fhe_rns_ctx *rns, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem,
                      void *stream);
/* key-switching key (scheme/ckks/src/ckks.rs:86-88, 143-161): ksk_b, ksk_a [L+K][n], coefficient domain. */
int fhe_ckks_ksk_prepare(const fhe_rns_ctx *rns, const uint64_t *ksk_b, const uint64_t *ksk_a, size_t n, fhe_mem mem,
                         fhe_ckks_key **out);
void fhe_ckks_key_destroy(fhe_ckks_key *key);
/* scheme/ckks/src/ckks.rs:284-293 `Ckks::key_switch(param, ksk, CkksCiphertext(ct_b, ct_a))`, in place, for `batch`
 * ciphertexts: ct_b, ct_a [batch][L][n] over qs, coefficient domain.  n up to 2^17. */
int fhe_ckks_key_switch(const fhe_rns_ctx *rns, const fhe_ckks_key *key, uint64_t *ct_b, uint64_t *ct_a, size_t batch,
                        fhe_mem mem, void *stream);
/* ---- the key switch with its RNS limbs sharded over devices -------------------------------
 * `Ckks::key_switch` (scheme/ckks/src/ckks.rs:284-293) is limb-wise except for its two base conversions.  Every device holds the
 * context and the prepared key (replicated: 8 MiB at cfg4) and OWNS the q-limbs [q_lo, q_hi) and the p-limbs [p_lo, p_hi); all devices
 * own equally many p-limbs (np divides K).  Per batch of ciphertexts:
 *   1. fhe_ckks_shard_products  local: ct.a (all L q-limbs, replicated at staging) -> the two key products on the owned limbs;
 *   2. the caller ALL-GATHERS prod_p over the devices -- the one exchange of the path: an RCCL all-gather in the caller's own
 *      communicator, or hipMemcpyPeerAsync into each peer's gather buffer (examples/multi_gpu_ckks_demo.c); contribution of the
 *      device that owns p-limbs [p_lo, p_hi) goes to slot p_lo / np;
 *   3. fhe_ckks_shard_finish    local: rescale_k on the owned q-limbs against all K gathered p-limbs (+ ct.b's owned limbs).
 * The library links neither RCCL nor peer copies: step 2 sits between two calls, where the caller's communicator already lives.
 * Bases of pseudo-Mersenne primes of one bit length only (every CkksParam of the reference): FHE_ERR_UNSUPPORTED otherwise. */
typedef struct fhe_ckks_shard fhe_ckks_shard;
int fhe_ckks_shard_create(const fhe_rns_ctx *rns, const fhe_ckks_key *key, int q_lo, int q_hi, int p_lo, int p_hi, fhe_ckks_shard **out);
void fhe_ckks_shard_destroy(fhe_ckks_shard *shard);
/* ct_a [batch][L][n] -> prod_q [2][batch][nq][n], prod_p [2][batch][np][n] (part 0: ksk.b * a~, part 1: ksk.a * a~; an internal
 * coefficient-domain form that only fhe_ckks_shard_finish reads) */
int fhe_ckks_shard_products(const fhe_ckks_shard *shard, const uint64_t *ct_a, uint64_t *prod_q, uint64_t *prod_p, size_t batch, fhe_mem mem,
                            void *stream);
/* prod_q (stage 1), gathered_p [K / np][2][batch][np][n], ct_b [batch][nq][n] (ct.b's owned limbs, or NULL) -> out_b, out_a
 * [batch][nq][n]: the owned limbs of the switched ciphertext, bit-identical to fhe_ckks_key_switch's */
int fhe_ckks_shard_finish(const fhe_ckks_shard *shard, const uint64_t *prod_q, const uint64_t *gathered_p, const uint64_t *ct_b,
                          uint64_t *out_b, uint64_t *out_a, size_t batch, fhe_mem mem, void *stream);
/* util/src/ring/rns.rs:99-101 `RnsRq::rescale()` = rescale_k(1) over qs alone: drops the last q-limb (the K == 1 branch of
 * rns.rs:104-111, NOT centred).  in [batch][L][n] -> out [batch][L-1][n]; L >= 2. */
int fhe_rns_rescale(const fhe_rns_ctx *rns, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem, void *stream);
/* util/src/ring/rns.rs:254-270 `RnsRq += RnsRq`, `-= RnsRq`, unary `-` in either basis (what `CkksCiphertext` + / - do on both
 * components: ckks.rs `add_sub`): a <- a + b, a - b, -a over [batch][limbs][n]; extended = 0: qs, 1: qs ++ ps. */
int fhe_rns_add(const fhe_rns_ctx *rns, int extended, uint64_t *a, const uint64_t *b, size_t n, size_t batch, fhe_mem mem, void *stream);
int fhe_rns_sub(const fhe_rns_ctx *rns, int extended, uint64_t *a, const uint64_t *b, size_t n, size_t batch, fhe_mem mem, void *stream);
int fhe_rns_neg(const fhe_rns_ctx *rns, int extended, uint64_t *a, size_t n, size_t batch, fhe_mem mem, void *stream);
/* scheme/ckks/src/ckks.rs:127-129 `automorphism(t)` of an RnsRq (util/src/avec.rs:34-50 on every limb): in, out [batch][L][n],
 * in != out.  t is taken mod 2n; CKKS only uses odd t (5^j, -1) and an even t returns FHE_ERR_UNSUPPORTED. */
int fhe_rns_automorphism(const fhe_rns_ctx *rns, int64_t t, const uint64_t *in, uint64_t *out, size_t n, size_t batch, fhe_mem mem,
                         void *stream);
/* scheme/ckks/src/ckks.rs:274-282 `Ckks::rotate` (key = rtk for j, t = 5^j mod 2n = `param.pow5(j)`) and `Ckks::conjugate`
 * (key = cjk, t = -1): automorphism of both halves, then the key switch.  ct_b, ct_a [batch][L][n], in place. */
int fhe_ckks_rotate(const fhe_rns_ctx *rns, const fhe_ckks_key *key, int64_t t, uint64_t *ct_b, uint64_t *ct_a, size_t batch, fhe_mem mem,
                    void *stream);
/* scheme/ckks/src/ckks.rs:250-263 `Ckks::mul(param, rlk, ct0, ct1)`: tensor, relinearisation (ckks.rs:265-272) and the closing
 * `rescale()`.  ct0_*, ct1_* [batch][L][n] over qs, coefficient domain -> out_b, out_a [batch][L-1][n] over qs[0..L-1).
 * The inputs are transformed once and stay in the evaluation domain through the tensor (7 L transforms + the key switch where
 * the reference's four `Rq * Rq` take 12 L).  A ciphertext on fewer limbs multiplies through an fhe_rns_ctx over that prefix of
 * qs (rns.rs:148-158 multiplies on the intersection of the two limb sets). */
int fhe_ckks_mul(const fhe_rns_ctx *rns, const fhe_ckks_key *rlk, const uint64_t *ct0_b, const uint64_t *ct0_a, const uint64_t *ct1_b,
                 const uint64_t *ct1_a, uint64_t *out_b, uint64_t *out_a, size_t batch, fhe_mem mem, void *stream);


/* ---- TFHE torus path (row T), k = 1: fused kernels (any rank k: the fhe_*k_* entries further down) ---------- */
/* Torus data are uint64_t values of Z/2^64 (util/src/torus.rs `T64`).  Where the reference multiplies torus polynomials
 * with an f64 FFT (util/src/ring/fft/c64.rs:11-56, error <= 2^(64+log_b+log_n-53), c64.rs:186-208), this backend returns
 * the EXACT product (both operands read as signed 64-bit integers, as c64.rs:23-27 does): never less exact than the
 * reference, identical at decode level. */
typedef struct fhe_torus_ctx fhe_torus_ctx;   /* two 60-bit NTT primes + CRT constants */
typedef struct fhe_tggsw_key fhe_tggsw_key;   /* prepared TGGSW ciphertexts (bootstrapping key) */
int fhe_torus_ctx_create(int device, fhe_torus_ctx **out);
void fhe_torus_ctx_destroy(fhe_torus_ctx *t);
/* util/src/misc/decompose.rs:66-81, 114-135 `Base2Decomposor::<T64>::new(log_b, d).decompose(poly)`: [polys][n] ->
 * [polys][d][n] signed digits as two's-complement uint64_t, least significant first.  Bit-exact. */
int fhe_torus_decompose(int log_b, int d, const uint64_t *in, size_t n, size_t polys, uint64_t *out, fhe_mem mem, void *stream);
/* util/src/ring.rs:315-320 `Rt *= &Rt`: a <- a * b in Z_{2^64}[X]/(X^n+1), exact; |b_i| < 2^log_bound_b with
 * n * 2^(64 + log_bound_b) < 2^118 (gadget digits, small secrets: what the reference's FFT product is used for). */
int fhe_torus_mul(const fhe_torus_ctx *t, uint64_t *a, const uint64_t *b, int log_bound_b, size_t n, size_t batch, fhe_mem mem,
                  void *stream);
/* `count` TGGSW ciphertexts (scheme/tfhe/src/tggsw.rs:44-88): rows_a / rows_b [count][2d][n].  n = 256 .. 2048.  The exact products run on
 * one of three forms chosen from the gadget: key words cut into three signed pieces through f64 transforms whose rounded results are exact
 * (2d n 2^log_b <= 2^21, base <= 2^7: BASELINE config 5), three 30-bit NTT primes (2d n 2^(62 + log_b) < 2^88), two 60-bit primes
 * (< 2^118; beyond that FHE_ERR_UNSUPPORTED).  All three give the same bits. */
int fhe_tggsw_prepare(const fhe_torus_ctx *t, int log_b, int d, const uint64_t *rows_a, const uint64_t *rows_b, size_t n, size_t count,
                      fhe_mem mem, fhe_tggsw_key **out);
void fhe_tggsw_key_destroy(fhe_tggsw_key *key);
/* The same key prepared for the f64 FFT product the reference itself computes with (util/src/ring/fft/c64.rs:11-56: fold N reals into
 * N/2 complex values, transform, multiply, transform back, `f64_mod_u64`).  Every entry point below that takes the key (external
 * product, cmux, blind rotation, bootstrap) then runs that mode: one pass of half-size complex transforms where the exact path runs
 * three passes of 30-bit ones.  NOT exact and not bit-reproducible against anything (the reference's own low bits depend on its
 * libm): per product |result - exact| <= 2^(64 + log_b + log2 n - 53), the reference's own bound (c64.rs:186-208); decode-level
 * results agree.  The exact mode (fhe_tggsw_prepare) stays the default and the parity checker. */
int fhe_tggsw_prepare_fft64(const fhe_torus_ctx *t, int log_b, int d, const uint64_t *rows_a, const uint64_t *rows_b, size_t n, size_t count,
                            fhe_mem mem, fhe_tggsw_key **out);
/* scheme/tfhe/src/tggsw.rs:100-112 `Tggsw::external_product(param, key[index], ct)`, in place on [batch][n] a / b. */
int fhe_tggsw_external_product(const fhe_torus_ctx *t, const fhe_tggsw_key *key, size_t index, uint64_t *ct_a, uint64_t *ct_b,
                               size_t batch, fhe_mem mem, void *stream);
/* scheme/tfhe/src/bootstrapping.rs:99-104 `mod_switch`: rounding_shr(v, 64 - log2(2 big_n)) for `count` torus values. */
int fhe_tfhe_mod_switch(const uint64_t *in, uint64_t *out, size_t count, size_t big_n, fhe_mem mem, void *stream);
/* scheme/tfhe/src/tggsw.rs:114-121 `Tggsw::cmux(key[index], ct0, ct1)` = ct0 + external_product(key[index], ct1 - ct0) for `batch`
 * pairs of TGLWE ciphertexts, [batch][n] each; out may alias ct0 or ct1. */
int fhe_tggsw_cmux(const fhe_torus_ctx *t, const fhe_tggsw_key *key, size_t index, const uint64_t *ct0_a, const uint64_t *ct0_b,
                   const uint64_t *ct1_a, const uint64_t *ct1_b, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream);
/* scheme/tfhe/src/tglwe.rs:61-66 `TglweCiphertext::rotate(i)`: both halves times X^i (util/src/ring.rs:299-313 on T64), any
 * integer i (taken mod 2n); ct, out [batch][n], out != ct. */
int fhe_tglwe_rotate(const uint64_t *ct_a, const uint64_t *ct_b, size_t n, int64_t i, uint64_t *out_a, uint64_t *out_b, size_t batch,
                     fhe_mem mem, void *stream);
/* scheme/tfhe/src/bootstrapping.rs:84-96 `blind_rotate`: n_lwe CMUXes (tggsw.rs:114-121) per ciphertext.  a_tilde
 * [batch][n_lwe], b_tilde [batch]: mod-switched TLWE ciphertexts; v [n]: the encoded test polynomial; out [batch][n]. */
int fhe_tfhe_blind_rotate(const fhe_torus_ctx *t, const fhe_tggsw_key *brk, const uint64_t *a_tilde, const uint64_t *b_tilde,
                          const uint64_t *v, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream);
/* scheme/tfhe/src/tglwe.rs:115-127 `Tglwe::sample_extract(ct, index)`. */
int fhe_tglwe_sample_extract(const uint64_t *ct_a, const uint64_t *ct_b, size_t n, size_t index, uint64_t *out_a, uint64_t *out_b,
                             size_t batch, fhe_mem mem, void *stream);
/* scheme/tfhe/src/tlwe.rs:144-153 `Tlwe::key_switch`: ksk_a [n_in*d][n_out], ksk_b [n_in*d], rows digit-major. */
int fhe_tlwe_key_switch(int log_b, int d, const uint64_t *ksk_a, const uint64_t *ksk_b, const uint64_t *ct_a, const uint64_t *ct_b,
                        size_t n_in, size_t n_out, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem, void *stream);
/* scheme/tfhe/src/bootstrapping.rs:78-82 `Bootstrapping::bootstrap(bsk, v, ct)` for `batch` TLWE ciphertexts in one call:
 * mod switch (99-104) -> blind rotation (84-96) -> sample_extract(0) -> key switch, on `stream`.  lwe_a [batch][n_lwe], lwe_b
 * [batch] (n_lwe = number of TGGSW ciphertexts in brk), v [n] the encoded test polynomial, ksk as in fhe_tlwe_key_switch with
 * n_in = n, n_out = n_lwe; out_a [batch][n_lwe], out_b [batch].  Bit-identical to the four calls above chained. */
int fhe_tfhe_bootstrap(const fhe_torus_ctx *t, const fhe_tggsw_key *brk, int ks_log_b, int ks_d, const uint64_t *ksk_a,
                       const uint64_t *ksk_b, const uint64_t *v, const uint64_t *lwe_a, const uint64_t *lwe_b, uint64_t *out_a,
                       uint64_t *out_b, size_t batch, fhe_mem mem, void *stream);

/* ---- TFHE torus path at any TGLWE rank k (the reference's `TglweParam::n`: tglwe.rs:11-35; its TGLWE / TGGSW tests run at k = 2,
 * N = 256, base 2^8, d = 8: tglwe.rs:138-166, tggsw.rs:134-181) ------------------------------------------------------------- */
/* A rank-k TGLWE ciphertext is ONE buffer [k + 1][n]: a_0 .. a_{k-1}, b (tglwe.rs:49-50 `TglweCiphertext(AVec<Rt>, Rt)`); a TGGSW
 * ciphertext is (k + 1) d of them in the order tggsw.rs:80-87 builds (message on a_0: d rows, .., on a_{k-1}, on b).  Products are
 * the exact ones of the k = 1 entries; k = 1 through these entries gives bit-identical results.  1 <= k <= 8, n = 2 .. 2^15. */
typedef struct fhe_tggswk_key fhe_tggswk_key;
/* rows [count][(k + 1) d][k + 1][n] -> prepared (evaluation-domain) key; FHE_ERR_UNSUPPORTED if (k + 1) d n 2^(62 + log_b) >= 2^118 */
int fhe_tggswk_prepare(const fhe_torus_ctx *t, int k, int log_b, int d, const uint64_t *rows, size_t n, size_t count, fhe_mem mem,
                       fhe_tggswk_key **out);
void fhe_tggswk_key_destroy(fhe_tggswk_key *key);
/* tggsw.rs:100-112 `Tggsw::external_product(param, key[index], ct)`, in place on ct [batch][k + 1][n] */
int fhe_tggswk_external_product(const fhe_torus_ctx *t, const fhe_tggswk_key *key, size_t index, uint64_t *ct, size_t batch, fhe_mem mem,
                                void *stream);
/* tggsw.rs:114-121 `Tggsw::cmux(key[index], ct0, ct1)`; [batch][k + 1][n] each, out may alias ct0 or ct1 */
int fhe_tggswk_cmux(const fhe_torus_ctx *t, const fhe_tggswk_key *key, size_t index, const uint64_t *ct0, const uint64_t *ct1, uint64_t *out,
                    size_t batch, fhe_mem mem, void *stream);
/* tglwe.rs:61-66 `TglweCiphertext::rotate(i)`; ct, out [batch][k + 1][n], out != ct */
int fhe_tglwek_rotate(const uint64_t *ct, int k, size_t n, int64_t i, uint64_t *out, size_t batch, fhe_mem mem, void *stream);
/* tglwe.rs:115-127 `Tglwe::sample_extract(ct, index)`: a TLWE ciphertext of dimension k n: out_a [batch][k n], out_b [batch] */
int fhe_tglwek_sample_extract(const uint64_t *ct, int k, size_t n, size_t index, uint64_t *out_a, uint64_t *out_b, size_t batch, fhe_mem mem,
                              void *stream);
/* bootstrapping.rs:84-96 `blind_rotate`: a_tilde [batch][n_lwe], b_tilde [batch] mod-switched, v [n] encoded -> out [batch][k + 1][n] */
int fhe_tfhek_blind_rotate(const fhe_torus_ctx *t, const fhe_tggswk_key *brk, const uint64_t *a_tilde, const uint64_t *b_tilde, const uint64_t *v,
                           uint64_t *out, size_t batch, fhe_mem mem, void *stream);
/* bootstrapping.rs:78-82 `Bootstrapping::bootstrap`: mod switch -> blind rotation -> sample_extract(0) -> key switch with
 * n_in = k n, n_out = n_lwe (ksk_a [k n ks_d][n_lwe], ksk_b [k n ks_d]); lwe_a, out_a [batch][n_lwe], lwe_b, out_b [batch] */
int fhe_tfhek_bootstrap(const fhe_torus_ctx *t, const fhe_tggswk_key *brk, int ks_log_b, int ks_d, const uint64_t *ksk_a, const uint64_t *ksk_b,
                        const uint64_t *v, const uint64_t *lwe_a, const uint64_t *lwe_b, uint64_t *out_a, uint64_t *out_b, size_t batch,
                        fhe_mem mem, void *stream);

/* ---- key material on the device ------------------------------------------------- */
/* Randomness.  Every producer below draws from an `fhe_rng`: a 256-bit ChaCha20 key.  Value i of a draw is a word of ChaCha20
 * block (i / words per block) -- counter based: every GPU lane finds its value without shared state, a draw is reproducible and
 * order independent -- under a PER-CALL key derived from (generator key, stream_id, the entry point's purpose tag), so different
 * entry points never share keystream even when a caller reuses a stream id: a secret key drawn with fhe_sample_binary(rng, 7) has
 * nothing in common with the public mask of fhe_tlwe_sk_encrypt(rng, 7).  What the caller MUST still guarantee: two calls of the
 * SAME entry point on the same generator use different stream ids (equal ids reproduce the same mask and the same noise:
 * b1 - b2 = pt1 - pt2).  Pass FHE_STREAM_AUTO to let the generator number the calls itself (fresh id per call, from its own
 * counter, in the upper half of the id space: keep explicit ids below 2^63 when mixing both).  The reference draws from
 * `thread_rng()`; draws are not parity relevant, these entry points are validated at decrypt level and statistically.
 *   fhe_rng_create(key32, &rng): key32 = 32 bytes of caller entropy, or NULL = 32 bytes from the operating system (getrandom),
 *                                what the reference's thread_rng() is seeded with.
 *   fhe_rng_create_from_seed:    TESTS AND REPRODUCIBLE RUNS ONLY -- 64 bits of entropy stretched to a key (SplitMix64); keys made
 *                                from it are as strong as the seed, never stronger.
 *   fhe_chacha20_block:          the block function alone (host), for known-answer tests: ChaCha20 in the original layout, 64-bit
 *                                block counter and 64-bit nonce (RFC 8439's function with its 32/96-bit split undone). */
typedef struct fhe_rng fhe_rng;
#define FHE_STREAM_AUTO UINT64_MAX
int fhe_rng_create(const uint8_t *key32, fhe_rng **out);
int fhe_rng_create_from_seed(uint64_t seed, fhe_rng **out);
void fhe_rng_destroy(fhe_rng *rng);
int fhe_chacha20_block(const uint8_t *key32, uint64_t nonce, uint64_t counter, uint8_t *out64);
/* util/src/zq.rs:91-93 `Zq::sample_uniform`: `count` values uniform in [0, q) */
int fhe_sample_uniform(uint64_t q, const fhe_rng *rng, uint64_t stream_id, uint64_t *out, size_t count, fhe_mem mem, void *stream);
/* util/src/torus.rs `T64::sample_uniform`: uniform 64-bit torus values */
int fhe_sample_torus(const fhe_rng *rng, uint64_t stream_id, uint64_t *out, size_t count, fhe_mem mem, void *stream);
/* util/src/misc/distribution.rs:23-46 `dg(std_dev, n)` (weights from the reference's erf approximation, support
 * [-floor(n std_dev), floor(n std_dev)]) as `Zq::sample_i64` (zq.rs:95-97); q = 0: plain two's-complement integers */
int fhe_sample_dg(uint64_t q, double std_dev, int n_sigma, const fhe_rng *rng, uint64_t stream_id, uint64_t *out, size_t count, fhe_mem mem,
                  void *stream);
/* util/src/misc/decompose.rs:35-40 `power_up`: out[p][j] = in[p] * 2^(rounding_bits + j log_b) mod q, [polys][d][n] */
int fhe_power_up(uint64_t q, int log_b, int d, const uint64_t *in, size_t n, size_t polys, uint64_t *out, fhe_mem mem, void *stream);
/* scheme/fhew/src/rlwe.rs:146-156 `Rlwe::sk_encrypt` for `batch` plaintexts (pt [batch][n] or NULL = zeros): a uniform,
 * e <- dg(3.2, 6), b = a sk + e + pt.  sk [n]: the secret key as Zq values (`Zq::from_i64` of its coefficients). */
int fhe_rlwe_sk_encrypt(const fhe_ctx *ctx, const uint64_t *sk, const uint64_t *pt, size_t n, size_t batch, const fhe_rng *rng, uint64_t stream_id, uint64_t *ct_a, uint64_t *ct_b, fhe_mem mem, void *stream);
/* rlwe.rs:237-249 `Rlwe::share_encrypt(param, a, sk, pt)`: b = a sk + e + pt for a GIVEN mask a -- what the multi-party protocol of
 * rlwe.rs:205-324 is made of: `pk_share_gen` (pt NULL, a = the common reference string), `ksk_share_gen` / `ak_share_gen` (one row per
 * call of the gadget), `share_decrypt` (pt NULL, a = the ciphertext's mask); the merges are fhe_rq_sum / fhe_rq_sub.  a [a_rows][n] with
 * a_rows = rows or 1 (one polynomial shared by every row); sk [n] as Zq values; pt [rows][n] or NULL; out_b [rows][n]. */
int fhe_rlwe_share_encrypt(const fhe_ctx *ctx, const uint64_t *a, size_t a_rows, const uint64_t *sk, const uint64_t *pt, size_t n, size_t rows,
                           const fhe_rng *rng, uint64_t stream_id, uint64_t *out_b, fhe_mem mem, void *stream);
/* rlwe.rs:158-170 `Rlwe::pk_encrypt`: u <- zo(0.5), e0, e1 <- dg(3.2, 6) per ciphertext; a = pk.a u + e0, b = pk.b u + e1 + pt.  pk_a, pk_b
 * [n] (rlwe.rs:98-101 `pk_gen` = fhe_rlwe_sk_encrypt of zero, or a merged multi-party key); pt [batch][n] or NULL. */
int fhe_rlwe_pk_encrypt(const fhe_ctx *ctx, const uint64_t *pk_a, const uint64_t *pk_b, const uint64_t *pt, size_t n, size_t batch, const fhe_rng *rng,
                        uint64_t stream_id, uint64_t *ct_a, uint64_t *ct_b, fhe_mem mem, void *stream);
/* rlwe.rs:172-175 `Rlwe::decrypt`: pt = b - a sk; [batch][n]; pt may alias ct_b. */
int fhe_rlwe_decrypt(const fhe_ctx *ctx, const uint64_t *sk, const uint64_t *ct_a, const uint64_t *ct_b, size_t n, size_t batch, uint64_t *pt,
                     fhe_mem mem, void *stream);
/* scheme/fhew/src/rgsw.rs:84-105 `Rgsw::sk_encrypt` of `count` plaintext polynomials (pt [count][n], `Rgsw::encode`d):
 * rows_a / rows_b [count][2d][n], the layout fhe_rgsw_prepare takes. */
int fhe_rgsw_encrypt(const fhe_ctx *ctx, int log_b, int d, const uint64_t *sk, const uint64_t *pt, size_t n, size_t count,
                     const fhe_rng *rng, uint64_t stream_id, uint64_t *rows_a, uint64_t *rows_b, fhe_mem mem, void *stream);
/* rgsw.rs:75-83 `Rgsw::pk_encrypt`: as fhe_rgsw_encrypt with the 2d encryptions of zero made by fhe_rlwe_pk_encrypt (what the reference's
 * RGSW tests encrypt with; `Bootstrapping::key_share_gen`, bootstrapping.rs:277-283, under the merged public key). */
int fhe_rgsw_pk_encrypt(const fhe_ctx *ctx, int log_b, int d, const uint64_t *pk_a, const uint64_t *pk_b, const uint64_t *pt, size_t n, size_t count,
                        const fhe_rng *rng, uint64_t stream_id, uint64_t *rows_a, uint64_t *rows_b, fhe_mem mem, void *stream);
/* scheme/fhew/src/rlwe.rs:109-120 `Rlwe::ksk_gen(param, sk0, sk1)` (t = 0) and 122-132 `Rlwe::ak_gen(param, t, sk0)` (t != 0,
 * sk1 ignored: the key switches sk0(X^t) back to sk0): rows_a / rows_b [d][n], the layout fhe_ksk_prepare takes. */
int fhe_rlwe_ksk_gen(const fhe_ctx *ctx, int log_b, int d, const uint64_t *sk0, const uint64_t *sk1, int64_t t, size_t n,
                     const fhe_rng *rng, uint64_t stream_id, uint64_t *rows_a, uint64_t *rows_b, fhe_mem mem, void *stream);
/* scheme/fhew/src/lwe.rs:128-139 `Lwe::sk_encrypt` for `rows` plaintexts over any modulus q < 2^62 (pt [rows] or NULL = zeros):
 * out_a [rows][n] uniform, out_b[r] = <a[r], sk> + pt[r] + e[r], e <- dg(3.2, 6). */
int fhe_lwe_sk_encrypt(uint64_t q, const uint64_t *sk, const uint64_t *pt, size_t n, size_t rows, const fhe_rng *rng, uint64_t stream_id,
                       uint64_t *out_a, uint64_t *out_b, fhe_mem mem, void *stream);
/* scheme/fhew/src/lwe.rs:108-119 `Lwe::ksk_gen(param, sk0, sk1)`: ksk_a [n1 d][n0], ksk_b [n1 d] (digit-major rows), the layout
 * fhe_lwe_key_switch / fhe_fhew_bootstrap take with n_in = n1, n_out = n0. */
int fhe_lwe_ksk_gen(uint64_t q, int log_b, int d, const uint64_t *sk0, size_t n0, const uint64_t *sk1, size_t n1, const fhe_rng *rng, uint64_t stream_id, uint64_t *ksk_a, uint64_t *ksk_b, fhe_mem mem, void *stream);
/* lwe.rs:169-183 `Lwe::sk_share_encrypt(param, a, sk, pt)` / 197-207 `share_decrypt` (pt NULL): b[r] = <a[r], sk> + pt[r] + e[r] for GIVEN
 * masks a [rows][n] (a common reference string, or a ciphertext's mask); the merges (lwe.rs:185-195, 209-212) are sums (fhe_rq_sum). */
int fhe_lwe_share_encrypt(uint64_t q, const uint64_t *a, const uint64_t *sk, const uint64_t *pt, size_t n, size_t rows, const fhe_rng *rng,
                          uint64_t stream_id, uint64_t *out_b, fhe_mem mem, void *stream);
/* lwe.rs:214-226 `Lwe::ksk_share_gen(param, crs, sk0, sk1)`: crs [n1 d][n0] is the ksk_a every party (and the merged key) uses; out_b [n1 d];
 * the merged key-switching key is (crs, fhe_rq_sum of the shares) (lwe.rs:228-237). */
int fhe_lwe_ksk_share_gen(uint64_t q, int log_b, int d, const uint64_t *crs, const uint64_t *sk0, size_t n0, const uint64_t *sk1, size_t n1,
                          const fhe_rng *rng, uint64_t stream_id, uint64_t *out_b, fhe_mem mem, void *stream);
/* util/src/ring.rs:328-341 `Rq: Sum`: out[i] = sum_k in[k][i] mod q, in [count][len] (callers used to loop fhe_rq_add) */
int fhe_rq_sum(uint64_t q, const uint64_t *in, size_t len, size_t count, uint64_t *out, fhe_mem mem, void *stream);
/* ---- CKKS key material (scheme/ckks/src/ckks.rs:139-183, 215-225).  Secret keys are two's-complement i64 vectors. */
/* util/src/misc/distribution.rs:10-21 `zo(rho)` (ckks.rs:139-141 `Ckks::sk_gen`: rho = 0.5): -1 / +1 / 0 as i64 */
int fhe_sample_zo(double rho, const fhe_rng *rng, uint64_t stream_id, uint64_t *out, size_t count, fhe_mem mem, void *stream);
/* ckks.rs:215-225 `Ckks::sk_encrypt` for `batch` plaintexts over qs (extended = 0) or qs ++ ps: b = -(a sk) + e + pt, a uniform,
 * e <- dg(3.2, 6).  pt [batch][limbs][n] or NULL (zeros: ckks.rs:143-146 `pk_gen`); out_b, out_a [batch][limbs][n]. */
int fhe_ckks_sk_encrypt(const fhe_rns_ctx *rns, int extended, const uint64_t *sk, const uint64_t *pt, size_t n, size_t batch, const fhe_rng *rng, uint64_t stream_id, uint64_t *out_b, uint64_t *out_a, fhe_mem mem, void *stream);
/* ckks.rs:154-161 `Ckks::ksk_gen(param, sk, sk_prime)` -> ksk_b, ksk_a [L+K][n] for fhe_ckks_ksk_prepare.  sk_prime NULL: sk^2
 * (ckks.rs:163-166 `rlk_gen`), squared on the device -- exact while n max|sk_i|^2 < q_0 / 2 (every `zo` key), checked: FHE_ERR_INVALID otherwise; `cjk_gen` / `rtk_gen` (ckks.rs:168-183): sk_prime = sk(X^t), t = -1 / 5^j. */
int fhe_ckks_ksk_gen(const fhe_rns_ctx *rns, const uint64_t *sk, const uint64_t *sk_prime, size_t n, const fhe_rng *rng, uint64_t stream_id,
                     uint64_t *ksk_b, uint64_t *ksk_a, fhe_mem mem, void *stream);

/* ckks.rs:227-238 `Ckks::pk_encrypt` for `batch` plaintexts over qs: u <- zo(0.5), e0, e1 <- dg(3.2, 6) per ciphertext; a = pk.a u + e0,
 * b = pk.b u + e1 + pt.  pk_b, pk_a [L][n]: ckks.rs:143-146 `pk_gen` = fhe_ckks_sk_encrypt with pt NULL.  pt [batch][L][n] or NULL. */
int fhe_ckks_pk_encrypt(const fhe_rns_ctx *rns, const uint64_t *pk_b, const uint64_t *pk_a, const uint64_t *pt, size_t n, size_t batch,
                        const fhe_rng *rng, uint64_t stream_id, uint64_t *out_b, uint64_t *out_a, fhe_mem mem, void *stream);
/* ckks.rs:240-248 `Ckks::decrypt`: pt = b + a sk over qs; sk [n] i64; ct_b, ct_a, pt [batch][L][n]; pt may alias ct_b.  (A ciphertext on
 * fewer limbs decrypts through an fhe_rns_ctx over that prefix of qs.) */
int fhe_ckks_decrypt(const fhe_rns_ctx *rns, const uint64_t *sk, const uint64_t *ct_b, const uint64_t *ct_a, size_t n, size_t batch, uint64_t *pt,
                     fhe_mem mem, void *stream);
/* ckks.rs:250-253 `Ckks::mul_constant` after its `encode` (host-side F256 arithmetic, out of scope): (pt * b, pt * a).rescale().  pt
 * [pt_batch][L][n] encoded plaintexts, pt_batch = 1 (one constant for the batch) or batch; ct [batch][L][n] -> out [batch][L-1][n]. */
int fhe_ckks_mul_plain(const fhe_rns_ctx *rns, const uint64_t *pt, size_t pt_batch, const uint64_t *ct_b, const uint64_t *ct_a, uint64_t *out_b,
                       uint64_t *out_a, size_t n, size_t batch, fhe_mem mem, void *stream);

/* ---- TFHE key material, k = 1 (rank k: fhe_tglwek_sk_encrypt / fhe_tggswk_encrypt at the end).  Draws are counter based (ChaCha20, as above): reproducible per
 * (generator key, stream_id), checked at decode level like the reference's own tests (its draws are unseeded). */
/* util/src/misc/distribution.rs:49-54 `tdg(std_dev)`: torus Gaussian noise (Box-Muller deviate, fractional part scaled by 2^64) */
int fhe_sample_tdg(double std_dev, const fhe_rng *rng, uint64_t stream_id, uint64_t *out, size_t count, fhe_mem mem, void *stream);
/* distribution.rs `binary()` (scheme/tfhe/src/tlwe.rs:96-98 `Tlwe::sk_gen`): one uniform bit per output word */
int fhe_sample_binary(const fhe_rng *rng, uint64_t stream_id, uint64_t *out, size_t count, fhe_mem mem, void *stream);
/* scheme/tfhe/src/tlwe.rs:122-132 `Tlwe::sk_encrypt` for `rows` plaintexts (pt [rows] or NULL): out_a [rows][n], out_b [rows] */
int fhe_tlwe_sk_encrypt(const uint64_t *sk, const uint64_t *pt, size_t n, size_t rows, double std_dev, const fhe_rng *rng, uint64_t stream_id,
                        uint64_t *out_a, uint64_t *out_b, fhe_mem mem, void *stream);
/* scheme/tfhe/src/tlwe.rs:100-111 `Tlwe::ksk_gen(param, sk0, sk1)`: ksk_a [n1 d][n0], ksk_b [n1 d] (digit-major), the layout
 * fhe_tlwe_key_switch / fhe_tfhe_bootstrap take with n_in = n1, n_out = n0 */
int fhe_tlwe_ksk_gen(int log_b, int d, const uint64_t *sk0, size_t n0, const uint64_t *sk1, size_t n1, double std_dev, const fhe_rng *rng, uint64_t stream_id, uint64_t *ksk_a, uint64_t *ksk_b, fhe_mem mem, void *stream);
/* scheme/tfhe/src/tglwe.rs:91-103 `Tglwe::sk_encrypt`: ct_a uniform, ct_b = ct_a * sk + e + pt, [rows][n]; sk [n] binary; pt NULL = zeros.
 * The product a * sk is the exact integer product of row T. */
int fhe_tglwe_sk_encrypt(const fhe_torus_ctx *t, const uint64_t *sk, const uint64_t *pt, size_t n, size_t rows, double std_dev, const fhe_rng *rng, uint64_t stream_id, uint64_t *ct_a, uint64_t *ct_b, fhe_mem mem, void *stream);
/* scheme/tfhe/src/tggsw.rs:73-88 `Tggsw::sk_encrypt` for `count` plaintext polynomials pt [count][n] (bootstrapping.rs:64-69: the
 * constants z_i): rows_a, rows_b [count][2d][n], the layout fhe_tggsw_prepare takes */
int fhe_tggsw_encrypt(const fhe_torus_ctx *t, int log_b, int d, const uint64_t *sk, const uint64_t *pt, size_t n, size_t count, double std_dev,
                      const fhe_rng *rng, uint64_t stream_id, uint64_t *rows_a, uint64_t *rows_b, fhe_mem mem, void *stream);

/* tglwe.rs:91-103 `Tglwe::sk_encrypt` at rank k: sk [k n] binary (the TLWE key that `as_rings` cuts into k rings, tglwe.rs:40-44);
 * pt [rows][n] or NULL; ct [rows][k + 1][n] with b = sum_j a_j s_j + e + pt */
int fhe_tglwek_sk_encrypt(const fhe_torus_ctx *t, int k, const uint64_t *sk, const uint64_t *pt, size_t n, size_t rows, double std_dev,
                          const fhe_rng *rng, uint64_t stream_id, uint64_t *ct, fhe_mem mem, void *stream);
/* tggsw.rs:73-88 `Tggsw::sk_encrypt` at rank k of pt [count][n]: rows [count][(k + 1) d][k + 1][n], the layout fhe_tggswk_prepare takes */
int fhe_tggswk_encrypt(const fhe_torus_ctx *t, int k, int log_b, int d, const uint64_t *sk, const uint64_t *pt, size_t n, size_t count,
                       double std_dev, const fhe_rng *rng, uint64_t stream_id, uint64_t *rows, fhe_mem mem, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FHE_RING_H */
