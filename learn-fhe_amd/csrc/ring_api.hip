// extern "C" entry points declared in include/fhe_ring.h: context set-up and the transform launches.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <cstring>
#include <mutex>
#include <new>
#include <set>
#include <utility>

#include "../../include/fhe_ring.h"
#include "api_common.hpp"
#include "ctx.hpp"
#include "ntt_kernels.hpp"
#include "ntt14w.hpp"

using fhe::u64;

namespace {
// validates (ctx, n) the way the reference would panic
int check_transform(const fhe_ctx *ctx, const void *a, size_t n, size_t batch) {
    if (!ctx || !is_pow2(n) || (!a && batch)) return FHE_ERR_INVALID;
    if (n == 1) return FHE_OK;
    int log_n = ilog2(n);
    if (log_n > ctx->s - 1) return FHE_ERR_NO_ROOT;
    if (log_n > ctx->log_cap) return FHE_ERR_UNSUPPORTED;
    return FHE_OK;
}

// hipFuncSetAttribute once per (kernel, device), not per launch
// `done`: one bit per device, owned by the launch-template instantiation of this kernel -- the attribute is set once per (kernel,
// device) and every later launch costs one relaxed load (no lock, no lookup)
hipError_t set_max_lds(const void *kernel, int bytes, std::atomic<uint64_t> &done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = uint64_t(1) << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

template <class A, int LOG_N, int LOG_E, int PPW, bool PFX = false>
int launch_gen(bool inverse, const fhe::ModDesc *descs, unsigned n_desc, u64 *a, size_t subs, int pb, hipStream_t st, fhe::NttIo io) {
    using C = fhe::NttCfg<LOG_N, LOG_E, PPW>;
    // measured on MI355X (tools/ntt_lab.hip): staging through LDS wins for the forward stores, direct 16-byte loads win
    // for the inverse
    auto k = inverse ? fhe::ntt_inv_kernel<A, LOG_N, LOG_E, PPW, PFX, true> : fhe::ntt_fwd_kernel<A, LOG_N, LOG_E, PPW, PFX, false>;
    if (C::LDS_BYTES > 64 * 1024) {
        static std::atomic<uint64_t> done[2];
        HIP_TRY(set_max_lds((const void *)k, (int)C::LDS_BYTES, done[inverse ? 1 : 0]));
    }
    unsigned grid = (unsigned)((subs + PPW - 1) / PPW);
    hipLaunchKernelGGL(k, dim3(grid), dim3(C::THREADS), C::LDS_BYTES, st, a, descs, n_desc, (unsigned)subs, pb, io);
    HIP_TRY(hipGetLastError());
    return FHE_OK;
}

// (LOG_N -> LOG_E, PPW): 16 coefficients per thread from N = 128 up; small rings pack many polynomials
// into one 64..256-thread workgroup
template <class A>
int dispatch_small(bool inv, int log_n, const fhe::ModDesc *d, unsigned nd, u64 *a, size_t subs, hipStream_t st, fhe::NttIo io) {
    switch (log_n) {
        case 1: return launch_gen<A, 1, 1, 64>(inv, d, nd, a, subs, 0, st, io);
        case 2: return launch_gen<A, 2, 2, 64>(inv, d, nd, a, subs, 0, st, io);
        case 3: return launch_gen<A, 3, 3, 64>(inv, d, nd, a, subs, 0, st, io);
        case 4: return launch_gen<A, 4, 4, 64>(inv, d, nd, a, subs, 0, st, io);
        case 5: return launch_gen<A, 5, 3, 16>(inv, d, nd, a, subs, 0, st, io);
        case 6: return launch_gen<A, 6, 3, 16>(inv, d, nd, a, subs, 0, st, io);
        case 7: return launch_gen<A, 7, 4, 16>(inv, d, nd, a, subs, 0, st, io);
        case 8: return launch_gen<A, 8, 4, 16>(inv, d, nd, a, subs, 0, st, io);
        case 9: return launch_gen<A, 9, 4, 8>(inv, d, nd, a, subs, 0, st, io);
        default: return FHE_ERR_UNSUPPORTED;
    }
}
template <class A>
int dispatch_large(bool inv, int log_n, const fhe::ModDesc *d, unsigned nd, u64 *a, size_t subs, int pb, hipStream_t st, fhe::NttIo io) {
    switch (log_n) {
        case 10: return launch_gen<A, 10, 4, 4>(inv, d, nd, a, subs, 0, st, io);
        case 11: return launch_gen<A, 11, 4, 2>(inv, d, nd, a, subs, 0, st, io);
        case 12: return launch_gen<A, 12, 4, 1>(inv, d, nd, a, subs, 0, st, io);
        case 13: return launch_gen<A, 13, 4, 1>(inv, d, nd, a, subs, 0, st, io);
        default: return FHE_ERR_UNSUPPORTED;
    }
}

// N = 2^14 (and the 2^14 sub-transforms of larger rings): the register-resident, wave-local kernels of ntt14w.hpp, two
// workgroups per CU.  AF / AI: the arithmetic policy of each direction (measured, tools/ntt_lab2.hip, 4096 transforms at 60 bits:
// forward 0.274 ms with the two-operand twiddles (ArithDS) against 0.300 with the 8-byte ones (ArithPM); inverse 0.296 against
// 0.320 since its first pass runs in diagonal form and nothing spills: ntt14w.hpp).
inline bool limb_major_disabled() { return fhe::opt(fhe::OPT_NO_LIMB_MAJOR) != 0; }       // lab switch (api_common.hpp)
inline bool wave_local_small_disabled() { return fhe::opt(fhe::OPT_NO_W12) != 0; }       // 2^12 / 2^13 on the generic kernels
// a launch of the two-workgroups-per-CU kernels spans several generations of workgroups from this size on
inline size_t several_generations() { return size_t(8) * (size_t)fhe::current_cu_count(); }
template <class AF, class AI, int R0 = 3>
int launch14(bool inv, const fhe::ModDesc *d, unsigned nd, u64 *a, size_t subs, int pb, hipStream_t st, fhe::NttIo io) {
    auto k = pb ? (inv ? (io.mul ? fhe::ntt14w_inv_kernel<AI, true, true, R0> : fhe::ntt14w_inv_kernel<AI, true, false, R0>) : fhe::ntt14w_fwd_kernel<AF, true, R0>)
                : (inv ? (io.mul ? fhe::ntt14w_inv_kernel<AI, false, true, R0> : fhe::ntt14w_inv_kernel<AI, false, false, R0>) : fhe::ntt14w_fwd_kernel<AF, false, R0>);
    static std::atomic<uint64_t> done[8];
    HIP_TRY(set_max_lds((const void *)k, (int)fhe::w14::lds_bytes<R0>(), done[(pb ? 4 : 0) | (inv ? 2 : 0) | (io.mul ? 1 : 0)]));
    // several moduli: modulus-major dispatch order (ntt14w.hpp, sub_of_block)
    const size_t polys = subs >> pb;
    // (only launches of several generations of workgroups: when the whole launch is resident at once the order is irrelevant, and
    // measured 3 % slower at cfg4 batch 8)
    const bool by_mod = nd > 1 && nd <= 65535 && polys % nd == 0 && subs >= several_generations() && !limb_major_disabled();
    const dim3 grid = by_mod ? dim3((unsigned)(subs / nd), nd) : dim3((unsigned)subs);
    hipLaunchKernelGGL(k, grid, dim3(fhe::w14::threads<R0>()), fhe::w14::lds_bytes<R0>(), st, a, d, nd, (unsigned)subs, pb, io);
    HIP_TRY(hipGetLastError());
    return FHE_OK;
}

// the fused ring product (ntt14w_mul_kernel): a <- inverse(forward(a) (.) io.mul), 2^13 .. 2^15
template <class A, int R0>
int launch_mul14(const fhe::ModDesc *d, unsigned nd, u64 *a, size_t subs, hipStream_t st, fhe::NttIo io) {
    auto k = fhe::ntt14w_mul_kernel<A, R0>;
    static std::atomic<uint64_t> done;
    HIP_TRY(set_max_lds((const void *)k, (int)fhe::w14::lds_bytes<R0>(), done));
    const bool by_mod = nd > 1 && nd <= 65535 && subs % nd == 0 && subs >= several_generations() && !limb_major_disabled();
    const dim3 grid = by_mod ? dim3((unsigned)(subs / nd), nd) : dim3((unsigned)subs);
    hipLaunchKernelGGL(k, grid, dim3(fhe::w14::threads<R0>()), fhe::w14::lds_bytes<R0>(), st, a, d, nd, (unsigned)subs, 0, io);
    HIP_TRY(hipGetLastError());
    return FHE_OK;
}
template <class A>
int dispatch_mul14(int log_n, const fhe::ModDesc *d, unsigned nd, u64 *a, size_t subs, hipStream_t st, fhe::NttIo io) {
    switch (log_n) {
        case 13: return launch_mul14<A, 2>(d, nd, a, subs, st, io);
        case 14: return launch_mul14<A, 3>(d, nd, a, subs, st, io);
        case 15: return launch_mul14<A, 4>(d, nd, a, subs, st, io);
        default: return FHE_ERR_UNSUPPORTED;
    }
}

// pm = common bit length of pseudo-Mersenne eligible moduli; kernels are instantiated for 60, 55 and 54 bits (anything else: Shoup)
int sub_transform(bool inv, const fhe::ModDesc *d, unsigned nd, u64 *a, int log_n, size_t subs, int pb, int pm, hipStream_t st, fhe::NttIo io) {
    if (pb && log_n != 14) return FHE_ERR_UNSUPPORTED;
    if (log_n == 15) {  // whole 2^15 rings in one pass over HBM (ntt14w.hpp, R0 = 4): one workgroup of 1024 threads per CU
        if (pm == 60) return launch14<fhe::ArithDS<60>, fhe::ArithDS<60>, 4>(inv, d, nd, a, subs, 0, st, io);
        if (pm == 54) return launch14<fhe::ArithDS<54>, fhe::ArithDS<54>, 4>(inv, d, nd, a, subs, 0, st, io);
        if (pm == 55) return launch14<fhe::ArithDS<55>, fhe::ArithDS<55>, 4>(inv, d, nd, a, subs, 0, st, io);
        return launch14<fhe::ArithShoup, fhe::ArithShoup, 4>(inv, d, nd, a, subs, 0, st, io);
    }
    if ((log_n == 12 || log_n == 13) && !wave_local_small_disabled()) {
        // 2^12 / 2^13 rings in the wave-local form as well (R0 = 1 / 2: two / four waves, each owning a 2^11 block; eight / four
        // workgroups per CU): at 60 bits forward 3.3 -> 4.0 and 2.8 -> 4.1 TB/s, inverse 2.9 -> 4.2 and 2.5 -> 3.9
        if (pm == 60) return log_n == 13 ? launch14<fhe::ArithDS<60>, fhe::ArithDS<60>, 2>(inv, d, nd, a, subs, 0, st, io)
                                         : launch14<fhe::ArithDS<60>, fhe::ArithDS<60>, 1>(inv, d, nd, a, subs, 0, st, io);
        if (pm == 54) return log_n == 13 ? launch14<fhe::ArithDS<54>, fhe::ArithDS<54>, 2>(inv, d, nd, a, subs, 0, st, io)
                                         : launch14<fhe::ArithDS<54>, fhe::ArithDS<54>, 1>(inv, d, nd, a, subs, 0, st, io);
        if (pm == 55) return log_n == 13 ? launch14<fhe::ArithDS<55>, fhe::ArithDS<55>, 2>(inv, d, nd, a, subs, 0, st, io)
                                         : launch14<fhe::ArithDS<55>, fhe::ArithDS<55>, 1>(inv, d, nd, a, subs, 0, st, io);
        return log_n == 13 ? launch14<fhe::ArithShoup, fhe::ArithShoup, 2>(inv, d, nd, a, subs, 0, st, io)
                           : launch14<fhe::ArithShoup, fhe::ArithShoup, 1>(inv, d, nd, a, subs, 0, st, io);
    }
    if (log_n < 10) return dispatch_small<fhe::ArithShoup>(inv, log_n, d, nd, a, subs, st, io);
    if (pm == 60) return log_n == 14 ? launch14<fhe::ArithDS<60>, fhe::ArithDS<60>>(inv, d, nd, a, subs, pb, st, io)
                                     : dispatch_large<fhe::ArithPM<60>>(inv, log_n, d, nd, a, subs, pb, st, io);
    if (pm == 54) return log_n == 14 ? launch14<fhe::ArithDS<54>, fhe::ArithDS<54>>(inv, d, nd, a, subs, pb, st, io)
                                     : dispatch_large<fhe::ArithPM<54>>(inv, log_n, d, nd, a, subs, pb, st, io);
    // the reference's own 55-bit parameter sets (scheme/fhew/examples/multi_key_uint8.rs:15-29, util/src/ring/rns.rs:373-386)
    if (pm == 55) return log_n == 14 ? launch14<fhe::ArithDS<55>, fhe::ArithDS<55>>(inv, d, nd, a, subs, pb, st, io)
                                     : dispatch_large<fhe::ArithPM<55>>(inv, log_n, d, nd, a, subs, pb, st, io);
    return log_n == 14 ? launch14<fhe::ArithShoup, fhe::ArithShoup>(inv, d, nd, a, subs, pb, st, io)
                       : dispatch_large<fhe::ArithShoup>(inv, log_n, d, nd, a, subs, pb, st, io);
}
int sub_fwd(const fhe::ModDesc *d, unsigned nd, u64 *a, int log_n, size_t subs, int pb, int pm, hipStream_t st, fhe::NttIo io) {
    return sub_transform(false, d, nd, a, log_n, subs, pb, pm, st, io);
}
int sub_inv(const fhe::ModDesc *d, unsigned nd, u64 *a, int log_n, size_t subs, int pb, int pm, hipStream_t st, fhe::NttIo io) {
    return sub_transform(true, d, nd, a, log_n, subs, pb, pm, st, io);
}

inline unsigned pass_grid(size_t total) {
    size_t b = (total + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b ? b : 1));
}

}  // namespace

namespace fhe {

// rings above 2^14 do not fit one workgroup's LDS: one radix-2^pb pass over HBM + 2^pb sub-transforms of 2^14.
// io counts POLYNOMIALS; the sub-transform launches take it in 2^14 blocks.
int ntt_fwd_multi(const ModDesc *descs, unsigned n_desc, u64 *a, int log_n, size_t batch, hipStream_t st, int pm, NttIo io) {
    if (io.mul) return FHE_ERR_INVALID;  // the pointwise multiplier belongs to the inverse
    if (log_n <= 15) return sub_fwd(descs, n_desc, a, log_n, batch, 0, pm, st, io);
    if (io.src_group) return FHE_ERR_UNSUPPORTED;  // several sources: single-pass rings only (callers fall back to one launch each)
    const int pb = log_n - 14;
    const size_t cols = batch << 14;
    switch (pb) {  // the opening pass reads the source, everything after it runs in place
        case 1: hipLaunchKernelGGL(ntt_big_fwd_pass<1>, dim3(pass_grid(cols)), dim3(256), 0, st, a, descs, n_desc, (unsigned)batch, log_n, io.src, io.src_mod); break;
        case 2: hipLaunchKernelGGL(ntt_big_fwd_pass<2>, dim3(pass_grid(cols)), dim3(256), 0, st, a, descs, n_desc, (unsigned)batch, log_n, io.src, io.src_mod); break;
        case 3: hipLaunchKernelGGL(ntt_big_fwd_pass<3>, dim3(pass_grid(cols)), dim3(256), 0, st, a, descs, n_desc, (unsigned)batch, log_n, io.src, io.src_mod); break;
        default: return FHE_ERR_UNSUPPORTED;
    }
    HIP_TRY(hipGetLastError());
    return sub_fwd(descs, n_desc, a, 14, batch << pb, pb, pm, st, NttIo());
}

int ntt_inv_multi(const ModDesc *descs, unsigned n_desc, u64 *a, int log_n, size_t batch, hipStream_t st, int pm, NttIo io) {
    if (io.src_group) return FHE_ERR_INVALID;
    if (log_n <= 15) return sub_inv(descs, n_desc, a, log_n, batch, 0, pm, st, io);
    const int pb = log_n - 14;
    NttIo sub_io = io;  // polynomial counts -> 2^14 block counts
    sub_io.src_mod = io.src_mod << pb; sub_io.mul_div = io.mul_div << pb; sub_io.mul_period = io.mul_period << pb;
    int rc = sub_inv(descs, n_desc, a, 14, batch << pb, pb, pm, st, sub_io);
    if (rc != FHE_OK) return rc;
    const size_t cols = batch << 14;
    switch (pb) {
        case 1: hipLaunchKernelGGL(ntt_big_inv_pass<1>, dim3(pass_grid(cols)), dim3(256), 0, st, a, descs, n_desc, (unsigned)batch, log_n); break;
        case 2: hipLaunchKernelGGL(ntt_big_inv_pass<2>, dim3(pass_grid(cols)), dim3(256), 0, st, a, descs, n_desc, (unsigned)batch, log_n); break;
        case 3: hipLaunchKernelGGL(ntt_big_inv_pass<3>, dim3(pass_grid(cols)), dim3(256), 0, st, a, descs, n_desc, (unsigned)batch, log_n); break;
        default: return FHE_ERR_UNSUPPORTED;
    }
    HIP_TRY(hipGetLastError());
    return FHE_OK;
}

// N = 2^15 with the outermost layer left to the caller (rns_kernels.hpp, the edge kernels of the key switch): what remains of
// every polynomial are two independent 2^14 sub-transforms -- the two-workgroups-per-CU kernel instead of the one-per-CU 2^15 one.
// Forward: the caller has applied layer 0; inverse: the caller applies layer 0 and n^-1.  io counts polynomials.
int ntt_fwd_inner15(const ModDesc *descs, unsigned n_desc, u64 *a, size_t batch, hipStream_t st, int pm, NttIo io) {
    if (io.mul) return FHE_ERR_INVALID;
    io.src_mod <<= 1;
    return sub_fwd(descs, n_desc, a, 14, batch << 1, 1, pm, st, io);
}
int ntt_inv_inner15(const ModDesc *descs, unsigned n_desc, u64 *a, size_t batch, hipStream_t st, int pm, NttIo io) {
    io.src_mod <<= 1; io.mul_div <<= 1; io.mul_period <<= 1;
    return sub_inv(descs, n_desc, a, 14, batch << 1, 1, pm, st, io);
}

// a[s] <- inverse(forward(a[s]) (.) io.mul[..]) in one launch where a fused kernel exists (2^13 .. 2^15), else forward + inverse
// with the product on the inverse's load.  io.mul as for ntt_inv_multi; io.src must be null.
int ntt_mul_multi(const ModDesc *descs, unsigned n_desc, u64 *a, int log_n, size_t batch, hipStream_t st, int pm, NttIo io) {
    if (!io.mul || io.src) return FHE_ERR_INVALID;
    const bool off = fhe::opt(fhe::OPT_NO_FUSED_MUL) != 0;  // lab switch (api_common.hpp)
    if (log_n >= 13 && log_n <= 15 && !off) {
        if (pm == 60) return dispatch_mul14<ArithDS<60>>(log_n, descs, n_desc, a, batch, st, io);
        if (pm == 54) return dispatch_mul14<ArithDS<54>>(log_n, descs, n_desc, a, batch, st, io);
        return dispatch_mul14<ArithShoup>(log_n, descs, n_desc, a, batch, st, io);
    }
    int rc = ntt_fwd_multi(descs, n_desc, a, log_n, batch, st, pm);
    return rc != FHE_OK ? rc : ntt_inv_multi(descs, n_desc, a, log_n, batch, st, pm, io);
}

int ntt_fwd_device(const fhe_ctx *c, u64 *a, int log_n, size_t batch, hipStream_t st) { return ntt_fwd_multi(c->d_desc, 1, a, log_n, batch, st, c->pm_b); }
int ntt_inv_device(const fhe_ctx *c, u64 *a, int log_n, size_t batch, hipStream_t st) { return ntt_inv_multi(c->d_desc, 1, a, log_n, batch, st, c->pm_b); }

int ctx_build_host(uint64_t q, fhe_ctx *c) {
    if (q < 3) return FHE_ERR_INVALID;
    if (!is_prime_u64(q)) return FHE_ERR_NOT_PRIME;
    if (q >> 62) return FHE_ERR_UNSUPPORTED;  // lazy [0, 4q) arithmetic needs q < 2^62
    c->q = q;
    c->s = __builtin_ctzll(q - 1);
    c->g = smallest_nonresidue(q);
    if (c->g == 0) return FHE_ERR_INVALID;  // q = 3: `Zq::generator` finds nothing in 1..q-1 and unwraps None (zq.rs:99-105)
    c->omega = powmod(c->g, (q - 1) >> c->s, q);
    // the reference table has 2^(s-1) entries, entry j = omega^bitrev_{s-1}(j).  Bit-reversed tables are
    // prefix-nested, so the first 2^k entries equal psi^bitrev_k(j) with psi = omega^(2^(s-1-k)).
    c->log_cap = c->s - 1 < MAX_LOG_N ? c->s - 1 : MAX_LOG_N;
    const size_t cap = size_t(1) << c->log_cap;
    const u64 psi = powmod(c->omega, u64(1) << (c->s - 1 - c->log_cap), q);
    const u64 psi_inv = invmod(psi, q);
    std::vector<uint64_t> pw(cap), pwi(cap);
    uint64_t x = 1, y = 1;
    for (size_t i = 0; i < cap; ++i) {
        pw[i] = x; pwi[i] = y;
        x = mulmod(x, psi, q); y = mulmod(y, psi_inv, q);
    }
    c->tw.resize(cap); c->twi.resize(cap);
    for (size_t j = 0; j < cap; ++j) {
        size_t r = bitrev((unsigned)j, c->log_cap);
        c->tw[j] = pw[r]; c->twi[j] = pwi[r];
    }
    for (int k = 0; k < 32; ++k) {
        uint64_t n = (uint64_t(1) << k) % q;
        c->ninv[k] = n ? invmod(n, q) : 0;
        c->ninv_s[k] = shoup(c->ninv[k], q);
        c->ninv_w[k] = (n && c->twi.size() > 1) ? mulmod(c->ninv[k], c->twi[1], q) : 0;
        c->ninv_w_s[k] = shoup(c->ninv_w[k], q);
    }
    int nbits = 64 - __builtin_clzll(q);
    // pseudo-Mersenne eligibility (arith.hpp): q = 2^b - c with 33 <= b <= 60 and c <= 2^(b-33)
    c->pm_b = 0; c->pm_c = 0;
    if (nbits >= 34 && nbits <= 60) {
        const uint64_t cc = (uint64_t(1) << nbits) - q;
        if (cc <= (uint64_t(1) << (nbits - 33))) { c->pm_b = nbits; c->pm_c = (unsigned)cc; }
    }
    c->barrett = fhe::make_barrett(q);
    return FHE_OK;
}

}  // namespace fhe

extern "C" {

const char *fhe_version(void) { return "learn-fhe_amd 0.1 (gfx950)"; }
int fhe_last_hip_error(void) { return g_last_hip; }

// hand the library's cached stream-ordered scratch (api_common.hpp: one private pool per device) back to the driver
int fhe_trim(void) {
    int rc = FHE_OK;
    std::lock_guard<std::mutex> lock(fhe::pool_mutex());
    for (int dev = 0; dev < fhe::MAX_DEVICES; ++dev)
        if (fhe::g_pools[dev] && hipMemPoolTrimTo(fhe::g_pools[dev], 0) != hipSuccess) rc = FHE_ERR_HIP;
    return rc;
}

// lab switches (api_common.hpp): name without the FHE_RING_ prefix, upper or lower case
int fhe_set_option(const char *name, long value) {
    if (!name) return FHE_ERR_INVALID;
    for (int i = 0; i < fhe::OPT_COUNT; ++i) {
        const char *a = name, *b = fhe::OPT_NAMES[i];
        while (*a && *b && (*a == *b || *a - 32 == *b)) { ++a; ++b; }
        if (!*a && !*b) { fhe::options().v[i].store(value, std::memory_order_relaxed); return FHE_OK; }
    }
    return FHE_ERR_INVALID;
}

int fhe_is_prime(uint64_t q) { return fhe::is_prime_u64(q) ? 1 : 0; }

int fhe_two_adic_primes(int bits, int log_n, int count, uint64_t *out) {
    if (!out || bits <= log_n || bits > 63 || log_n < 0 || count < 0) return 0;
    uint64_t lo = 1ull << (bits - log_n - 1), hi = 1ull << (bits - log_n);
    int found = 0;
    for (uint64_t k = hi - 1; k >= lo && found < count; --k) {
        uint64_t c = (k << log_n) + 1;
        if (fhe::is_prime_u64(c)) out[found++] = c;
    }
    return found;
}

int fhe_ctx_create(uint64_t q, int device, fhe_ctx **out) {
    if (!out) return FHE_ERR_INVALID;
    *out = nullptr;
    fhe_ctx *c = new (std::nothrow) fhe_ctx();
    if (!c) return FHE_ERR_INVALID;
    int rc = fhe::ctx_build_host(q, c);
    if (rc != FHE_OK) { delete c; return rc; }
    c->device = device;
    if (device >= 0) {
        DeviceGuard guard(device);
        if (!guard.ok) { delete c; return FHE_ERR_HIP; }
        const size_t cap = c->tw.size();
        std::vector<fhe::TwPair> pf(cap), pi(cap);
        for (size_t j = 0; j < cap; ++j) {
            pf[j] = {c->tw[j], fhe::shoup(c->tw[j], q)};
            pi[j] = {c->twi[j], fhe::shoup(c->twi[j], q)};
        }
        hipError_t e = hipMalloc(&c->d_tw, cap * sizeof(fhe::TwPair));
        if (e == hipSuccess) e = hipMalloc(&c->d_twi, cap * sizeof(fhe::TwPair));
        if (e == hipSuccess) e = hipMemcpy(c->d_tw, pf.data(), cap * sizeof(fhe::TwPair), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->d_twi, pi.data(), cap * sizeof(fhe::TwPair), hipMemcpyHostToDevice);
        if (e == hipSuccess && c->pm_b) {
            // twiddles in the 8-byte form the pseudo-Mersenne policy loads: {wl, wh}, w = wl + wh 2^(b-31) (arith.hpp PmRaw)
            const int b = c->pm_b;
            std::vector<uint64_t> lf(cap), li(cap);
            auto pack = [b](uint64_t w) { return ((w >> (b - 31)) << 32) | (w & ((uint64_t(1) << (b - 31)) - 1)); };
            for (size_t j = 0; j < cap; ++j) { lf[j] = pack(c->tw[j]); li[j] = pack(c->twi[j]); }
            e = hipMalloc((void **)&c->d_tww, 2 * cap * sizeof(uint64_t));
            c->d_twwi = c->d_tww + cap;
            if (e == hipSuccess) e = hipMemcpy(c->d_tww, lf.data(), cap * sizeof(uint64_t), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(c->d_twwi, li.data(), cap * sizeof(uint64_t), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess && c->pm_b) {
            // ... and in the 16-byte two-operand form {a0, a1, b0, b1} of ArithDS (w and w 2^32 mod q, each split at b - 31 bits)
            const int b = c->pm_b;
            std::vector<uint4> df(cap), di(cap);
            auto split = [b, q](uint64_t w) {
                const uint64_t w1 = (uint64_t)((((unsigned __int128)w) << 32) % q), lo = (uint64_t(1) << (b - 31)) - 1;
                return uint4{(unsigned)(w & lo), (unsigned)(w >> (b - 31)), (unsigned)(w1 & lo), (unsigned)(w1 >> (b - 31))};
            };
            for (size_t j = 0; j < cap; ++j) { df[j] = split(c->tw[j]); di[j] = split(c->twi[j]); }
            for (int k = 0; k < 20; ++k) { c->h_desc.ds_ninv[k] = split(c->ninv[k]); c->h_desc.ds_ninv_w[k] = split(c->ninv_w[k]); }
            c->h_desc.ds_one = split(1);
            e = hipMalloc((void **)&c->d_twd, 2 * cap * sizeof(uint4));
            c->d_twdi = c->d_twd + cap;
            if (e == hipSuccess) e = hipMemcpy(c->d_twd, df.data(), cap * sizeof(uint4), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(c->d_twdi, di.data(), cap * sizeof(uint4), hipMemcpyHostToDevice);
            // diagonal multipliers of the inverse's first pass (ntt14w.hpp): twd3i[(p - 1) stride + blk] = twi[4 blk]^p, p = 1..7
            if (e == hipSuccess && cap >= 8) {
                const size_t stride = cap / 4;
                std::vector<uint4> d3(7 * stride, uint4{0, 0, 0, 0});
                for (size_t blk = 1; blk < stride; ++blk) {
                    const uint64_t th = c->twi[4 * blk];
                    uint64_t pw = th;
                    for (int p = 0; p < 7; ++p) { d3[p * stride + blk] = split(pw); pw = fhe::mulmod(pw, th, q); }
                }
                e = hipMalloc((void **)&c->d_twd3i, d3.size() * sizeof(uint4));
                if (e == hipSuccess) e = hipMemcpy(c->d_twd3i, d3.data(), d3.size() * sizeof(uint4), hipMemcpyHostToDevice);
                c->h_desc.twd3i = c->d_twd3i; c->h_desc.twd3_stride = (unsigned)stride;
            }
        }
        c->h_desc.tww = c->d_tww;
        c->h_desc.twwi = c->d_twwi;
        c->h_desc.twd = c->d_twd;
        c->h_desc.twdi = c->d_twdi;
        c->h_desc.ds_pow = c->pm_b ? 1u << (c->pm_b - 31) : 0;
        c->h_desc.pm_b = c->pm_b;
        c->h_desc.pm_c = c->pm_c;
        c->h_desc.q = q;
        c->h_desc.tw = c->d_tw;
        c->h_desc.twi = c->d_twi;
        for (int k = 0; k < 20; ++k) {
            c->h_desc.ninv[k] = c->ninv[k]; c->h_desc.ninv_s[k] = c->ninv_s[k];
            c->h_desc.ninv_w[k] = c->ninv_w[k]; c->h_desc.ninv_w_s[k] = c->ninv_w_s[k];
        }
        c->h_desc.one_s = fhe::shoup(1, q);
        c->h_desc.bar_mu = c->barrett.mu; c->h_desc.bar_sh1 = c->barrett.sh1; c->h_desc.bar_sh2 = c->barrett.sh2;
        if (e == hipSuccess) e = hipMalloc(&c->d_desc, sizeof(fhe::ModDesc));
        if (e == hipSuccess) e = hipMemcpy(c->d_desc, &c->h_desc, sizeof(fhe::ModDesc), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            g_last_hip = (int)e;
            fhe_ctx_destroy(c);
            return FHE_ERR_HIP;
        }
    }
    *out = c;
    return FHE_OK;
}

void fhe_ctx_destroy(fhe_ctx *c) {
    if (!c) return;
    if (c->device >= 0) {
        DeviceGuard guard(c->device);
        if (c->d_tw) (void)hipFree(c->d_tw);
        if (c->d_twi) (void)hipFree(c->d_twi);
        if (c->d_desc) (void)hipFree(c->d_desc);
        if (c->d_tww) (void)hipFree(c->d_tww);
        if (c->d_twd) (void)hipFree(c->d_twd);
        if (c->d_twd3i) (void)hipFree(c->d_twd3i);
    }
    delete c;
}

int fhe_ctx_info(const fhe_ctx *c, uint64_t *q, int *s, uint64_t *g, uint64_t *omega) {
    if (!c) return FHE_ERR_INVALID;
    if (q) *q = c->q;
    if (s) *s = c->s;
    if (g) *g = c->g;
    if (omega) *omega = c->omega;
    return FHE_OK;
}

int fhe_ctx_twiddles(const fhe_ctx *c, int inverse, uint64_t *out, size_t count) {
    if (!c || !out) return FHE_ERR_INVALID;
    const auto &t = inverse ? c->twi : c->tw;
    if (count > t.size()) return FHE_ERR_UNSUPPORTED;
    std::memcpy(out, t.data(), count * sizeof(uint64_t));
    return FHE_OK;
}

static int transform_entry(const fhe_ctx *ctx, uint64_t *a, size_t n, size_t batch, fhe_mem mem, void *stream, bool inverse) {
    int rc = check_transform(ctx, a, n, batch);
    if (rc != FHE_OK) return rc;
    if (ctx->device < 0) return FHE_ERR_NO_DEVICE;
    if (n == 1 || batch == 0) return FHE_OK;
    if (batch > 0xffffffffull) return FHE_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    Mirror m(a, n * batch, mem, true, st);
    if (m.rc != FHE_OK) return m.rc;
    rc = inverse ? fhe::ntt_inv_device(ctx, m.d, ilog2(n), batch, st) : fhe::ntt_fwd_device(ctx, m.d, ilog2(n), batch, st);
    if (rc != FHE_OK) return rc;
    return m.sync_out(st);
}

int fhe_ntt_fwd(const fhe_ctx *ctx, uint64_t *a, size_t n, size_t batch, fhe_mem mem, void *stream) {
    return transform_entry(ctx, a, n, batch, mem, stream, false);
}

int fhe_ntt_inv(const fhe_ctx *ctx, uint64_t *a, size_t n, size_t batch, fhe_mem mem, void *stream) {
    return transform_entry(ctx, a, n, batch, mem, stream, true);
}

static int launch_pointwise(const fhe_ctx *ctx, u64 *a, const u64 *b, size_t len, hipStream_t st) {
    size_t blocks = (len + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fhe::pointwise_mul_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, len, ctx->barrett);
    HIP_TRY(hipGetLastError());
    return FHE_OK;
}

int fhe_pointwise_mul(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, size_t len, fhe_mem mem, void *stream) {
    if (!ctx || ((!a || !b) && len)) return FHE_ERR_INVALID;
    if (ctx->device < 0) return FHE_ERR_NO_DEVICE;
    if (len == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    Mirror ma(a, len, mem, true, st), mb(b, len, mem, true, st);
    if (ma.rc != FHE_OK || mb.rc != FHE_OK) return FHE_ERR_HIP;
    int rc = launch_pointwise(ctx, ma.d, mb.d, len, st);
    if (rc != FHE_OK) return rc;
    return ma.sync_out(st);
}

namespace {
int rq_elementwise(uint64_t q, const void *a, const void *b, uint64_t *out, size_t len, int op, uint64_t scalar, fhe_mem mem, void *stream) {
    PtrDeviceGuard pguard(out, mem);
    if (!pguard.ok) return FHE_ERR_HIP;
    if (q < 2 || (q >> 62) || ((!a || !out || (op < 2 && !b)) && len) || (op == 3 && scalar >= q)) return FHE_ERR_INVALID;
    if (len == 0) return FHE_OK;
    hipStream_t st = (hipStream_t)stream;
    Mirror ma(a, len, mem, true, st), mb(op < 2 ? b : nullptr, op < 2 ? len : 0, mem, true, st), mo(out, len, mem, false, st);
    if (ma.rc | mb.rc | mo.rc) return FHE_ERR_HIP;
    hipLaunchKernelGGL(fhe::rq_elementwise_kernel, dim3(pass_grid(len)), dim3(256), 0, st, (const u64 *)ma.d, (const u64 *)mb.d, mo.d, len, op,
                       (u64)scalar, fhe::make_barrett(q));
    HIP_TRY(hipGetLastError());
    return mo.sync_out(st);
}
}  // namespace

int fhe_rq_add(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t len, fhe_mem mem, void *stream) {
    return rq_elementwise(q, a, b, out, len, 0, 0, mem, stream);
}
int fhe_rq_sub(uint64_t q, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t len, fhe_mem mem, void *stream) {
    return rq_elementwise(q, a, b, out, len, 1, 0, mem, stream);
}
int fhe_rq_neg(uint64_t q, const uint64_t *a, uint64_t *out, size_t len, fhe_mem mem, void *stream) {
    return rq_elementwise(q, a, nullptr, out, len, 2, 0, mem, stream);
}
int fhe_rq_scalar_mul(uint64_t q, const uint64_t *a, uint64_t scalar, uint64_t *out, size_t len, fhe_mem mem, void *stream) {
    return rq_elementwise(q, a, nullptr, out, len, 3, scalar, mem, stream);
}
int fhe_rq_from_i64(uint64_t q, const int64_t *in, uint64_t *out, size_t len, fhe_mem mem, void *stream) {
    return rq_elementwise(q, in, nullptr, out, len, 4, 0, mem, stream);
}

int fhe_ntt_mul(const fhe_ctx *ctx, uint64_t *a, const uint64_t *b, size_t n, size_t batch, fhe_mem mem,
                void *stream) {
    int rc = check_transform(ctx, a, n, batch);
    if (rc != FHE_OK) return rc;
    if (!b && batch) return FHE_ERR_INVALID;
    if (ctx->device < 0) return FHE_ERR_NO_DEVICE;
    if (batch == 0) return FHE_OK;
    if (batch > 0xffffffffull) return FHE_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return FHE_ERR_HIP;
    const size_t count = n * batch;
    const int log_n = ilog2(n);
    Mirror ma(a, count, mem, true, st);
    if (ma.rc != FHE_OK) return ma.rc;
    // util/src/ring/fft/zq.rs:14-19 in three launches and 56 N bytes of HBM traffic instead of the reference's shape (copy of b,
    // two in-place transforms, a pointwise pass, the inverse: 88 N): b is transformed OUT OF PLACE into the scratch the reference
    // allocates too (fft/zq.rs:21-25), and the pointwise product rides on the load of the inverse transform.
    StreamWs ws(count * sizeof(u64), st);
    if (ws.rc != FHE_OK) return ws.rc;
    u64 *tb = ws.as<u64>();
    if (n > 1 && mem == FHE_MEM_DEVICE) {
        fhe::NttIo src_b;
        src_b.src = (const u64 *)b; src_b.src_mod = (unsigned)batch;
        rc = fhe::ntt_fwd_multi(ctx->d_desc, 1, tb, log_n, batch, st, ctx->pm_b, src_b);
    } else {  // host operand (or n = 1): the copy is the upload
        hipMemcpyKind kind = mem == FHE_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        rc = hipMemcpyAsync(tb, b, count * sizeof(u64), kind, st) == hipSuccess ? FHE_OK : FHE_ERR_HIP;
        if (rc == FHE_OK && n > 1) rc = fhe::ntt_fwd_device(ctx, tb, log_n, batch, st);
    }
    if (rc == FHE_OK && n > 1) {  // forward of a, product, inverse: one launch at 2^13 .. 2^15 (ntt14w_mul_kernel: 40 N bytes in all)
        fhe::NttIo mul_b;
        mul_b.mul = tb; mul_b.mul_div = (unsigned)batch; mul_b.mul_period = (unsigned)batch;
        rc = fhe::ntt_mul_multi(ctx->d_desc, 1, ma.d, log_n, batch, st, ctx->pm_b, mul_b);
    } else if (rc == FHE_OK) {
        rc = launch_pointwise(ctx, ma.d, tb, count, st);  // n = 1: `a[0] *= b[0]`
    }
    if (rc == FHE_OK) rc = ma.sync_out(st);
    return rc;  // tb is released in stream order
}

}  // extern "C"
