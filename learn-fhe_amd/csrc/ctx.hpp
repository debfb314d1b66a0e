// Per-modulus context: what the reference keeps in its global twiddle cache
// (util/src/ring/fft/zq.rs:38-67), built once, immutable, plus the HBM copies the kernels read.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "dev_arith.hpp"
#include "modmath.hpp"
#include "ntt_kernels.hpp"

struct fhe_ctx {
    uint64_t q = 0;
    int s = 0;           // trailing_zeros(q - 1)
    uint64_t g = 0;      // smallest quadratic non-residue
    uint64_t omega = 0;  // g^((q-1) >> s), order 2^s
    int log_cap = 0;     // tables hold 2^log_cap entries = min(2^(s-1), 2^FHE_MAX_LOG_N)
    std::vector<uint64_t> tw, twi;  // reference layout: entry j = omega^(+-bitrev_{s-1}(j))
    uint64_t ninv[32] = {0}, ninv_s[32] = {0};  // (2^k)^-1 mod q and its Shoup companion
    uint64_t ninv_w[32] = {0}, ninv_w_s[32] = {0};  // (2^k)^-1 * twi[1] mod q and its Shoup companion
    fhe::Barrett barrett{};
    int device = -1;
    fhe::TwPair *d_tw = nullptr, *d_twi = nullptr;  // {w, floor(w 2^64 / q)} pairs in HBM
    u64 *d_tww = nullptr, *d_twwi = nullptr;        // plain twiddles for the pseudo-Mersenne path (null if not eligible)
    uint4 *d_twd = nullptr, *d_twdi = nullptr;      // the same twiddles in the two-operand split form (ArithDS, arith.hpp)
    uint4 *d_twd3i = nullptr;                       // ModDesc::twd3i: [7][cap / 4] powers of the last-layer inverse twiddles (arith.hpp)
    int pm_b = 0;                                   // q = 2^pm_b - pm_c, or 0
    unsigned pm_c = 0;
    fhe::ModDesc *d_desc = nullptr;                 // this modulus as the kernels read it (1 entry)
    fhe::ModDesc h_desc{};
};

namespace fhe {
constexpr int MAX_LOG_N = 17;
int ctx_build_host(uint64_t q, fhe_ctx *c);  // returns FHE_* status
// batched transforms over `batch` polynomials of degree 2^log_n; polynomial p uses descs[p % n_desc]
// pm: common bit length b if EVERY descriptor is pseudo-Mersenne eligible with the same b (the N = 2^14 kernels then use
// ArithPM<b> when instantiated), else 0
// io (arith.hpp NttIo): optional out-of-place source / fused pointwise multiplier, indices in POLYNOMIALS of this launch
int ntt_fwd_multi(const ModDesc *descs, unsigned n_desc, u64 *a, int log_n, size_t batch, hipStream_t st, int pm = 0, NttIo io = NttIo());
int ntt_inv_multi(const ModDesc *descs, unsigned n_desc, u64 *a, int log_n, size_t batch, hipStream_t st, int pm = 0, NttIo io = NttIo());
// a <- inverse(forward(a) (.) io.mul): one fused launch at 2^13 .. 2^15, forward + multiplying inverse otherwise
int ntt_mul_multi(const ModDesc *descs, unsigned n_desc, u64 *a, int log_n, size_t batch, hipStream_t st, int pm, NttIo io);
// 2^15 rings without their outermost layer (two independent 2^14 sub-transforms per polynomial): see ring_api.hip
int ntt_fwd_inner15(const ModDesc *descs, unsigned n_desc, u64 *a, size_t batch, hipStream_t st, int pm, NttIo io = NttIo());
int ntt_inv_inner15(const ModDesc *descs, unsigned n_desc, u64 *a, size_t batch, hipStream_t st, int pm, NttIo io = NttIo());
}  // namespace fhe
