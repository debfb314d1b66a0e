// What the torus entry points share (torus_api.hip: k = 1, fused kernels; torusk_api.hip: any rank k, composed kernels): the
// context and prepared-key structures and two small host helpers.
#pragma once
#include <hip/hip_runtime.h>

#include "api_common.hpp"
#include "ctx.hpp"
#include "torus_kernels.hpp"
#include "torus30_kernels.hpp"
#include "torusf_kernels.hpp"

struct fhe_torus_ctx {
    int device = -1;
    fhe_ctx *mods[2] = {nullptr, nullptr};
    fhe::ModDesc *d_descs = nullptr;  // [2]
    fhe::TorusConsts T{};
    // the three-prime 30-bit path (torus30_kernels.hpp)
    void *d_blob30 = nullptr;         // twiddle tables + descriptors
    fhe::Torus30Consts T30{};
    // the f64 FFT mode (torusf_kernels.hpp): T[2^l + i] = cis(pi (2 bitrev_l(i) + 1) / 2^(l+1)), 2^11 entries
    double2 *d_twf = nullptr;
};

struct fhe_tggsw_key {
    const fhe_torus_ctx *t = nullptr;
    int log_n = 0, log_b = 0, d = 0;
    size_t count = 0;
    u64 *d_rows[2] = {nullptr, nullptr};  // per prime: [count][2d][2][N] evaluation domain, key_perm layout
    unsigned *d_rows30 = nullptr;         // 30-bit path: [3 primes][count][2d][2][N] Montgomery residues, key_perm30 layout
    double2 *d_rowsf = nullptr;           // fft64 mode: [count][2d][2][N / 2] complex evaluations (torusf_kernels.hpp)
    double2 *d_rowsx3 = nullptr;          // exact through f64 transforms: [count][2d][2][3 pieces][N / 2] complex evaluations (torusf_kernels.hpp)
    fhe::TDecomp P{};
};

namespace {
// the two largest pseudo-Mersenne primes of two_adic_primes(60, 16): rings up to N = 2^15
constexpr uint64_t TORUS_P0 = 1152921504606584833ull, TORUS_P1 = 1152921504598720513ull;

inline int make_tdecomp(int log_b, int d, fhe::TDecomp *P) {
    if (log_b < 1 || log_b > 63 || d < 1 || d > 64) return FHE_ERR_INVALID;
    const int rb = 64 - log_b * d > 0 ? 64 - log_b * d : 0;
    if (rb >= 64) return FHE_ERR_INVALID;
    P->rnd = (uint64_t(1) << rb) >> 1;
    P->mask = (uint64_t(1) << log_b) - 1;
    P->log_b = log_b;
    P->d = d;
    P->rb = rb;
    return FHE_OK;
}

inline unsigned grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b ? b : 1));
}
}  // namespace
