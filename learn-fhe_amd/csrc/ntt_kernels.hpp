// Batched negacyclic NTT / inverse NTT kernels for gfx950 (MI355X).
//
// One polynomial is owned by T = N / E threads (E = 2^LOG_E coefficients per thread, kept in
// registers); PPW polynomials share a workgroup when N is small.  The log2(N) butterfly layers are
// cut into passes of up to LOG_E layers; inside a pass a thread owns complete radix-2^R butterfly
// groups, so the layers of a pass run entirely in registers.  Between passes the coefficients are
// exchanged through an LDS image of the polynomial (padded by one 8-byte slot per 16 coefficients,
// which makes the strided pass patterns bank-conflict free for ds_read/write_b64).  HBM is touched
// exactly once per coefficient per direction: 16*N algorithmic bytes per transform.
//
// Layer/twiddle order follows the reference exactly (util/src/ring/fft.rs:40-77): forward =
// Cooley-Tukey layers 0..logN-1 with twiddle tw[2^layer + i], natural in / bit-reversed out;
// inverse = Gentleman-Sande layers logN-1..0 with twi[2^layer + i], then * n^-1.
#pragma once
#include "arith.hpp"

namespace fhe {

template <int LOG_N, int LOG_E, int PPW>
struct NttCfg {
    static constexpr int N = 1 << LOG_N;
    static constexpr int E = 1 << LOG_E;
    static constexpr int T = N / E;                             // threads per polynomial
    static constexpr int THREADS = T * PPW;                     // workgroup size
    static constexpr int P = (LOG_N + LOG_E - 1) / LOG_E;       // number of passes
    static constexpr int R0 = LOG_N - (P - 1) * LOG_E;          // layers in the short pass (layers 0..R0-1)
    static constexpr int PN = N + (N >> 4);                     // padded LDS image, in u64
    static constexpr size_t LDS_BYTES = (P > 1) ? size_t(PN) * PPW * 8 : 0;
    static_assert(LOG_E >= 1 && LOG_E <= LOG_N, "bad LOG_E");
    static_assert(THREADS >= 1 && THREADS <= 1024, "bad workgroup size");
};

__device__ __forceinline__ int lds_phys(int i) { return i + (i >> 4); }

// DIRECT (template flag of the transform kernels): the pass that owns 2^LOG_E contiguous coefficients per thread
// (last forward pass, first inverse pass) moves them between registers and HBM itself with 16-byte accesses
// (each thread owns whole 128-byte lines) instead of staging the polynomial through LDS once more.  Measured on
// MI355X (tools/ntt_lab.hip): a win for the inverse's loads, a loss for the forward's stores.

// Developer-lab ablation hooks (tools/ntt_lab.hip); never defined in the product build.
#ifdef FHE_ABLATE_NO_GLOBAL
__device__ __forceinline__ u64 gload(const u64 *p, int i) { return (u64)i * 0x9E3779B97F4A7C15ull >> 4; }
__device__ __forceinline__ void gstore(u64 *p, int i, u64 v) { if (v == 0xdeadbeefcafef00dull) p[i] = v; }
#else
__device__ __forceinline__ u64 gload(const u64 *p, int i) { return p[i]; }
__device__ __forceinline__ void gstore(u64 *p, int i, u64 v) { p[i] = v; }
#endif

// element index of register r of butterfly group `grp` in the pass that covers layers [L0, L0+R)
template <int LOG_N, int L0, int R>
__device__ __forceinline__ int pass_index(int grp, int r) {
    constexpr int h = LOG_N - L0 - R;
    const int top = grp >> h, low = grp & ((1 << h) - 1);
    return (top << (h + R)) | (r << h) | low;
}

// wave-uniform `top` can be moved to an SGPR so the twiddle loads become scalar loads
template <int LOG_N, int L0, int R, int T>
__device__ __forceinline__ int pass_top(int grp) {
    constexpr int h = LOG_N - L0 - R;
    int top = grp >> h;
    if constexpr (h >= 6 && T >= 64) top = __builtin_amdgcn_readfirstlane(top);
    return top;
}

// exchange-point synchronisation: a workgroup barrier when several waves share the polynomial, only an LDS
// drain when ONE wave owns it (LDS operations of a single wave complete in order)
template <bool WAVE>
__device__ __forceinline__ void exchange_sync() {
    if constexpr (WAVE) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    else __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// forward transform: passes run L0 = 0, R0, R0+LOG_E, ...
// REGS_IO: the caller supplies x[] in the first pass's layout (pass_index<LOG_N,0,R0>(t + T*gg, r) at
// x[gg*2^R0 + r]) and gets the evaluations back in x[] in the last pass's layout (coefficient t*E + r at
// x[r]), still lazy in [0, 4q); nothing touches global memory.
// ---------------------------------------------------------------------------------------------
template <class A, typename C, int LOG_N, int LOG_E, int L0, bool REGS_IO = false, bool WAVE = false, bool DIRECT = false>
__device__ __forceinline__ void fwd_run(typename A::Elem (&x)[1 << LOG_E], int t, typename A::Elem *__restrict__ g, typename A::Elem *lds,
                                        bool active,
                                        const typename A::K &k, const typename A::Elem *__restrict__ gsrc = nullptr) {
    constexpr int E = 1 << LOG_E;
    constexpr int R = (L0 == 0) ? C::R0 : LOG_E;
    constexpr int G = E >> R;  // butterfly groups per thread in this pass
    constexpr bool first = (L0 == 0), last = (L0 + R == LOG_N);
    if constexpr (first && !REGS_IO) { if (!gsrc) gsrc = g; }  // out-of-place forward: read there, write here
    // load
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
        const int grp = t + C::T * gg;
#pragma unroll
        for (int r = 0; r < (1 << R); ++r) {
            const int i = pass_index<LOG_N, L0, R>(grp, r);
            if constexpr (first) {
                if constexpr (!REGS_IO) x[gg * (1 << R) + r] = active ? gload(gsrc, i) : 0;
            } else {
                x[gg * (1 << R) + r] = lds[lds_phys(i)];
            }
        }
    }
    // A register-resident transform (REGS_IO) is one of a chain that reuses this LDS image.  Inside a transform a thread only
    // ever overwrites elements it read itself, but the NEXT user's first writes follow another ownership layout: when several
    // waves share the image they must all have finished this last read first.  (One wave: its LDS operations complete in order.)
    if constexpr (last && !first && REGS_IO && !WAVE) __syncthreads();
    // butterflies; lazy values are folded back at a pass boundary only when the policy's reduction-free budget (CT_LAYERS) is
    // smaller than the whole transform: 54-bit pseudo-Mersenne moduli run all log2 N <= 13 layers without a single fold
    // (values < (2 log2 N + 1) q < 2^59), 60-bit ones fold at every boundary
    if constexpr (!first && A::PASS_FOLD && A::CT_LAYERS < LOG_N) {
#pragma unroll
        for (int r = 0; r < E; ++r) x[r] = A::fold(x[r], k);
    }
    static_for<0, G>([&](auto gg_c) {
        constexpr int gg = decltype(gg_c)::value;
        const int top = pass_top<LOG_N, L0, R, C::T>(t + C::T * gg);
#ifndef FHE_ABLATE_NO_COMPUTE
        ct_net<A, L0, R, gg * (1 << R), E>(x, top, k);
#endif
    });
    // store
    if constexpr (last && REGS_IO) {
        return;
    } else if constexpr (last && C::P == 1) {
#pragma unroll
        for (int gg = 0; gg < G; ++gg)
#pragma unroll
            for (int r = 0; r < (1 << R); ++r)
                if (active) gstore(g, pass_index<LOG_N, L0, R>(t + C::T * gg, r), A::canon_fwd(x[gg * (1 << R) + r], k));
    } else if constexpr (last && DIRECT) {
        // h == 0: register r holds coefficient (t << LOG_E) + r
        static_assert(G == 1, "last pass is a full pass");
        if (active) {
            ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(g + (size_t(t) << LOG_E));
#pragma unroll
            for (int r = 0; r < E; r += 2) {
                ulonglong2 v;
                v.x = A::canon_fwd(x[r], k);
                v.y = A::canon_fwd(x[r + 1], k);
#ifdef FHE_ABLATE_NO_GLOBAL
                if (v.x == 0xdeadbeefcafef00dull) dst[r >> 1] = v;
#else
                dst[r >> 1] = v;
#endif
            }
        }
    } else {
#pragma unroll
        for (int gg = 0; gg < G; ++gg)
#pragma unroll
            for (int r = 0; r < (1 << R); ++r) {
                typename A::Elem v = x[gg * (1 << R) + r];
                if constexpr (last) v = A::canon_fwd(v, k);
                lds[lds_phys(pass_index<LOG_N, L0, R>(t + C::T * gg, r))] = v;
            }
        exchange_sync<WAVE>();
        if constexpr (!last) fwd_run<A, C, LOG_N, LOG_E, L0 + R, REGS_IO, WAVE, DIRECT>(x, t, g, lds, active, k);
    }
}

// data: `subs` sub-polynomials of size 2^LOG_N, contiguous; sub s belongs to polynomial s >> pb (modulus descriptor
// (s >> pb) % n_desc) and is its sub-transform number s & (2^pb - 1).  pb = 0: plain transforms.
template <class A, int LOG_N, int LOG_E, int PPW, bool PFX = false, bool DIRECT = false>
__global__ __launch_bounds__((NttCfg<LOG_N, LOG_E, PPW>::THREADS)) void ntt_fwd_kernel(
    u64 *__restrict__ data, const ModDesc *__restrict__ descs, unsigned n_desc, unsigned subs, int pb_arg, NttIo io) {
    const int pb = PFX ? pb_arg : 0;  // PFX = false: every twiddle index folds to a constant expression
    using C = NttCfg<LOG_N, LOG_E, PPW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int t = threadIdx.x % C::T, pw = threadIdx.x / C::T;
    const unsigned poly = blockIdx.x * PPW + pw;
    const bool active = poly < subs;
    const ModDesc &D = descs[((active ? poly : 0u) >> pb) % n_desc];
    const typename A::K k = A::make(D, LOG_N, pb, int(poly & ((1u << pb) - 1)));
    u64 *g = data + size_t(poly) * C::N;
    u64 *lds = reinterpret_cast<u64 *>(smem_raw) + pw * C::PN;
    u64 x[C::E];
    fwd_run<A, C, LOG_N, LOG_E, 0, false, false, DIRECT>(x, t, g, lds, active, k, io.src ? ntt_src(io, active ? poly : 0u, LOG_N, nullptr) : nullptr);
    if constexpr (C::P > 1 && !DIRECT) {
        // the canonical image sits in LDS: stream it out with consecutive lanes on consecutive addresses
        if (active) {
#pragma unroll
            for (int kk = 0; kk < C::E; ++kk) {
                const int i = t + C::T * kk;
                gstore(g, i, lds[lds_phys(i)]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// inverse transform: passes run from the last layers back to layer 0
// ---------------------------------------------------------------------------------------------
// REGS_IO: x[] comes in in the last-layer pass layout (coefficient t*E + r at x[r], values in [0, 2q)) and
// leaves in the first-pass layout, multiplied by n^-1 and canonical.
template <class A, typename C, int LOG_N, int LOG_E, int LEND, bool REGS_IO = false, bool WAVE = false, bool DIRECT = false>  // layers [L0, LEND)
__device__ __forceinline__ void inv_run(typename A::Elem (&x)[1 << LOG_E], int t, typename A::Elem *__restrict__ g, typename A::Elem *lds,
                                        bool active,
                                        const typename A::K &k, const typename A::Elem *__restrict__ gsrc = nullptr,
                                        const typename A::Elem *__restrict__ gmul = nullptr) {
    constexpr int E = 1 << LOG_E;
    constexpr int R = (LEND == C::R0) ? C::R0 : LOG_E;
    if constexpr (LEND == LOG_N && !REGS_IO) { if (!gsrc) gsrc = g; }
    constexpr int L0 = LEND - R;
    constexpr int G = E >> R;
    constexpr bool final_pass = (L0 == 0);
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
        const int grp = t + C::T * gg;
#pragma unroll
        for (int r = 0; r < (1 << R); ++r) {
            const int i = pass_index<LOG_N, L0, R>(grp, r);
            if constexpr (REGS_IO && LEND == LOG_N) {
                (void)i;
            } else if constexpr (DIRECT && LEND == LOG_N && C::P > 1) {
                if ((r & 1) == 0) {
#ifdef FHE_ABLATE_NO_GLOBAL
                    x[r] = gload(g, i); x[r + 1] = gload(g, i + 1);
#else
                    const ulonglong2 v = active ? *reinterpret_cast<const ulonglong2 *>(gsrc + i) : ulonglong2{0, 0};
                    x[r] = v.x; x[r + 1] = v.y;
                    if (gmul && active) {
                        const ulonglong2 m = *reinterpret_cast<const ulonglong2 *>(gmul + i);
                        x[r] = A::mulvar(v.x, m.x, k); x[r + 1] = A::mulvar(v.y, m.y, k);
                    }
#endif
                }
            } else if constexpr (C::P == 1) {
                x[gg * (1 << R) + r] = active ? gload(gsrc, i) : 0;
                if (gmul && active) x[gg * (1 << R) + r] = A::mulvar(x[gg * (1 << R) + r], gload(gmul, i), k);
            } else {
                x[gg * (1 << R) + r] = lds[lds_phys(i)];
            }
        }
    }
    if constexpr (final_pass && C::P > 1 && REGS_IO && !WAVE) __syncthreads();  // as in fwd_run: the image is reused by the next transform
    static_for<0, G>([&](auto gg_c) {
        constexpr int gg = decltype(gg_c)::value;
        const int top = pass_top<LOG_N, L0, R, C::T>(t + C::T * gg);
#ifndef FHE_ABLATE_NO_COMPUTE
        gs_net<A, L0, R, gg * (1 << R), E>(x, top, k);
#endif
    });
    if constexpr (final_pass) {
        // * n^-1 (util/src/ring/fft.rs:76), canonicalise, store: consecutive lanes -> consecutive addresses
#pragma unroll
        for (int gg = 0; gg < G; ++gg)
#pragma unroll
            for (int r = 0; r < (1 << R); ++r) {
                const typename A::Elem v = A::finish_inv(x[gg * (1 << R) + r], k);
                if constexpr (REGS_IO) x[gg * (1 << R) + r] = v;
                else if (active) gstore(g, pass_index<LOG_N, L0, R>(t + C::T * gg, r), v);
            }
    } else {
#pragma unroll
        for (int gg = 0; gg < G; ++gg)
#pragma unroll
            for (int r = 0; r < (1 << R); ++r)
                lds[lds_phys(pass_index<LOG_N, L0, R>(t + C::T * gg, r))] = x[gg * (1 << R) + r];
        exchange_sync<WAVE>();
        inv_run<A, C, LOG_N, LOG_E, L0, REGS_IO, WAVE, DIRECT>(x, t, g, lds, active, k);
    }
}

template <class A, int LOG_N, int LOG_E, int PPW, bool PFX = false, bool DIRECT = true>
__global__ __launch_bounds__((NttCfg<LOG_N, LOG_E, PPW>::THREADS)) void ntt_inv_kernel(
    u64 *__restrict__ data, const ModDesc *__restrict__ descs, unsigned n_desc, unsigned subs, int pb_arg, NttIo io) {
    const int pb = PFX ? pb_arg : 0;
    using C = NttCfg<LOG_N, LOG_E, PPW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int t = threadIdx.x % C::T, pw = threadIdx.x / C::T;
    const unsigned poly = blockIdx.x * PPW + pw;
    const bool active = poly < subs;
    const ModDesc &D = descs[((active ? poly : 0u) >> pb) % n_desc];
    // with pb > 0 the n^-1 scaling belongs to the closing radix-2^pb pass (make() then scales by 1)
    const typename A::K k = A::make(D, LOG_N, pb, int(poly & ((1u << pb) - 1)));
    u64 *g = data + size_t(poly) * C::N;
    u64 *lds = reinterpret_cast<u64 *>(smem_raw) + pw * C::PN;
    u64 x[C::E];
    const u64 *gsrc = io.src ? io.src + size_t(poly % io.src_mod) * C::N : g;
    const u64 *gmul = io.mul ? io.mul + (size_t(poly / io.mul_div) * io.mul_period + poly % io.mul_period) * C::N : nullptr;
    if constexpr (C::P > 1 && !DIRECT) {
#pragma unroll
        for (int kk = 0; kk < C::E; ++kk) {
            const int i = t + C::T * kk;
            u64 v = active ? gload(gsrc, i) : 0;
            if (gmul && active) v = A::mulvar(v, gload(gmul, i), k);
            lds[lds_phys(i)] = v;
        }
        __syncthreads();
    }
    inv_run<A, C, LOG_N, LOG_E, LOG_N, false, false, DIRECT>(x, t, g, lds, active, k, gsrc, gmul);
}

// Opening radix-2^PB pass of a ring of degree 2^log_n (layers 0..PB-1 across the 2^PB blocks), one thread per
// column; leaves canonical values.  Followed by ntt_fwd_kernel<log_n - PB, ...> with pb = PB.
template <int PB>
__global__ void ntt_big_fwd_pass(u64 *__restrict__ data, const ModDesc *__restrict__ descs, unsigned n_desc, unsigned batch, int log_n,
                                 const u64 *__restrict__ src, unsigned src_mod) {
    const int lc = log_n - PB;  // log2 of the column count
    const size_t total = size_t(batch) << lc;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const unsigned poly = unsigned(idx >> lc);
        const size_t low = idx & ((size_t(1) << lc) - 1);
        const ModDesc &D = descs[poly % n_desc];
        const u64 q = D.q, q2 = 2 * q;
        u64 *g = data + (size_t(poly) << log_n) + low;
        const u64 *gs = src ? src + (size_t(poly % src_mod) << log_n) + low : g;
        u64 x[1 << PB];
#pragma unroll
        for (int r = 0; r < (1 << PB); ++r) x[r] = gs[size_t(r) << lc];
        const ArithShoup::K k = ArithShoup::make(D, log_n, 0, 0);
        ct_net<ArithShoup, 0, PB, 0, (1 << PB)>(x, 0, k);
#pragma unroll
        for (int r = 0; r < (1 << PB); ++r) g[size_t(r) << lc] = canon4(x[r], q, q2);
    }
}

// Closing pass of the inverse: layers PB-1..0 across the blocks, then * n^-1 (util/src/ring/fft.rs:76)
template <int PB>
__global__ void ntt_big_inv_pass(u64 *__restrict__ data, const ModDesc *__restrict__ descs, unsigned n_desc, unsigned batch, int log_n) {
    const int lc = log_n - PB;
    const size_t total = size_t(batch) << lc;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * blockDim.x) {
        const unsigned poly = unsigned(idx >> lc);
        const size_t low = idx & ((size_t(1) << lc) - 1);
        const ModDesc &D = descs[poly % n_desc];
        const u64 q = D.q;
        u64 *g = data + (size_t(poly) << log_n) + low;
        u64 x[1 << PB];
#pragma unroll
        for (int r = 0; r < (1 << PB); ++r) x[r] = g[size_t(r) << lc];
        const ArithShoup::K k = ArithShoup::make(D, log_n, 0, 0);
        gs_net<ArithShoup, 0, PB, 0, (1 << PB)>(x, 0, k);
#pragma unroll
        for (int r = 0; r < (1 << PB); ++r) g[size_t(r) << lc] = csub(mul_shoup_lazy(x[r], D.ninv[log_n], D.ninv_s[log_n], q), q);
    }
}

// a[i] <- a[i] * b[i] mod q (evaluation-domain product, util/src/ring.rs:266-270)
FHE_HEADER_KERNEL void pointwise_mul_kernel(u64 *__restrict__ a, const u64 *__restrict__ b, size_t len, Barrett B) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < len; i += size_t(gridDim.x) * blockDim.x)
        a[i] = mulmod_barrett(a[i], b[i], B);
}

// util/src/ring.rs:328-366 (`Rq` +, -, unary -, scalar `*= Zq`) and 272-282 (`AVec<i64>` -> Zq, zq.rs:63-69 `from_i64`):
// element-wise over any modulus q < 2^62; op 0: a + b, 1: a - b, 2: -a, 3: a * scalar, 4: from_i64(a)
FHE_HEADER_KERNEL void rq_elementwise_kernel(const u64 *__restrict__ a, const u64 *__restrict__ b, u64 *__restrict__ out, size_t count, int op,
                                             u64 scalar, Barrett B) {
    const u64 q = B.q;
    for (size_t idx = blockIdx.x * size_t(blockDim.x) + threadIdx.x; idx < count; idx += size_t(gridDim.x) * blockDim.x) {
        const u64 x = a[idx];
        u64 r;
        switch (op) {
            case 0: r = csub(x + b[idx], q); break;
            case 1: { const u64 y = b[idx]; r = x >= y ? x - y : x + q - y; break; }
            case 2: r = x ? q - x : 0; break;
            case 3: r = mulmod_barrett(x, scalar, B); break;
            default: {  // i64 -> [0, q): rem_euclid
                const long long v = (long long)x;
                const u64 m = (v < 0 ? 0ull - (u64)v : (u64)v) % q;
                r = (v < 0 && m) ? q - m : m;
            }
        }
        out[idx] = r;
    }
}

}  // namespace fhe
