// LWE key switch (scheme/fhew/src/lwe.rs:151-160, scheme/tfhe/src/tlwe.rs:144-153) as a tiled matrix-vector product:
// out[ct][col] = sum_rows digit[ct][row] * ksk[row][col], rows = d * n_in digit-major (`decompose(a).flatten()`), column n_out
// carries b.  One block = TILE ciphertexts: their digits are computed once into LDS ([row][TILE] 32-bit words), then every
// thread owns output columns and streams the key rows (coalesced across columns) against LDS-broadcast digits, so a key
// element fetched from L2 serves TILE ciphertexts and no digit is recomputed per column (the one-thread-per-output kernels
// this replaces did both: 1.96 ms per 1024 cfg3 ciphertexts, 8.5 % of a gate bootstrap).
#pragma once
#include "fhew_kernels.hpp"
#include "torus_kernels.hpp"

namespace fhe {

// digit policies: how a coefficient becomes d digits, how a digit word widens, how terms accumulate
struct KsZqPow2 {  // Z_q, q a power of two < 2^32: sums wrap mod 2^64, one mask at the end
    DecompParams P;
    __device__ __forceinline__ int d() const { return P.d; }
    __device__ __forceinline__ u64 init(u64 v) const { return decomp_init(v, P); }
    __device__ __forceinline__ unsigned next(u64 &c) const { return (unsigned)decomp_next(c, P); }
    __device__ __forceinline__ u64 term(u64 acc, u64 k, unsigned dg) const { return acc + k * dg; }
    __device__ __forceinline__ u64 finish(u64 acc) const { return acc & (P.q - 1); }
};
struct KsZq {  // any q < 2^32: products < 2^64, partial sums < rows * q
    DecompParams P;
    __device__ __forceinline__ int d() const { return P.d; }
    __device__ __forceinline__ u64 init(u64 v) const { return decomp_init(v, P); }
    __device__ __forceinline__ unsigned next(u64 &c) const { return (unsigned)decomp_next(c, P); }
    __device__ __forceinline__ u64 term(u64 acc, u64 k, unsigned dg) const { return acc + k * dg % P.q; }
    __device__ __forceinline__ u64 finish(u64 acc) const { return acc % P.q; }
};
struct KsTorus {  // T64: signed digits (|digit| <= 2^(log_b-1), log_b <= 31), arithmetic mod 2^64
    TDecomp P;
    __device__ __forceinline__ int d() const { return P.d; }
    __device__ __forceinline__ u64 init(u64 v) const { return tdecomp_init(v, P); }
    __device__ __forceinline__ unsigned next(u64 &c) const { return (unsigned)tdecomp_next(c, P); }
    __device__ __forceinline__ u64 term(u64 acc, u64 k, unsigned dg) const { return acc + k * (u64)(long long)(int)dg; }
    __device__ __forceinline__ u64 finish(u64 acc) const { return acc; }
};

constexpr int KS_THREADS = 128;

template <class DEC, int TILE>
__global__ __launch_bounds__(KS_THREADS) void lwe_key_switch_tiled(const u64 *__restrict__ ct_a, const u64 *__restrict__ ct_b, unsigned n_in,
                                                                   unsigned n_out, unsigned batch, const u64 *__restrict__ ksk_a,
                                                                   const u64 *__restrict__ ksk_b, DEC dec, u64 *__restrict__ out_a,
                                                                   u64 *__restrict__ out_b) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned *dig = reinterpret_cast<unsigned *>(smem_raw);  // [rows][TILE]
    const unsigned ct0 = blockIdx.x * TILE;
    const unsigned rows = n_in * dec.d();
    for (unsigned idx = threadIdx.x; idx < n_in * TILE; idx += KS_THREADS) {
        const unsigned c = idx / n_in, i = idx - c * n_in;
        const bool live = ct0 + c < batch;
        u64 st = live ? dec.init(ct_a[size_t(ct0 + c) * n_in + i]) : 0;
        for (int j = 0; j < dec.d(); ++j) dig[(size_t(j) * n_in + i) * TILE + c] = live ? dec.next(st) : 0u;
    }
    __syncthreads();
    for (unsigned col = threadIdx.x; col <= n_out; col += KS_THREADS) {
        u64 acc[TILE];
#pragma unroll
        for (int c = 0; c < TILE; ++c) acc[c] = 0;
        const u64 *kp = col < n_out ? ksk_a + col : ksk_b;
        const size_t stride = col < n_out ? n_out : 1;
#pragma unroll 4
        for (unsigned row = 0; row < rows; ++row) {
            const u64 kv = kp[row * stride];
#pragma unroll
            for (int c = 0; c < TILE; ++c) acc[c] = dec.term(acc[c], kv, dig[row * TILE + c]);
        }
#pragma unroll
        for (int c = 0; c < TILE; ++c) {
            if (ct0 + c >= batch) continue;
            if (col < n_out) out_a[size_t(ct0 + c) * n_out + col] = dec.finish(acc[c]);
            else out_b[ct0 + c] = dec.finish(acc[c] + ct_b[ct0 + c]);
        }
    }
}

// Wide-output variant (TFHE: 630 output columns, 5120 rows; torus digits with log_b <= 7 lie in [-64, 64]: bytes).  The grid splits the
// ciphertexts (TILE per block: a key element fetched from L2 serves all of them), the output columns (KS_THREADS per block) AND the input
// coefficients (`chunk` per block, all d digits of each); a block adds its partial sums to the outputs with 64-bit atomics -- additions
// mod 2^64 commute, so the result is the same bits in any order.  Round 2's form walked all 5120 rows in one block: 640 blocks of two
// waves at batch 1024 (2.5 blocks per CU, four key loads in flight per wave: 64 % of its cycles parked, 1.86 ms per 1024 ciphertexts --
// 15 % of a gate in the fft64 mode); here the grid is Z times larger and the digit image Z times smaller.  Digits are stored with +128 (unsigned bytes), so
// a term is ONE v_mad_u64_u32 per output dword; the offset leaves as 128 x (the sum of the key column), taken once per block.
// out_a, out_b must be ZERO before the launch.
template <int TILE>
__global__ __launch_bounds__(KS_THREADS) void tlwe_key_switch_split(const u64 *__restrict__ ct_a, const u64 *__restrict__ ct_b, unsigned n_in,
                                                                    unsigned n_out, unsigned batch, const u64 *__restrict__ ksk_a,
                                                                    const u64 *__restrict__ ksk_b, TDecomp P, u64 *__restrict__ out_a,
                                                                    u64 *__restrict__ out_b, unsigned chunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char *dig = smem_raw;  // [d][chunk][TILE]
    const unsigned ct0 = blockIdx.x * TILE, i0 = blockIdx.z * chunk;
    const unsigned ni = min(chunk, n_in - i0);
    for (unsigned idx = threadIdx.x; idx < ni * TILE; idx += KS_THREADS) {
        const unsigned c = idx / ni, ii = idx - c * ni;
        const bool live = ct0 + c < batch;
        u64 st = live ? tdecomp_init(ct_a[size_t(ct0 + c) * n_in + i0 + ii], P) : 0;
        for (int j = 0; j < P.d; ++j) dig[(size_t(j) * chunk + ii) * TILE + c] = (unsigned char)(128 + (live ? (int)(long long)tdecomp_next(st, P) : 0));
    }
    __syncthreads();
    const unsigned col = blockIdx.y * KS_THREADS + threadIdx.x;
    if (col > n_out) return;
    unsigned lo[TILE], hi[TILE];  // the low and the high dword of each sum, carried apart
#pragma unroll
    for (int c = 0; c < TILE; ++c) lo[c] = hi[c] = 0;
    u64 ksum = 0;
    const u64 *kp = col < n_out ? ksk_a + col : ksk_b;
    const size_t stride = col < n_out ? n_out : 1;
    for (int j = 0; j < P.d; ++j) {
        const u64 *kr = kp + (size_t(j) * n_in + i0) * stride;
        const unsigned char *dj = dig + size_t(j) * chunk * TILE;
#pragma unroll 4
        for (unsigned ii = 0; ii < ni; ++ii) {
            const u64 kv = kr[ii * stride];
            ksum += kv;
            const unsigned klo = (unsigned)kv, khi = (unsigned)(kv >> 32);
            const unsigned char *dr = dj + size_t(ii) * TILE;
#pragma unroll
            for (int c = 0; c < TILE; ++c) {
                const unsigned dg = dr[c];
                const u64 t = (u64)klo * dg + lo[c];  // < 2^40 + 2^32: the carry into the high dword rides in t's upper half
                lo[c] = (unsigned)t;
                hi[c] += khi * dg + (unsigned)(t >> 32);
            }
        }
    }
    const u64 off = ksum << 7;  // 128 x sum of the key entries
#pragma unroll
    for (int c = 0; c < TILE; ++c) {
        if (ct0 + c >= batch) continue;
        u64 v = (((u64)hi[c] << 32) | lo[c]) - off;
        if (col < n_out) atomicAdd(reinterpret_cast<unsigned long long *>(out_a + size_t(ct0 + c) * n_out + col), (unsigned long long)v);
        else atomicAdd(reinterpret_cast<unsigned long long *>(out_b + ct0 + c), (unsigned long long)(v + (blockIdx.z == 0 ? ct_b[ct0 + c] : 0)));
    }
}

// host: ciphertexts per block for this batch and row count (LDS budget 128 KiB); 0 = digits of one ciphertext do not fit
inline int ks_tile(size_t batch, size_t rows) {
    int tile = batch >= 2048 ? 4 : (batch >= 512 ? 2 : 1);
    while (tile > 1 && rows * tile * 4 > 128 * 1024) tile >>= 1;
    return rows * tile * 4 > 128 * 1024 ? 0 : tile;
}

template <class DEC>
int launch_key_switch_tiled(const DEC &dec, const u64 *ct_a, const u64 *ct_b, size_t n_in, size_t n_out, size_t batch, const u64 *ksk_a,
                            const u64 *ksk_b, u64 *out_a, u64 *out_b, int tile, hipStream_t st) {
    const size_t lds = n_in * dec.P.d * tile * 4;
#define KS_LAUNCH(T)                                                                                                            \
    {                                                                                                                           \
        auto k = lwe_key_switch_tiled<DEC, T>;                                                                                  \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return 1;                                                                                                           \
        hipLaunchKernelGGL(k, dim3((unsigned)((batch + T - 1) / T)), dim3(KS_THREADS), lds, st, ct_a, ct_b, (unsigned)n_in,     \
                           (unsigned)n_out, (unsigned)batch, ksk_a, ksk_b, dec, out_a, out_b);                                  \
    }
    if (tile == 4) KS_LAUNCH(4) else if (tile == 2) KS_LAUNCH(2) else KS_LAUNCH(1)
#undef KS_LAUNCH
    return 0;
}

}  // namespace fhe
