// Developer lab: the one dense contraction on the key-switch path -- the 8 x 8 constant matrix of
// `Rns::extend_bases` / `rescale_k` (util/src/ring/rns.rs:331-345) applied to N x batch columns of 60-bit residues,
//     out[j][col] = sum_i M[j][i] vs[i][col]  mod b_j,        8 sixty-bit pseudo-Mersenne moduli (cfg4's ps) --
// in three forms, each checked against unsigned __int128 on the host, each timed over the same resident data:
//   A  term by term: one two-operand modular product per (j, i) (round 2's kernels: 6 multiply-adds + a fold each)
//   B  unreduced 30-bit Karatsuba accumulation, one reduction per output (pm_dot.hpp: what round 3 ships)
//   C  int8 MFMA (v_mfma_i32_16x16x64_i8): a little-endian u64 IS an 8-wide K slice, so K = 8 limbs x 8 bytes = 64 exactly; the matrix
//      becomes a Toeplitz operand of signed byte digits (rows = 16 diagonals of one output), the 15 diagonals of i32 partial sums
//      are recombined by shift-adds and reduced once per output.
// `reps` > 1 repeats the arithmetic on a loaded column (inputs perturbed, results xor-ed) so that the instruction streams, not
// the 2 x 128 MiB of HBM traffic, set the time.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/baseconv_lab tools/baseconv_lab.hip ; run: tools/baseconv_lab [log2 columns]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../learn-fhe_amd/csrc/pm_dot.hpp"

using namespace fhe::pd;
typedef unsigned __int128 u128;
typedef int v4i __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <class T>
__device__ __forceinline__ T ldc(const T *p, int i) { return ((const __attribute__((address_space(4))) T *)p)[i]; }

struct Tab {
    const unsigned *ds;    // [8][8][4]  two-operand form of M[j][i] (form A)
    const unsigned *k3;    // [8][3][8]  k0 | k1 | kk rows (form B)
    const unsigned *cc;    // [8][2]     {c, c60}
    const u64 *b;          // [8]
    const v4i *atile;      // [8 tiles][64 lanes] the Toeplitz operand (form C)
    const u64 *lane_mod;   // [64][2][2]: per lane (c, g): {b, c} of outputs g and g + 4 (form C)
};

// ---- form A -----------------------------------------------------------------------------------------------------------------
template <int REPS>
__global__ void conv_a(const u64 *__restrict__ in, u64 *__restrict__ out, size_t cols, Tab T, Uni U) {
    for (size_t col = blockIdx.x * size_t(blockDim.x) + threadIdx.x; col < cols; col += size_t(gridDim.x) * blockDim.x) {
        u64 v[8], acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = in[size_t(i) * cols + col];
#pragma unroll 1
        for (int r = 0; r < REPS; ++r) {
#pragma unroll 1
            for (int j = 0; j < 8; ++j) {
                const unsigned c = ldc(T.cc, 2 * j);
                u64 dot = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned *w = T.ds + (j * 8 + i) * 4;
                    dot += fold(ds_mul_raw(v[i] + r, ldc(w, 0), ldc(w, 1), ldc(w, 2), ldc(w, 3), 2 * c, U), c, U);  // each < q + 9c
                }
                const u64 d = csub(fold(dot, c, U), ldc(T.b, j));
                if (REPS == 1) out[size_t(j) * cols + col] = d;
                else acc[j] ^= d;
            }
        }
        if (REPS > 1) {
            u64 x = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) x ^= acc[j];
            out[col] = x;
        }
    }
}

// ---- form B -----------------------------------------------------------------------------------------------------------------
template <int REPS>
__global__ void conv_b(const u64 *__restrict__ in, u64 *__restrict__ out, size_t cols, Tab T, Uni U) {
    for (size_t col = blockIdx.x * size_t(blockDim.x) + threadIdx.x; col < cols; col += size_t(gridDim.x) * blockDim.x) {
        u64 v[8], acc = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = in[size_t(i) * cols + col];
#pragma unroll 1
        for (int r = 0; r < REPS; ++r) {
            unsigned y0[8], y1[8], yk[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { const Y3 y = split30(v[i] + r); y0[i] = y.y0; y1[i] = y.y1; yk[i] = y.yk; }
#pragma unroll 1
            for (int j = 0; j < 8; ++j) {
                const unsigned *k = T.k3 + j * 24;
                u64 s00 = 0, s11 = 0, sk = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    s00 += (u64)ldc(k, i) * y0[i];
                    s11 += (u64)ldc(k, 8 + i) * y1[i];
                    sk += (u64)ldc(k, 16 + i) * yk[i];
                }
                const u64 d = csub(reduce_lazy(s00, sk - s00 - s11, 0, s11, ldc(T.cc, 2 * j), ldc(T.cc, 2 * j + 1), U), ldc(T.b, j));
                if (REPS == 1) out[size_t(j) * cols + col] = d;
                else acc ^= d;
            }
        }
        if (REPS > 1) out[col] = acc;
    }
}

// ---- form C -----------------------------------------------------------------------------------------------------------------
// One wave = 64 columns as four tiles of 16.  Lane (c = l & 15, g = l >> 4) of a tile holds, as the B operand, the 16 bytes
// {vs[2g][col], vs[2g+1][col]} in signed-digit form, (v + 0x80..80) ^ 0x80..80 (the digits d_b = byte_b(v + C) - 128 sum back to v);
// the A operand of tile t is the constant Toeplitz block whose row 4 g' + r is diagonal 4 (t & 3) + r of output g' + 4 (t >> 2), so
// after the eight MFMAs lane (c, g) owns ALL 16 diagonals of outputs g and g + 4 of its column.
__device__ __forceinline__ u64 recombine(const v4i &d0, const v4i &d1, const v4i &d2, const v4i &d3, u64 b, unsigned c) {
    // word w: t_w = D[4w] + D[4w+1] 2^8 + D[4w+2] 2^16 + D[4w+3] 2^24  (|t_w| < 2^46)
    auto word = [](const v4i &d) { return (long long)d.x + ((long long)d.y << 8) + ((long long)d.z << 16) + ((long long)d.w << 24); };
    const long long s0 = word(d0), s1 = word(d1) + (s0 >> 32), s2 = word(d2) + (s1 >> 32), s3 = word(d3) + (s2 >> 32);
    const u64 A = (u64)(unsigned)s0 | ((u64)(unsigned)s1 << 32), B = (u64)(unsigned)s2 | ((u64)(unsigned)s3 << 32);  // X = A + B 2^64, B < 2^60
    // 2^64 = 16 c (mod b): X = A + lo32(B) 16c + hi32(B) 16c 2^32, g = hi32(B) 16c < 2^59, g 2^32 = (g mod 2^28) 2^32 + (g >> 28) c
    const unsigned c16 = c << 4;
    const u64 a0 = (A & ((u64(1) << 60) - 1)) + (A >> 60) * c;
    const u64 g = (u64)(unsigned)(B >> 32) * c16;
    u64 r = (u64)(unsigned)B * c16 + a0;
    r += (u64)((unsigned)g & 0x0fffffffu) << 32;
    r += (u64)(unsigned)(g >> 28) * c;
    r = (r & ((u64(1) << 60) - 1)) + (r >> 60) * c;
    return csub(r, b);
}
template <int REPS>
__global__ void __launch_bounds__(256) conv_c(const u64 *__restrict__ in, u64 *__restrict__ out, size_t cols, Tab T) {
    const int lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4;
    v4i At[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) At[t] = T.atile[t * 64 + lane];
    const u64 bA = T.lane_mod[lane * 4 + 0], bB = T.lane_mod[lane * 4 + 2];
    const unsigned cA = (unsigned)T.lane_mod[lane * 4 + 1], cB = (unsigned)T.lane_mod[lane * 4 + 3];
    const size_t waves = (size_t(gridDim.x) * blockDim.x) >> 6;
    for (size_t w = (blockIdx.x * size_t(blockDim.x) + threadIdx.x) >> 6; w * 64 < cols; w += waves) {
        u64 accx = 0;
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            const size_t col = w * 64 + t * 16 + c16;
            const u64 v0 = in[size_t(2 * g) * cols + col], v1 = in[size_t(2 * g + 1) * cols + col];
#pragma unroll 1
            for (int r = 0; r < REPS; ++r) {
                const u64 C8 = 0x8080808080808080ull;
                const u64 e0 = ((v0 + r) + C8) ^ C8, e1 = ((v1 + r) + C8) ^ C8;
                const v4i Bop = {(int)(unsigned)e0, (int)(unsigned)(e0 >> 32), (int)(unsigned)e1, (int)(unsigned)(e1 >> 32)};
                const v4i z = {0, 0, 0, 0};
                v4i D[8];
#pragma unroll
                for (int tt = 0; tt < 8; ++tt) D[tt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(At[tt], Bop, z, 0, 0, 0);
                const u64 oA = recombine(D[0], D[1], D[2], D[3], bA, cA), oB = recombine(D[4], D[5], D[6], D[7], bB, cB);
                if (REPS == 1) {
                    out[size_t(g) * cols + col] = oA;
                    out[size_t(g + 4) * cols + col] = oB;
                } else {
                    accx ^= oA ^ oB;
                }
            }
        }
        if (REPS > 1) out[w * 64 + lane] = accx;
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------------
static u64 rng_state = 88172645463325252ull;
static u64 rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

template <class K>
static float time_kernel(K launch, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char **argv) {
    const int log_cols = argc > 1 ? atoi(argv[1]) : 21;  // 2^21 = 64 ciphertexts x 2^15 coefficients
    const size_t cols = size_t(1) << log_cols;
    // cfg4's moduli (two_adic_primes(60, 16): SURVEY.md section 8(a) row a2): source base qs, target base ps
    const u64 pre = 1152921504000000000ull;
    const u64 qs[8] = {pre + 606584833, pre + 598720513, pre + 597016577, pre + 595968001, pre + 595640321, pre + 593412097, pre + 592822273, pre + 592429057};
    const u64 ps[8] = {pre + 589938689, pre + 586530817, pre + 585547777, pre + 583647233, pre + 581877761, pre + 581419009, pre + 580894721, pre + 578666497};
    const Uni U = make_uni(60);
    // M[j][i] = (Q / q_i) mod p_j (rns.rs:305-313)
    u64 M[8][8];
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 8; ++i) {
            u64 r = 1;
            for (int k = 0; k < 8; ++k)
                if (k != i) r = (u64)((u128)r * (qs[k] % ps[j]) % ps[j]);
            M[j][i] = r;
        }
    std::vector<unsigned> ds(8 * 8 * 4), k3(8 * 24), cc(16);
    std::vector<v4i> atile(8 * 64);
    std::vector<u64> lane_mod(64 * 4);
    for (int j = 0; j < 8; ++j) {
        const u64 b = ps[j], c = (u64(1) << 60) - b;
        cc[2 * j] = (unsigned)c; cc[2 * j + 1] = (unsigned)c;
        for (int i = 0; i < 8; ++i) {
            const u64 w = M[j][i], w1 = (u64)((((u128)w) << 32) % b), lo = (u64(1) << 29) - 1;
            unsigned *d = &ds[(j * 8 + i) * 4];
            d[0] = (unsigned)(w & lo); d[1] = (unsigned)(w >> 29); d[2] = (unsigned)(w1 & lo); d[3] = (unsigned)(w1 >> 29);
            const Y3 k = split30(w);
            k3[j * 24 + i] = k.y0; k3[j * 24 + 8 + i] = k.y1; k3[j * 24 + 16 + i] = k.yk;
        }
    }
    // signed byte digits of M: d_a = byte_a(M + C) - 128
    auto digit = [&](int j, int i, int a) -> int {
        if (a < 0 || a > 7) return 0;
        const u64 e = (M[j][i] + 0x8080808080808080ull) ^ 0x8080808080808080ull;
        return (int)(signed char)(unsigned char)(e >> (8 * a));
    };
    for (int t = 0; t < 8; ++t)
        for (int l = 0; l < 64; ++l) {
            const int row = l & 15, kg = l >> 4, j = (row >> 2) + 4 * (t >> 2), s = 4 * (t & 3) + (row & 3);
            unsigned char bytes[16];
            for (int jj = 0; jj < 16; ++jj) {
                const int i = 2 * kg + (jj >> 3), bb = jj & 7;
                bytes[jj] = (unsigned char)(signed char)digit(j, i, s - bb);
            }
            std::memcpy(&atile[t * 64 + l], bytes, 16);
        }
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4;
        lane_mod[l * 4 + 0] = ps[g]; lane_mod[l * 4 + 1] = (u64(1) << 60) - ps[g];
        lane_mod[l * 4 + 2] = ps[g + 4]; lane_mod[l * 4 + 3] = (u64(1) << 60) - ps[g + 4];
    }
    std::vector<u64> hin(8 * cols);
    for (int i = 0; i < 8; ++i)
        for (size_t c = 0; c < cols; ++c) hin[size_t(i) * cols + c] = c < 8 ? qs[i] - 1 - c : rnd() % (qs[i] - 64);  // extremes first
    u64 *din, *dout;
    unsigned *d_ds, *d_k3, *d_cc;
    u64 *d_b, *d_lm;
    v4i *d_at;
    CK(hipMalloc(&din, 8 * cols * 8)); CK(hipMalloc(&dout, 8 * cols * 8));
    CK(hipMalloc(&d_ds, ds.size() * 4)); CK(hipMalloc(&d_k3, k3.size() * 4)); CK(hipMalloc(&d_cc, cc.size() * 4));
    CK(hipMalloc(&d_b, 64)); CK(hipMalloc(&d_lm, lane_mod.size() * 8)); CK(hipMalloc(&d_at, atile.size() * 16));
    CK(hipMemcpy(din, hin.data(), 8 * cols * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ds, ds.data(), ds.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_k3, k3.data(), k3.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cc, cc.data(), cc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_b, ps, 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_lm, lane_mod.data(), lane_mod.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_at, atile.data(), atile.size() * 16, hipMemcpyHostToDevice));
    const Tab T{d_ds, d_k3, d_cc, d_b, d_at, d_lm};
    const unsigned grid = (unsigned)((cols + 255) / 256 > 16384 ? 16384 : (cols + 255) / 256);

    // ---- correctness: every form against unsigned __int128 on the first and last 4096 columns
    std::vector<u64> hout(8 * cols);
    auto check = [&](const char *name) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hout.data(), dout, 8 * cols * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t c = 0; c < cols; c = (c == 4095 && cols > 8192) ? cols - 4096 : c + 1)
            for (int j = 0; j < 8; ++j) {
                u128 acc = 0;
                for (int i = 0; i < 8; ++i) acc += (u128)M[j][i] * hin[size_t(i) * cols + c] % ps[j];
                if ((u64)(acc % ps[j]) != hout[size_t(j) * cols + c]) ++bad;
            }
        printf("form %s: %s (%zu mismatches)\n", name, bad ? "WRONG" : "bit-exact", bad);
        CK(hipMemset(dout, 0, 8 * cols * 8));
        return bad == 0;
    };
    bool ok = true;
    hipLaunchKernelGGL(conv_a<1>, dim3(grid), dim3(256), 0, 0, din, dout, cols, T, U); ok &= check("A (term by term)");
    hipLaunchKernelGGL(conv_b<1>, dim3(grid), dim3(256), 0, 0, din, dout, cols, T, U); ok &= check("B (unreduced Karatsuba)");
    hipLaunchKernelGGL(conv_c<1>, dim3(grid), dim3(256), 0, 0, din, dout, cols, T); ok &= check("C (int8 MFMA Toeplitz)");

    // ---- time: with the HBM traffic (reps 1) and compute bound (reps 8)
    const double outs = 8.0 * cols;
    auto report = [&](const char *name, float ms1, float ms8) {
        printf("%-28s reps=1 %8.1f us (%5.2f TB/s of 16 B per output)   reps=8 %8.1f us = %6.1f us per pass, %5.2f G outputs/s\n", name, ms1 * 1e3,
               outs * 16 / (ms1 * 1e-3) / 1e12, ms8 * 1e3, ms8 * 1e3 / 8, outs * 8 / (ms8 * 1e-3) / 1e9);
    };
    float a1 = time_kernel([&] { hipLaunchKernelGGL(conv_a<1>, dim3(grid), dim3(256), 0, 0, din, dout, cols, T, U); }, 20);
    float a8 = time_kernel([&] { hipLaunchKernelGGL(conv_a<8>, dim3(grid), dim3(256), 0, 0, din, dout, cols, T, U); }, 10);
    report("A term by term", a1, a8);
    float b1 = time_kernel([&] { hipLaunchKernelGGL(conv_b<1>, dim3(grid), dim3(256), 0, 0, din, dout, cols, T, U); }, 20);
    float b8 = time_kernel([&] { hipLaunchKernelGGL(conv_b<8>, dim3(grid), dim3(256), 0, 0, din, dout, cols, T, U); }, 10);
    report("B unreduced Karatsuba", b1, b8);
    float c1 = time_kernel([&] { hipLaunchKernelGGL(conv_c<1>, dim3(grid), dim3(256), 0, 0, din, dout, cols, T); }, 20);
    float c8 = time_kernel([&] { hipLaunchKernelGGL(conv_c<8>, dim3(grid), dim3(256), 0, 0, din, dout, cols, T); }, 10);
    report("C int8 MFMA Toeplitz", c1, c8);
    CK(hipDeviceSynchronize());
    return ok ? 0 : 1;
}
