// Integer-multiply issue-rate microbenchmark for gfx950: how many 32x32 multiplies per clock does a SIMD
// retire, and what does one Shoup butterfly cost?  Build: hipcc -O3 --offload-arch=gfx950 -o mb tools/microbench_intmul.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef unsigned int u32;

#define ITERS 4096
#define CHAINS 8

// pseudo-Mersenne q = 2^B - c: w*y mod q (lazy) with 7 full 32x32->64 multiply-adds and no companion table
template <int B>
__device__ __forceinline__ u64 pm_mul(u64 y, u64 w, u32 c) {
    const u32 y0 = (u32)y, y1 = (u32)(y >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
    const u64 m0 = (u64)w0 * y0;
    const u64 m1 = (u64)w0 * y1 + (m0 >> 32);
    const u64 m2 = (u64)w1 * y0 + (u32)m1;
    const u64 m3 = (u64)w1 * y1 + ((m1 >> 32) + (m2 >> 32));
    const u64 plo = ((u64)(u32)m2 << 32) | (u32)m0;
    const u64 H = (m3 << (64 - B)) | (plo >> B);
    const u64 L = plo & ((1ull << B) - 1);
    const u64 s0 = (u64)(u32)H * c + L;
    const u64 s1 = (u64)(u32)(H >> 32) * c + (s0 >> 32);
    const u32 H2 = (u32)(s1 >> (B - 32));
    const u64 L2 = ((s1 & ((1ull << (B - 32)) - 1)) << 32) | (u32)s0;
    return (u64)H2 * c + L2;
}

template <int OP>
__global__ void kern(u64 *out, u64 seed, u64 q) {
    u64 x[CHAINS];
    u32 y[CHAINS];
    double f[CHAINS];
    const u64 t = threadIdx.x + blockIdx.x * blockDim.x;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) { x[c] = seed * (t + c + 1) | 1; y[c] = (u32)(x[c] >> 7) | 1; f[c] = (double)(y[c] & 1023) + 0.5; }
    const u64 w = seed | 3, ws = ~seed;
    const u32 m = (u32)seed | 5;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) y[c] = y[c] * m + 1;                                   // v_mul_lo_u32 (+add)
            if (OP == 1) y[c] = __umulhi(y[c], m) + y[c];                       // v_mul_hi_u32
            if (OP == 2) x[c] = (u64)(u32)x[c] * (u64)m + x[c];                 // v_mad_u64_u32
            if (OP == 3) y[c] = __umul24(y[c], m) + 1;                          // v_mul_u32_u24
            if (OP == 4) x[c] = __umul64hi(x[c], ws) + 1;                       // 64x64 -> hi
            if (OP == 5) x[c] = x[c] * w + 1;                                   // 64x64 -> lo
            if (OP == 6) { u64 hi = __umul64hi(ws, x[c]); x[c] = w * x[c] - hi * q; }  // Shoup lazy mul
            if (OP == 7) f[c] = __builtin_fma(f[c], 1.0000001, 0.5);            // v_fma_f64
            if (OP == 8) x[c] = x[c] + (x[c] >> 3);                              // 64-bit add + shift
            if (OP == 10) x[c] = pm_mul<60>(x[c], w & ((1ull << 60) - 1), 98303u);
            if (OP == 11) {
                if ((c & 1) == 0) {
                    u64 X = x[c], Y = x[c + 1];
                    u64 tt = pm_mul<60>(Y, w & ((1ull << 60) - 1), 98303u);
                    x[c] = X + tt; x[c + 1] = X - tt + 2 * q;
                }
            }
            if (OP == 12) x[c] = (x[c] & ((1ull << 60) - 1)) + (u64)(u32)(x[c] >> 60) * 98303u;  // lazy fold
            if (OP == 9) {                                                        // full Harvey CT butterfly (pair c, c^1)
                if ((c & 1) == 0) {
                    u64 X = x[c], Y = x[c + 1], q2 = 2 * q;
                    u64 xx = X >= q2 ? X - q2 : X;
                    u64 hi = __umul64hi(ws, Y); u64 tt = w * Y - hi * q;
                    x[c] = xx + tt; x[c + 1] = xx - tt + q2;
                }
            }
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc += x[c] + y[c] + (u64)f[c];
    out[t] = acc;
}

template <int OP>
void run(const char *name, double ops_per_iter_chain) {
    const int blocks = 256 * 8, threads = 256;
    u64 *out;
    hipMalloc(&out, sizeof(u64) * blocks * threads);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    kern<OP><<<blocks, threads>>>(out, 0x9E3779B97F4A7C15ull, 1152921504606748673ull);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) kern<OP><<<blocks, threads>>>(out, 0x9E3779B97F4A7C15ull + r, 1152921504606748673ull);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double total = 5.0 * blocks * threads * (double)ITERS * CHAINS * ops_per_iter_chain;
    double gops = total / (ms * 1e-3) / 1e9;
    // lanes per clock per SIMD at 2.4 GHz, 1024 SIMDs
    printf("%-34s %9.1f Gop/s  -> %6.2f lane-ops/clk/SIMD @2.4GHz (%.3f ms)\n", name, gops, gops * 1e9 / (2.4e9 * 1024), ms / 5);
    hipFree(out);
}

int main() {
    run<0>("v_mul_lo_u32 (+v_add)", 1);
    run<1>("v_mul_hi_u32 (+v_add)", 1);
    run<2>("v_mad_u64_u32", 1);
    run<3>("v_mul_u32_u24 (+v_add)", 1);
    run<4>("__umul64hi (+add64)", 1);
    run<5>("mul64 lo (+add64)", 1);
    run<6>("Shoup lazy modmul", 1);
    run<7>("v_fma_f64", 1);
    run<8>("add64 + shr64", 1);
    run<9>("Harvey CT butterfly", 0.5);
    run<10>("PM (2^60-c) lazy modmul", 1);
    run<11>("PM CT butterfly", 0.5);
    run<12>("PM lazy fold", 1);
    return 0;
}
