// Developer lab: times the N = 2^14 forward/inverse kernels (and ablated builds) with hipEvents.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DFHE_ABLATE_NO_COMPUTE|-DFHE_ABLATE_NO_GLOBAL] -o lab tools/ntt_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../learn-fhe_amd/csrc/modmath.hpp"
#include "../learn-fhe_amd/csrc/ntt_kernels.hpp"
using namespace fhe;
#ifndef LAB_LOG_N
#define LAB_LOG_N 14
#endif
// counter calibration: a plain 8-byte-per-lane coalesced copy of a known byte count (same access width as
// the NTT kernels' global loads/stores)
__global__ void copy8_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) out[i] = in[i] + 1;
}
__global__ void copy16_kernel(const ulonglong2 *__restrict__ in, ulonglong2 *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        ulonglong2 v = in[i]; v.x += 1; out[i] = v;
    }
}

int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == 'c') {  // ./lab calib : 512 MiB in, 512 MiB out, both widths
        size_t n = size_t(64) << 20;
        u64 *a, *b;
        hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
        hipMemset(a, 1, n * 8);
        for (int r = 0; r < 3; ++r) {
            copy8_kernel<<<4096, 256>>>(a, b, n);
            copy16_kernel<<<4096, 256>>>((const ulonglong2 *)a, (ulonglong2 *)b, n / 2);
        }
        hipDeviceSynchronize();
        printf("calib: copy8_kernel and copy16_kernel each read %zu and write %zu bytes per launch\n", n * 8, n * 8);
        return 0;
    }
    const u64 q = 1152921504606748673ull;
    const int log_n = LAB_LOG_N, n = 1 << log_n, batch = argc > 1 ? atoi(argv[1]) : 4096;
    const int s = __builtin_ctzll(q - 1);
    u64 g = smallest_nonresidue(q), omega = powmod(g, (q - 1) >> s, q);
    int log_cap = s - 1;
    size_t cap = size_t(1) << log_cap;
    u64 psi_inv = invmod(omega, q);
    std::vector<TwPair> tw(cap), twi(cap);
    std::vector<u64> pw(cap), pwi(cap);
    u64 x = 1, y = 1;
    for (size_t i = 0; i < cap; ++i) { pw[i] = x; pwi[i] = y; x = mulmod(x, omega, q); y = mulmod(y, psi_inv, q); }
    for (size_t j = 0; j < cap; ++j) {
        size_t r = bitrev((unsigned)j, log_cap);
        tw[j] = {pw[r], shoup(pw[r], q)}; twi[j] = {pwi[r], shoup(pwi[r], q)};
    }
    TwPair *d_tw, *d_twi; u64 *d;
    hipMalloc(&d_tw, cap * sizeof(TwPair)); hipMalloc(&d_twi, cap * sizeof(TwPair));
    hipMemcpy(d_tw, tw.data(), cap * sizeof(TwPair), hipMemcpyHostToDevice);
    hipMemcpy(d_twi, twi.data(), cap * sizeof(TwPair), hipMemcpyHostToDevice);
    std::vector<u64> h(size_t(n) * batch);
    u64 st = 88172645463325252ull;
    for (auto &v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = st % q; }
    hipMalloc(&d, h.size() * 8);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    using C = NttCfg<LAB_LOG_N, 4, 1>;
    auto kf = ntt_fwd_kernel<LAB_LOG_N, 4, 1>;
    auto ki = ntt_inv_kernel<LAB_LOG_N, 4, 1>;
    hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    hipFuncSetAttribute((const void *)ki, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    u64 ninv = invmod((u64)n % q, q), ninv_s = shoup(ninv, q);
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    float best_f = 1e9, best_i = 1e9, sum_f = 0, sum_i = 0;
    const int reps = 12;
    for (int r = 0; r < reps + 2; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kf, dim3(batch), dim3(C::THREADS), C::LDS_BYTES, 0, d, d_tw, q, (unsigned)batch);
        hipEventRecord(e1);
        hipLaunchKernelGGL(ki, dim3(batch), dim3(C::THREADS), C::LDS_BYTES, 0, d, d_twi, q, (unsigned)batch, ninv, ninv_s);
        hipEventRecord(e2);
        hipEventSynchronize(e2);
        float f, i;
        hipEventElapsedTime(&f, e0, e1); hipEventElapsedTime(&i, e1, e2);
        if (r >= 2) { best_f = f < best_f ? f : best_f; best_i = i < best_i ? i : best_i; sum_f += f; sum_i += i; }
    }
    double bytes = 16.0 * n * batch;
    printf("N=2^%d batch=%d  fwd avg %.4f ms (min %.4f) %.0f GB/s | inv avg %.4f ms (min %.4f) %.0f GB/s\n", log_n, batch,
           sum_f / reps, best_f, bytes / (sum_f / reps * 1e-3) / 1e9, sum_i / reps, best_i, bytes / (sum_i / reps * 1e-3) / 1e9);
#if !defined(FHE_ABLATE_NO_COMPUTE) && !defined(FHE_ABLATE_NO_GLOBAL)
    std::vector<u64> out(h.size());
    hipMemcpy(out.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < h.size(); ++i) bad += out[i] != h[i];
    printf("round-trip mismatches: %zu\n", bad);
#endif
    return 0;
}
