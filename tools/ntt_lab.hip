// Developer lab: times the N = 2^14 forward/inverse kernels (and ablated builds) with hipEvents.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DFHE_ABLATE_NO_COMPUTE|-DFHE_ABLATE_NO_GLOBAL] -o lab tools/ntt_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../learn-fhe_amd/csrc/modmath.hpp"
#include "../learn-fhe_amd/csrc/ntt_kernels.hpp"
#include "../learn-fhe_amd/csrc/ntt14.hpp"
using namespace fhe;
// ablation policies for the ntt14 structure
struct ArithNone : ArithPM<60> {  // exchanges + HBM traffic only
    static __device__ __forceinline__ void ct(u64 &X, u64 &Y, const TwReg &, const K &) { X ^= 1; Y ^= 1; }
    template <int PH> static __device__ __forceinline__ void gs(u64 &X, u64 &Y, const TwReg &, const K &) { X ^= 1; Y ^= 1; }
    template <bool INV> static __device__ __forceinline__ TwRaw fetch(const K &, int idx) { return TwRaw{(unsigned)idx, 1u}; }
    static constexpr bool GS_FOLDS = false;
    static __device__ __forceinline__ u64 fold(u64 x, const K &) { return x; }
    static __device__ __forceinline__ u64 canon_fwd(u64 x, const K &) { return x; }
    static __device__ __forceinline__ u64 finish_inv(u64 x, const K &) { return x; }
};
struct ArithNoTw : ArithPM<60> {  // butterflies with a computed twiddle: no twiddle loads
    template <bool INV> static __device__ __forceinline__ TwRaw fetch(const K &k, int idx) { return TwRaw{k.ninv.wl + (unsigned)idx, k.ninv.wh}; }
};
struct ArithPrefetch : ArithPM<60> {  // the same arithmetic with every twiddle set fetched one unit ahead
    static constexpr int PREFETCH = 8;
};
struct ArithPrefetch4 : ArithPM<60> {  // only sets of <= 4 twiddles ahead
    static constexpr int PREFETCH = 4;
};
struct ArithPrefetch2 : ArithPM<60> {
    static constexpr int PREFETCH = 2;
};
// experiment: persistent workgroups (2 per CU) looping over polynomials, the second workgroup of a CU starting late
template <class A>
__global__ __launch_bounds__(N14_THREADS, 4) void ntt14_fwd_persistent(u64 *__restrict__ data, const ModDesc *__restrict__ descs,
                                                                        unsigned n_desc, unsigned subs, int delay) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const typename A::K k = A::make(descs[0], 14, 0, 0);
    if (blockIdx.x >= gridDim.x / 2)
        for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(127);
    for (unsigned sub = blockIdx.x; sub < subs; sub += gridDim.x) {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        ntt14_fwd_body<A>(data + (size_t(sub) << 14), k, lds, t);
    }
}
// experiment: PER polynomials per workgroup, back to back (fewer workgroup launches, still dynamically dispatched)
template <class A, int PER>
__global__ __launch_bounds__(N14_THREADS, 4) void ntt14_fwd_multi(u64 *__restrict__ data, const ModDesc *__restrict__ descs,
                                                                   unsigned n_desc, unsigned subs, int) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const typename A::K k = A::make(descs[0], 14, 0, 0);
#pragma unroll 1
    for (int r = 0; r < PER; ++r) {
        const unsigned sub = blockIdx.x * PER + r;
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));  // opaque per iteration: keeps the twiddle address arithmetic inside the loop body (no LICM -> no spills)
        if (sub < subs) ntt14_fwd_body<A>(data + (size_t(sub) << 14), k, lds, t);
    }
}
// experiment: the ordinary one-polynomial-per-workgroup kernel, second resident generation (blocks 256..511) starts late
template <class A>
__global__ __launch_bounds__(N14_THREADS, 4) void ntt14_fwd_dephased(u64 *__restrict__ data, const ModDesc *__restrict__ descs,
                                                                      unsigned n_desc, unsigned subs, int delay) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const typename A::K k = A::make(descs[0], 14, 0, 0);
    if (blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(127);
    ntt14_fwd_body<A>(data + (size_t(blockIdx.x) << 14), k, lds, threadIdx.x);
}
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
    ModDesc hd{};
    hd.q = q; hd.tw = d_tw; hd.twi = d_twi; hd.one_s = shoup(1, q);
    for (int k = 0; k < 20; ++k) { hd.ninv[k] = invmod((u64(1) << k) % q, q); hd.ninv_s[k] = shoup(hd.ninv[k], q);
                                  hd.ninv_w[k] = mulmod(hd.ninv[k], twi[1].w, q); hd.ninv_w_s[k] = shoup(hd.ninv_w[k], q); }
    {   // pseudo-Mersenne tables (q = 2^60 - 98303)
        std::vector<u64> w(cap), wi(cap);
        for (size_t j = 0; j < cap; ++j) { w[j] = ArithPM<60>::pack(tw[j].w); wi[j] = ArithPM<60>::pack(twi[j].w); }
        u64 *dw, *dwi;
        hipMalloc(&dw, cap * 8); hipMalloc(&dwi, cap * 8);
        hipMemcpy(dw, w.data(), cap * 8, hipMemcpyHostToDevice);
        hipMemcpy(dwi, wi.data(), cap * 8, hipMemcpyHostToDevice);
        hd.tww = (const u64 *)dw; hd.twwi = (const u64 *)dwi; hd.pm_b = 60; hd.pm_c = (unsigned)((u64(1) << 60) - q);
    }
    ModDesc *d_desc;
    hipMalloc(&d_desc, sizeof(ModDesc));
    hipMemcpy(d_desc, &hd, sizeof(ModDesc), hipMemcpyHostToDevice);
    std::vector<u64> h(size_t(n) * batch);
    u64 st = 88172645463325252ull;
    for (auto &v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = st % q; }
    hipMalloc(&d, h.size() * 8);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    using C = NttCfg<LAB_LOG_N, 4, 1>;
    typedef void (*kern_t)(u64 *, const ModDesc *, unsigned, unsigned, int);
    struct Variant { const char *name; kern_t f, i; };
    struct V14 { const char *name; kern_t f, i; };
    V14 v14[] = {
        {"ntt14 Shoup (512 thr, 2 WG/CU)", ntt14_fwd_kernel<ArithShoup, false>, ntt14_inv_kernel<ArithShoup, false>},
        {"ntt14 pseudo-Mersenne", ntt14_fwd_kernel<ArithPM<60>, false>, ntt14_inv_kernel<ArithPM<60>, false>},
        {"ntt14 ablation: no butterflies", ntt14_fwd_kernel<ArithNone, false>, ntt14_inv_kernel<ArithNone, false>},
        {"ntt14 ablation: no twiddle loads", ntt14_fwd_kernel<ArithNoTw, false>, ntt14_inv_kernel<ArithNoTw, false>},
        {"ntt14 pseudo-Mersenne, twiddle prefetch", ntt14_fwd_kernel<ArithPrefetch, false>, ntt14_inv_kernel<ArithPrefetch, false>},
        {"ntt14 pseudo-Mersenne, prefetch <= 4", ntt14_fwd_kernel<ArithPrefetch4, false>, ntt14_inv_kernel<ArithPrefetch4, false>},
        {"ntt14 pseudo-Mersenne, prefetch <= 2", ntt14_fwd_kernel<ArithPrefetch2, false>, ntt14_inv_kernel<ArithPrefetch2, false>},
    };
    Variant vars[] = {
        {"generic Shoup, fwd staged / inv direct", ntt_fwd_kernel<ArithShoup, LAB_LOG_N, 4, 1, false, false>, ntt_inv_kernel<ArithShoup, LAB_LOG_N, 4, 1, false, true>},
        {"generic pseudo-Mersenne (product)", ntt_fwd_kernel<ArithPM<60>, LAB_LOG_N, 4, 1, false, false>, ntt_inv_kernel<ArithPM<60>, LAB_LOG_N, 4, 1, false, true>},
        {"generic pseudo-Mersenne, fwd direct", ntt_fwd_kernel<ArithPM<60>, LAB_LOG_N, 4, 1, false, true>, ntt_inv_kernel<ArithPM<60>, LAB_LOG_N, 4, 1, false, false>},
    };
    const int nv = sizeof(vars) / sizeof(vars[0]);
    for (auto &v : vars) {
        hipFuncSetAttribute((const void *)v.f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
        hipFuncSetAttribute((const void *)v.i, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    }
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    const int reps = 10;
    std::vector<float> sf(nv, 0), si(nv, 0), bf(nv, 1e9f), bi(nv, 1e9f);
    for (int r = 0; r < reps + 2; ++r)
        for (int k = 0; k < nv; ++k) {  // interleaved rounds in one process
            hipEventRecord(e0);
            hipLaunchKernelGGL(vars[k].f, dim3(batch), dim3(C::THREADS), C::LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0);
            hipEventRecord(e1);
            hipLaunchKernelGGL(vars[k].i, dim3(batch), dim3(C::THREADS), C::LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0);
            hipEventRecord(e2);
            hipEventSynchronize(e2);
            float f, i;
            hipEventElapsedTime(&f, e0, e1); hipEventElapsedTime(&i, e1, e2);
            if (r >= 2) { sf[k] += f; si[k] += i; bf[k] = f < bf[k] ? f : bf[k]; bi[k] = i < bi[k] ? i : bi[k]; }
        }
    double bytes = 16.0 * n * batch;
    if (LAB_LOG_N == 14) {
        for (auto &v : v14) {
            hipFuncSetAttribute((const void *)v.f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)N14_LDS_BYTES);
            hipFuncSetAttribute((const void *)v.i, hipFuncAttributeMaxDynamicSharedMemorySize, (int)N14_LDS_BYTES);
            // correctness first: forward must equal the generic kernel's forward, round trip must be the identity
            std::vector<u64> ref(h.size()), got(h.size());
            hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(vars[0].f, dim3(batch), dim3(C::THREADS), C::LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0);
            hipMemcpy(ref.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(v.f, dim3(batch), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0);
            hipMemcpy(got.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            size_t badf = 0, badi = 0;
            for (size_t i = 0; i < h.size(); ++i) badf += got[i] != ref[i];
            hipLaunchKernelGGL(v.i, dim3(batch), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0);
            hipMemcpy(got.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < h.size(); ++i) badi += got[i] != h[i];
            float sf2 = 0, si2 = 0;
            for (int r = 0; r < reps + 2; ++r) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(v.f, dim3(batch), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0);
                hipEventRecord(e1);
                hipLaunchKernelGGL(v.i, dim3(batch), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0);
                hipEventRecord(e2);
                hipEventSynchronize(e2);
                float f, i;
                hipEventElapsedTime(&f, e0, e1); hipEventElapsedTime(&i, e1, e2);
                if (r >= 2) { sf2 += f; si2 += i; }
            }
            printf("%-36s fwd avg %.4f ms %5.0f GB/s | inv avg %.4f ms %5.0f GB/s | fwd mismatches %zu, round-trip mismatches %zu\n", v.name,
                   sf2 / reps, bytes / (sf2 / reps * 1e-3) / 1e9, si2 / reps, bytes / (si2 / reps * 1e-3) / 1e9, badf, badi);
        }
        for (int delay : {0, 3, 6, 9}) {  // persistent, dephased (s_sleep 127 = 8128 clocks... x delay)
            auto pk = ntt14_fwd_persistent<ArithPM<60>>;
            hipFuncSetAttribute((const void *)pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)N14_LDS_BYTES);
            for (int grid : {512, 1024}) {
                float sf2 = 0;
                for (int r = 0; r < reps + 2; ++r) {
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(pk, dim3(grid), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, delay);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float f; hipEventElapsedTime(&f, e0, e1);
                    if (r >= 2) sf2 += f;
                }
                printf("persistent fwd grid %4d delay %3d: %.4f ms\n", grid, delay, sf2 / reps);
            }
        }
        {   // forward-kernel variants, interleaved rounds (fwd variant, then the product inverse), so clocks and thermals are shared
            typedef void (*kern_t2)(u64 *, const ModDesc *, unsigned, unsigned, int);
            struct { const char *name; kern_t2 f; int per; float sum; } mv[] = {
                {"product kernel (ntt14_fwd_kernel<PM60, false>)", ntt14_fwd_kernel<ArithPM<60>, false>, 1, 0},
                {"1 polynomial per workgroup, opaque thread id", ntt14_fwd_multi<ArithPM<60>, 1>, 1, 0},
                {"2 polynomials per workgroup", ntt14_fwd_multi<ArithPM<60>, 2>, 2, 0},
                {"4 polynomials per workgroup", ntt14_fwd_multi<ArithPM<60>, 4>, 4, 0},
                {"persistent, 512 workgroups", ntt14_fwd_persistent<ArithPM<60>>, 0, 0},
                {"persistent, 1024 workgroups", ntt14_fwd_persistent<ArithPM<60>>, -1, 0}};
            for (auto &m : mv) hipFuncSetAttribute((const void *)m.f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)N14_LDS_BYTES);
            for (int r = 0; r < reps + 2; ++r)
                for (auto &m : mv) {
                    const int grid = m.per > 0 ? batch / m.per : (m.per == 0 ? 512 : 1024);
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(m.f, dim3(grid), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0);
                    hipEventRecord(e1);
                    hipLaunchKernelGGL(v14[1].i, dim3(batch), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0);
                    hipEventRecord(e2);
                    hipEventSynchronize(e2);
                    float f; hipEventElapsedTime(&f, e0, e1);
                    if (r >= 2) m.sum += f;
                }
            for (auto &m : mv) printf("fwd A/B, %-48s %.4f ms\n", m.name, m.sum / reps);
        }
        for (int delay : {0, 4}) {
            auto pk = ntt14_fwd_dephased<ArithPM<60>>;
            hipFuncSetAttribute((const void *)pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)N14_LDS_BYTES);
            float sf2 = 0;
            for (int r = 0; r < reps + 2; ++r) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(pk, dim3(batch), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, delay);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float f; hipEventElapsedTime(&f, e0, e1);
                if (r >= 2) sf2 += f;
            }
            printf("dephased fwd delay %3d: %.4f ms\n", delay, sf2 / reps);
            sf2 = 0;
            for (int r = 0; r < reps + 2; ++r) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(v14[1].f, dim3(batch), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float f; hipEventElapsedTime(&f, e0, e1);
                if (r >= 2) sf2 += f;
            }
            printf("   standard fwd kernel, forward-only loop: %.4f ms\n", sf2 / reps);
        }
        {   // occupancy experiment: the same kernel with 100 KiB of dynamic LDS requested -> ONE workgroup per CU
            const size_t big = 100 * 1024;
            hipFuncSetAttribute((const void *)v14[1].f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)big);
            float sf2 = 0;
            for (int r = 0; r < reps + 2; ++r) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(v14[1].f, dim3(batch), dim3(N14_THREADS), big, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float f; hipEventElapsedTime(&f, e0, e1);
                if (r >= 2) sf2 += f;
            }
            printf("ntt14 pseudo-Mersenne fwd, 1 workgroup per CU (100 KiB LDS requested): %.4f ms\n", sf2 / reps);
        }
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
#ifdef NTT14_STAMPS
        {   // where does a workgroup spend its life?  (forward pseudo-Mersenne kernel, the last launch)
            hipLaunchKernelGGL(v14[1].f, dim3(batch), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0);
            hipDeviceSynchronize();
            static unsigned long long hs[4096][16];
            hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamps), sizeof(hs));
            const char *names[] = {"HBM load", "pass0", "xchg01", "pass1", "xchg12", "pass2", "xchg23", "pass3", "store"};
            double sum[9] = {0}; double life = 0; int cnt = 0;
            for (int b = 0; b < batch && b < 4096; ++b) {
                if (hs[b][9] <= hs[b][0]) continue;
                for (int p = 0; p < 9; ++p) sum[p] += double(hs[b][p + 1] - hs[b][p]);
                life += double(hs[b][9] - hs[b][0]); ++cnt;
            }
            printf("stamps (s_memtime ticks, avg over %d workgroups; lifetime %.0f):\n", cnt, life / cnt);
            for (int p = 0; p < 9; ++p) printf("  %-9s %8.0f (%4.1f%%)\n", names[p], sum[p] / cnt, 100.0 * sum[p] / life);
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int b = 0; b < batch && b < 4096; ++b) { if (hs[b][0] < t0) t0 = hs[b][0]; if (hs[b][9] > t1) t1 = hs[b][9]; }
            printf("  kernel span %llu ticks\n", t1 - t0);
        }
#endif
    }
    for (int k = 0; k < nv; ++k)
        printf("%-36s fwd avg %.4f ms (min %.4f) %5.0f GB/s | inv avg %.4f ms (min %.4f) %5.0f GB/s\n", vars[k].name, sf[k] / reps, bf[k],
               bytes / (sf[k] / reps * 1e-3) / 1e9, si[k] / reps, bi[k], bytes / (si[k] / reps * 1e-3) / 1e9);
#if !defined(FHE_ABLATE_NO_COMPUTE) && !defined(FHE_ABLATE_NO_GLOBAL)
    std::vector<u64> out(h.size());
    hipMemcpy(out.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < h.size(); ++i) bad += out[i] != h[i];
    printf("round-trip mismatches: %zu\n", bad);
#endif
    return 0;
}
