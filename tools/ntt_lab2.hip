// Developer lab: A/B of arithmetic policies and ablations inside the N = 2^14 kernels (ntt14w.hpp), interleaved in one process.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o lab2 tools/ntt_lab2.hip      Run: ./lab2 [batch] [reps]
#define FHE_LAB_ABLATE 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
#include "../learn-fhe_amd/csrc/modmath.hpp"
#include "../learn-fhe_amd/csrc/ntt_kernels.hpp"
#include "../learn-fhe_amd/csrc/ntt14w.hpp"
using namespace fhe;

constexpr int N14_THREADS = w14::THREADS;
constexpr size_t N14_LDS_BYTES = w14::LDS_BYTES;
struct DSNoTw : ArithDS<60> {  // ablation: butterflies with a computed twiddle, no twiddle loads (wrong results, same work)
    template <bool INV> static __device__ __forceinline__ TwRaw fetch(const K &k, int idx) { return uint4{k.ninv.x + (unsigned)idx, k.ninv.y, k.ninv.z ^ (unsigned)idx, k.ninv.w}; }
};
template <unsigned M>
struct DSSkip : ArithDS<60> {  // ablation: the twiddles of the layers in M are computed, the others loaded
    static constexpr unsigned SKIP_MASK = M;
    static __device__ __forceinline__ TwRaw fake(const K &k, int idx) { return uint4{k.ninv.x + (unsigned)idx, k.ninv.y, k.ninv.z ^ (unsigned)idx, k.ninv.w}; }
};
template <unsigned M>
struct DSSkipV : ArithDS<60> {  // as DSSkip, but every word of a computed twiddle varies per lane (no scalar halves for the compiler to exploit)
    static constexpr unsigned SKIP_MASK = M;
    static __device__ __forceinline__ TwRaw fake(const K &k, int idx) {
        return uint4{k.ninv.x + (unsigned)idx, (k.ninv.y ^ (unsigned)idx) & 0x7fffffffu, k.ninv.z ^ (unsigned)idx, (k.ninv.w + (unsigned)idx) & 0x7fffffffu};
    }
};
struct DSNone : ArithDS<60> {  // ablation: HBM traffic + exchanges only
    static __device__ __forceinline__ void ct(u64 &X, u64 &Y, const TwReg &, const K &) { X ^= 1; Y ^= 1; }
    template <int PH> static __device__ __forceinline__ void gs(u64 &X, u64 &Y, const TwReg &, const K &) { X ^= 1; Y ^= 1; }
    template <bool INV> static __device__ __forceinline__ TwRaw fetch(const K &, int idx) { return uint4{(unsigned)idx, 1u, 2u, 3u}; }
    static constexpr bool GS_FOLDS = false;
    static __device__ constexpr bool ct_fold_at(int) { return false; }
    static __device__ __forceinline__ u64 fold(u64 x, const K &) { return x; }
    static __device__ __forceinline__ u64 canon_fwd(u64 x, const K &) { return x; }
    static __device__ __forceinline__ u64 finish_inv(u64 x, const K &) { return x; }
    template <int PH = 1> static __device__ __forceinline__ void gs_last_scaled(u64 &X, u64 &Y, const K &) { X ^= 1; Y ^= 1; }
};

// counter calibration: plain coalesced copies of a known byte count, 8 and 16 bytes per lane (the transforms' access widths)
__global__ void copy8_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) out[i] = in[i] + 1;
}
__global__ void copy16_kernel(const ulonglong2 *__restrict__ in, ulonglong2 *__restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        ulonglong2 v = in[i]; v.x += 1; out[i] = v;
    }
}
static int g_pgrid = 512;
#define GRID(v) (batch)
int main(int argc, char **argv) {
    if (argc > 3) g_pgrid = atoi(argv[3]);
    if (argc > 1 && argv[1][0] == 'c') {  // ./lab2 calib : 512 MiB in, 512 MiB out, both widths
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
    const int n = 1 << 14, batch = argc > 1 ? atoi(argv[1]) : 4096, reps = argc > 2 ? atoi(argv[2]) : 20;
    const int s = __builtin_ctzll(q - 1);
    u64 g = smallest_nonresidue(q), omega = powmod(g, (q - 1) >> s, q);
    const int log_cap = s - 1;
    const size_t cap = size_t(1) << log_cap;
    u64 psi_inv = invmod(omega, q);
    std::vector<TwPair> tw(cap), twi(cap);
    std::vector<u64> pw(cap), pwi(cap);
    u64 x = 1, y = 1;
    for (size_t i = 0; i < cap; ++i) { pw[i] = x; pwi[i] = y; x = mulmod(x, omega, q); y = mulmod(y, psi_inv, q); }
    for (size_t j = 0; j < cap; ++j) {
        size_t r = bitrev((unsigned)j, log_cap);
        tw[j] = {pw[r], shoup(pw[r], q)}; twi[j] = {pwi[r], shoup(pwi[r], q)};
    }
    ModDesc hd{};
    hd.q = q; hd.one_s = shoup(1, q);
    for (int k = 0; k < 20; ++k) { hd.ninv[k] = invmod((u64(1) << k) % q, q); hd.ninv_s[k] = shoup(hd.ninv[k], q);
                                  hd.ninv_w[k] = mulmod(hd.ninv[k], twi[1].w, q); hd.ninv_w_s[k] = shoup(hd.ninv_w[k], q); }
    {
        TwPair *d_tw, *d_twi;
        hipMalloc(&d_tw, cap * sizeof(TwPair)); hipMalloc(&d_twi, cap * sizeof(TwPair));
        hipMemcpy(d_tw, tw.data(), cap * sizeof(TwPair), hipMemcpyHostToDevice);
        hipMemcpy(d_twi, twi.data(), cap * sizeof(TwPair), hipMemcpyHostToDevice);
        hd.tw = d_tw; hd.twi = d_twi;
        std::vector<u64> w(cap), wi(cap);
        std::vector<uint4> wd(cap), wdi(cap);
        for (size_t j = 0; j < cap; ++j) {
            w[j] = ArithPM<60>::pack(tw[j].w); wi[j] = ArithPM<60>::pack(twi[j].w);
            wd[j] = ArithDS<60>::split(tw[j].w, q); wdi[j] = ArithDS<60>::split(twi[j].w, q);
        }
        u64 *dw, *dwi; uint4 *dd, *ddi;
        hipMalloc(&dw, cap * 8); hipMalloc(&dwi, cap * 8); hipMalloc(&dd, cap * 16); hipMalloc(&ddi, cap * 16);
        hipMemcpy(dw, w.data(), cap * 8, hipMemcpyHostToDevice);
        hipMemcpy(dwi, wi.data(), cap * 8, hipMemcpyHostToDevice);
        hipMemcpy(dd, wd.data(), cap * 16, hipMemcpyHostToDevice);
        hipMemcpy(ddi, wdi.data(), cap * 16, hipMemcpyHostToDevice);
        hd.tww = dw; hd.twwi = dwi; hd.pm_b = 60; hd.pm_c = (unsigned)((u64(1) << 60) - q);
        hd.twd = dd; hd.twdi = ddi; hd.ds_pow = 1u << (60 - 31);
        for (int k = 0; k < 20; ++k) { hd.ds_ninv[k] = ArithDS<60>::split(hd.ninv[k], q); hd.ds_ninv_w[k] = ArithDS<60>::split(hd.ninv_w[k], q); }
        hd.ds_one = ArithDS<60>::split(1, q);
        {   // diagonal multipliers of the inverse's first pass: twd3i[(p - 1) * stride + blk] = twi[4 blk]^p
            const size_t stride = cap / 4;
            std::vector<uint4> d3(7 * stride);
            for (size_t blk = 1; blk < stride; ++blk) {
                const u64 th = twi[4 * blk].w;
                u64 pw2 = th;
                for (int pp = 0; pp < 7; ++pp) { d3[pp * stride + blk] = ArithDS<60>::split(pw2, q); pw2 = mulmod(pw2, th, q); }
            }
            uint4 *dd3;
            hipMalloc(&dd3, d3.size() * 16);
            hipMemcpy(dd3, d3.data(), d3.size() * 16, hipMemcpyHostToDevice);
            hd.twd3i = dd3; hd.twd3_stride = (unsigned)stride;
        }
    }
    ModDesc *d_desc;
    hipMalloc(&d_desc, sizeof(ModDesc));
    hipMemcpy(d_desc, &hd, sizeof(ModDesc), hipMemcpyHostToDevice);
    std::vector<u64> h(size_t(n) * batch);
    u64 st = 88172645463325252ull;
    for (auto &v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = st % q; }
    // worst-case operands in the first polynomials: all q-1, all zero, alternating
    for (int i = 0; i < n && batch > 3; ++i) { h[i] = q - 1; h[n + i] = 0; h[2 * n + i] = (i & 1) ? q - 1 : 0; }
    u64 *d;
    hipMalloc(&d, h.size() * 8);
    typedef void (*kern_t)(u64 *, const ModDesc *, unsigned, unsigned, int, NttIo);
    struct V { const char *name; kern_t f, i; double sf, si; int persistent; };
    V vs[] = {
        {"wave-local, PM60", ntt14w_fwd_kernel<ArithPM<60>, false>, ntt14w_inv_kernel<ArithPM<60>, false>, 0, 0, 1},
        {"wave-local, DS60", ntt14w_fwd_kernel<ArithDS<60>, false>, ntt14w_inv_kernel<ArithDS<60>, false>, 0, 0, 1},
        {"wave-local, Shoup", ntt14w_fwd_kernel<ArithShoup, false>, ntt14w_inv_kernel<ArithShoup, false>, 0, 0, 1},
        {"wave-local DS60, no twiddle loads", ntt14w_fwd_kernel<DSNoTw, false>, ntt14w_inv_kernel<DSNoTw, false>, 0, 0, 1},
        {"DS60, pass-3 twiddles computed", ntt14w_fwd_kernel<DSSkip<0x3800>, false>, ntt14w_inv_kernel<DSSkip<0x3800>, false>, 0, 0, 1},
        {"DS60, pass-2 twiddles computed", ntt14w_fwd_kernel<DSSkip<0x780>, false>, ntt14w_inv_kernel<DSSkip<0x780>, false>, 0, 0, 1},
        {"DS60, pass-1 twiddles computed", ntt14w_fwd_kernel<DSSkip<0x78>, false>, ntt14w_inv_kernel<DSSkip<0x78>, false>, 0, 0, 1},
        {"DS60, pass-0 twiddles computed", ntt14w_fwd_kernel<DSSkip<0x7>, false>, ntt14w_inv_kernel<DSSkip<0x7>, false>, 0, 0, 1},
        {"DS60, pass-0/1 twiddles computed", ntt14w_fwd_kernel<DSSkip<0x7f>, false>, ntt14w_inv_kernel<DSSkip<0x7f>, false>, 0, 0, 1},
        {"DS60, pass-3 tw computed per lane", ntt14w_fwd_kernel<DSSkipV<0x3800>, false>, ntt14w_inv_kernel<DSSkipV<0x3800>, false>, 0, 0, 1},
        {"DS60, pass-2 tw computed per lane", ntt14w_fwd_kernel<DSSkipV<0x780>, false>, ntt14w_inv_kernel<DSSkipV<0x780>, false>, 0, 0, 1},
        {"DS60, pass-2+3 tw computed per lane", ntt14w_fwd_kernel<DSSkipV<0x3f80>, false>, ntt14w_inv_kernel<DSSkipV<0x3f80>, false>, 0, 0, 1},
        {"wave-local, no butterflies", ntt14w_fwd_kernel<DSNone, false>, ntt14w_inv_kernel<DSNone, false>, 0, 0, 1},
    };
    std::vector<u64> ref(h.size()), got(h.size());
    for (auto &v : vs) {
        hipFuncSetAttribute((const void *)v.f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)N14_LDS_BYTES);
        hipFuncSetAttribute((const void *)v.i, hipFuncAttributeMaxDynamicSharedMemorySize, (int)N14_LDS_BYTES);
        hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(v.f, dim3(GRID(v)), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0, NttIo());
        hipMemcpy(got.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        if (&v == &vs[0]) ref = got;
        size_t badf = 0, badi = 0;
        for (size_t i = 0; i < h.size(); ++i) badf += got[i] != ref[i];
        hipLaunchKernelGGL(v.i, dim3(GRID(v)), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0, NttIo());
        hipMemcpy(got.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < h.size(); ++i) badi += got[i] != h[i];
        hipError_t e = hipDeviceSynchronize();
        printf("%-32s forward mismatches vs PM60 %zu, round-trip mismatches %zu (%s)\n", v.name, badf, badi, hipGetErrorString(e));
    }
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    // pre-heat: ~100 ms of the first variant
    for (int r = 0; r < 150; ++r) {
        hipLaunchKernelGGL(vs[0].f, dim3(batch), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0, NttIo());
        hipLaunchKernelGGL(vs[0].i, dim3(batch), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0, NttIo());
    }
    hipDeviceSynchronize();
    for (int r = 0; r < reps; ++r)
        for (auto &v : vs) {
            // 4 back-to-back pairs per sample so that a sample is ~3 ms of steady load
            float f = 0, i = 0;
            for (int k = 0; k < 4; ++k) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(v.f, dim3(GRID(v)), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0, NttIo());
                hipEventRecord(e1);
                hipLaunchKernelGGL(v.i, dim3(GRID(v)), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0, NttIo());
                hipEventRecord(e2);
                hipEventSynchronize(e2);
                float a, b;
                hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
                f += a; i += b;
            }
            v.sf += f / 4; v.si += i / 4;
        }
    const double bytes = 16.0 * n * batch;
    for (auto &v : vs)
        printf("%-32s fwd %.4f ms %5.0f GB/s (%.3f of 8 TB/s) | inv %.4f ms %5.0f GB/s (%.3f)\n", v.name, v.sf / reps, bytes / (v.sf / reps * 1e-3) / 1e9,
               bytes / (v.sf / reps * 1e-3) / 8e12, v.si / reps, bytes / (v.si / reps * 1e-3) / 1e9, bytes / (v.si / reps * 1e-3) / 8e12);
#ifdef NTT14_STAMPS
    for (int vi : {0, 1, 3})
    for (int dir : {0, 1}) {   // where does a workgroup spend its life?
        auto &v = vs[vi];
        for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(dir ? v.i : v.f, dim3(GRID(v)), dim3(N14_THREADS), N14_LDS_BYTES, 0, d, (const ModDesc *)d_desc, 1u, (unsigned)batch, 0, NttIo());
        hipDeviceSynchronize();
        static unsigned long long hs[4096][16];
        hipMemcpyFromSymbol(hs, HIP_SYMBOL(w14::g_stamps), sizeof(hs));
        const char *fnames[] = {"issue loads", "pass0 (+load wait)", "xchg01", "pass1", "xchg12", "pass2", "xchg23", "pass3", "canon+store issue"};
        const char *inames[] = {"pass3 (+load wait)", "tw issue", "xchg32", "pass2", "xchg21", "pass1", "xchg10", "pass0 l2,l1", "last layer+stores"};
        const char **names = dir ? inames : fnames;
        double sum[9] = {0}; double life = 0; int cnt = 0;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int b = 0; b < batch && b < 4096; ++b) {
            if (hs[b][9] <= hs[b][0]) continue;
            for (int p2 = 0; p2 < 9; ++p2) sum[p2] += double(hs[b][p2 + 1] - hs[b][p2]);
            life += double(hs[b][9] - hs[b][0]); ++cnt;
            if (hs[b][0] < t0) t0 = hs[b][0];
            if (hs[b][9] > t1) t1 = hs[b][9];
        }
        printf("%s %s: stamps (s_memtime ticks, avg over %d workgroups; lifetime %.0f; kernel span %llu ticks)\n", v.name, dir ? "INVERSE" : "forward", cnt, life / cnt, t1 - t0);
        for (int p2 = 0; p2 < 9; ++p2) printf("  %-20s %8.0f (%4.1f%%)\n", names[p2], sum[p2] / cnt, 100.0 * sum[p2] / life);
        {
            double ck = 0; int c2 = 0; unsigned long long r0 = ~0ull, r1 = 0;
            for (int b = 0; b < batch && b < 4096; ++b) {
                if (hs[b][11] <= hs[b][10]) continue;
                ck += double(hs[b][9] - hs[b][0]) / double(hs[b][11] - hs[b][10]) * 0.1; ++c2;
                if (hs[b][10] < r0) r0 = hs[b][10];
                if (hs[b][11] > r1) r1 = hs[b][11];
            }
            // per-CU timelines from the 100 MHz stamps: how long does a CU hold 2 / 1 / 0 workgroups between first start and last end?
            {
                struct Ev { unsigned long long t; int d; };
                for (int from_entry : {0, 1}) {
                std::map<unsigned long long, std::vector<Ev>> cus;
                double pro = 0;
                for (int b = 0; b < batch && b < 4096; ++b) {
                    if (hs[b][11] <= hs[b][10]) continue;
                    const unsigned hw = (unsigned)hs[b][12], xcc = (unsigned)(hs[b][12] >> 32) & 0xf;
                    const unsigned long long key = ((unsigned long long)xcc << 16) | (hw & 0xff00);  // cu_id, sh_id, se_id
                    cus[key].push_back({from_entry ? hs[b][13] : hs[b][10], +1}); cus[key].push_back({hs[b][11], -1});
                    pro += double(hs[b][10] - hs[b][13]);
                }
                double occ[4] = {0}; size_t wgs = 0;
                for (auto &kv : cus) {
                    auto &ev = kv.second;
                    std::sort(ev.begin(), ev.end(), [](const Ev &a, const Ev &b2) { return a.t < b2.t || (a.t == b2.t && a.d < b2.d); });
                    int lvl = 0; unsigned long long prev = ev.front().t < r0 ? ev.front().t : r0;
                    for (auto &e : ev) { occ[lvl > 3 ? 3 : lvl] += double(e.t - prev); prev = e.t; lvl += e.d; }
                    occ[0] += double(r1 - prev);
                    wgs += ev.size() / 2;
                }
                const double tot = double(r1 - r0) * cus.size();
                printf("  [%s] %zu CUs, %.1f workgroups each; CU time holding 0 / 1 / 2 / 3+ workgroups: %.1f%% / %.1f%% / %.1f%% / %.1f%%; entry -> transform start %.2f us avg\n",
                       from_entry ? "kernel entry .. stores drained" : "transform start .. stores drained", cus.size(),
                       double(wgs) / cus.size(), 100 * occ[0] / tot, 100 * occ[1] / tot, 100 * occ[2] / tot, 100 * occ[3] / tot, pro / wgs / 100.0);
                }
            }
            printf("  in-kernel clock %.3f GHz (s_memtime / s_memrealtime); first start -> last end %.1f us\n", ck / c2, double(r1 - r0) / 100.0);
        }
    }
#endif
    return 0;
}
