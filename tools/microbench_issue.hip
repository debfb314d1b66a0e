// VALU issue-cost microbenchmark for gfx950: cycles per wave-instruction per SIMD, by instruction and by the number of
// waves resident on a SIMD (1, 2, 4, 8), measured with s_memtime inside the kernel (clock independent), plus the cost of
// one whole modular butterfly of each arithmetic policy.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o mbi tools/microbench_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../learn-fhe_amd/csrc/arith.hpp"
using namespace fhe;

#define CH 8      // independent chains
#define UNR 8     // statements per chain per loop trip
#define TRIPS 512

enum { OP_BFLY_DS2, OP_MIX, OP_MAD_VV, OP_MAD_SV, OP_MADI, OP_ADD64, OP_ADD32, OP_AND, OP_MOV, OP_NOP, OP_SUB64, OP_LSHR64, OP_MULLO, OP_MULHI, OP_BFLY_PM, OP_BFLY_DS, OP_BFLY_DS_GS, OP_FOLD_DS, N_OPS };
static const char *op_name[N_OPS] = {"butterfly DS, asm-free sub", "mix 7 mad + 6 x 32-bit", "v_mad_u64_u32 v,v,v", "v_mad_u64_u32 v,s,v", "v_mad_i64_i32 v,v,v", "v_lshl_add_u64", "v_add_u32", "v_and_b32",
                                     "v_mov_b32", "s_nop 0", "64-bit sub (sub_co+subb)", "v_lshrrev_b64", "v_mul_lo_u32", "v_mul_hi_u32",
                                     "butterfly ArithPM<60>::ct", "butterfly ArithDS<60>::ct", "butterfly ArithDS<60>::gs<0>", "fold ArithDS<60>"};
static const double op_insts[N_OPS] = {0.5, 13, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0.5, 0.5, 0.5, 1};  // counted units per chain statement

template <int OP, int W>
__global__ __launch_bounds__(256, W) void kern(u64 *out, unsigned long long *cyc, u64 seed, unsigned sm, const ModDesc *D) {
    u64 x[CH];
    const u64 t = threadIdx.x + blockIdx.x * blockDim.x;
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = seed * (t + c + 1) | 1;
    unsigned m = (unsigned)seed | 5;
    unsigned ms = sm;  // wave-uniform (SGPR)
    const ArithPM<60>::K kp = ArithPM<60>::make(*D, 14, 0, 0);
    const ArithDS<60>::K kd = ArithDS<60>::make(*D, 14, 0, 0);
    const PmTw wp = ArithPM<60>::split(seed & ((1ull << 60) - 1));
    const uint4 wd = ArithDS<60>::split(seed & ((1ull << 59) - 1), kd.m.q);
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < TRIPS; ++i) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if constexpr (OP == OP_MAD_VV) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "v"(m), "v"((unsigned)c + m) : "vcc");
                if constexpr (OP == OP_MAD_SV) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "v"(m), "s"(ms) : "vcc");
                if constexpr (OP == OP_MADI) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(x[c]) : "v"(m), "v"((unsigned)c + m) : "vcc");
                if constexpr (OP == OP_ADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[c]) : "v"(seed));
                if constexpr (OP == OP_ADD32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(*(unsigned *)&x[c]) : "v"(m));
                if constexpr (OP == OP_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(*(unsigned *)&x[c]) : "v"(m));
                if constexpr (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(*(unsigned *)&x[c]) : "v"(m));
                if constexpr (OP == OP_NOP) asm volatile("s_nop 0");
                if constexpr (OP == OP_SUB64) { x[c] -= seed; asm volatile("" : "+v"(x[c])); }
                if constexpr (OP == OP_LSHR64) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(x[c]));
                if constexpr (OP == OP_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(*(unsigned *)&x[c]) : "v"(m));
                if constexpr (OP == OP_MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(*(unsigned *)&x[c]) : "v"(m));
                if constexpr (OP == OP_BFLY_PM) { if ((c & 1) == 0) { ArithPM<60>::ct(x[c], x[c + 1], wp, kp); x[c] &= ~0ull >> 2; x[c + 1] &= ~0ull >> 2; } }
                if constexpr (OP == OP_BFLY_DS) { if ((c & 1) == 0) { ArithDS<60>::ct(x[c], x[c + 1], wd, kd); x[c] &= ~0ull >> 2; x[c + 1] &= ~0ull >> 2; } }
                if constexpr (OP == OP_BFLY_DS_GS) { if ((c & 1) == 0) { ArithDS<60>::gs<0>(x[c], x[c + 1], wd, kd); x[c] &= ~0ull >> 2; } }
                if constexpr (OP == OP_MIX) {
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_u32 %1, %1, %2\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_and_b32 %1, %1, %2\n\t"
                                 "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_u32 %1, %1, %2\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_and_b32 %1, %1, %2\n\t"
                                 "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_u32 %1, %1, %2\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_and_b32 %1, %1, %2\n\t"
                                 "v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[c]), "+v"(m) : "v"((unsigned)c + 77) : "vcc");
                }
                if constexpr (OP == OP_BFLY_DS2) { if ((c & 1) == 0) { ArithDS<60>::ct(x[c], x[c + 1], wd, kd); } }
                if constexpr (OP == OP_FOLD_DS) x[c] = ArithDS<60>::fold1(x[c] + seed, kd.m);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]) : "memory");
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) acc += x[c];
    out[t] = acc;
    if ((threadIdx.x & 63) == 0) cyc[t >> 6] = t1 - t0;
}

template <int OP>
void run(u64 *out, unsigned long long *cyc, const ModDesc *D) {
    std::vector<unsigned long long> h(256 * 8 * 4);
    printf("%-28s", op_name[OP]);
    for (int w : {1, 2, 4, 8}) {  // waves per SIMD: 256-thread blocks put one wave on each SIMD of a CU
        const int blocks = 256 * w;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        auto kf = w == 1 ? kern<OP, 1> : w == 2 ? kern<OP, 2> : w == 4 ? kern<OP, 4> : kern<OP, 8>;
        hipFuncAttributes fa;
        hipFuncGetAttributes(&fa, (const void *)kf);
        kf<<<blocks, 256>>>(out, cyc, 0x9E3779B97F4A7C15ull, 0x12345u, D);
        hipEventRecord(e0);
        kf<<<blocks, 256>>>(out, cyc, 0x9E3779B97F4A7C15ull, 0x12345u, D);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, size_t(blocks) * 4 * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < blocks * 4; ++i) s += double(h[i]);
        const double per_wave = s / (blocks * 4) / (double(TRIPS) * UNR * CH * op_insts[OP]);
        printf("  w=%d: %6.2f cyc/SIMD (%.2f GHz, %d regs%s)", w, per_wave / w, s / (blocks * 4) / (ms * 1e-3) / 1e9, fa.numRegs, fa.localSizeBytes ? ", SPILLS" : "");
    }
    printf("\n");
}

int main() {
    u64 *out; unsigned long long *cyc; ModDesc *D;
    hipMalloc(&out, 8ull * 256 * 8 * 256); hipMalloc(&cyc, 8ull * 256 * 8 * 4); hipMalloc(&D, sizeof(ModDesc));
    ModDesc hd{};
    hd.q = 1152921504606748673ull; hd.pm_b = 60; hd.pm_c = 98303; hd.ds_pow = 1u << 29;
    for (int k = 0; k < 20; ++k) { hd.ninv[k] = 12345 + k; hd.ninv_w[k] = 54321 + k; }
    hipMemcpy(D, &hd, sizeof(hd), hipMemcpyHostToDevice);
    printf("units: cycles per wave-instruction (or per butterfly); cyc/SIMD = cyc/wave / resident waves (throughput cost)\n");
    run<OP_BFLY_DS2>(out, cyc, D); run<OP_MIX>(out, cyc, D);
    run<OP_MAD_VV>(out, cyc, D); run<OP_MAD_SV>(out, cyc, D); run<OP_MADI>(out, cyc, D); run<OP_ADD64>(out, cyc, D);
    run<OP_ADD32>(out, cyc, D); run<OP_AND>(out, cyc, D); run<OP_MOV>(out, cyc, D); run<OP_NOP>(out, cyc, D);
    run<OP_SUB64>(out, cyc, D); run<OP_LSHR64>(out, cyc, D); run<OP_MULLO>(out, cyc, D); run<OP_MULHI>(out, cyc, D);
    run<OP_BFLY_PM>(out, cyc, D); run<OP_BFLY_DS>(out, cyc, D); run<OP_BFLY_DS_GS>(out, cyc, D); run<OP_FOLD_DS>(out, cyc, D);
    return 0;
}
