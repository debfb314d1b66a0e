// Row T on three 30-bit primes (arith30.hpp): the same team-per-ciphertext CMUX / blind-rotation kernels as torus_kernels.hpp,
// with 32-bit transforms.  Used for keys whose exact products fit 2^88 (BASELINE config 5 does: ~2^82).
#pragma once
#include "arith30.hpp"
#include "torus_kernels.hpp"

namespace fhe {

struct Torus30Consts {
    const Mod30Desc *descs;        // three primes p0, p1, p2 in HBM
    unsigned p[3];
    unsigned inv01, inv01_s;       // p0^-1 mod p1 (+ Shoup companion)
    unsigned inv02, inv02_s;       // p0^-1 mod p2
    unsigned inv12, inv12_s;       // p1^-1 mod p2
    u64 p01;                       // p0 p1 (< 2^60)
    u64 P_lo;                      // (p0 p1 p2) mod 2^64
    unsigned h0, h1, h2;           // mixed-radix digits of floor(P / 2): floor(P/2) = h0 + p0 h1 + p0 p1 h2
};

// x = r_k (mod p_k), |x| < P / 2  ->  x mod 2^64   (Garner's mixed-radix digits, all 32-bit products)
__device__ __forceinline__ u64 crt3_mod64(unsigned r0, unsigned r1, unsigned r2, const Torus30Consts &T) {
    const unsigned p1 = T.p[1], p2 = T.p[2];
    const unsigned r0m1 = csub32(r0, p1);                                    // p0 < 2 p1 (all three within a factor 2)
    const unsigned v1 = csub32(Arith30::mulred(r1 + p1 - r0m1, T.inv01, T.inv01_s, p1), p1);
    const unsigned r0m2 = csub32(r0, p2), v1m2 = csub32(v1, p2);
    const unsigned t2 = csub32(Arith30::mulred(r2 + p2 - r0m2, T.inv02, T.inv02_s, p2), p2);   // (r2 - v0) / p0 mod p2
    const unsigned v2 = csub32(Arith30::mulred(t2 + p2 - v1m2, T.inv12, T.inv12_s, p2), p2);   // (... - v1) / p1 mod p2
    const u64 lo = (u64)r0 + (u64)T.p[0] * v1 + T.p01 * v2;                  // x mod 2^64 (wrapping)
    const bool upper = v2 > T.h2 || (v2 == T.h2 && (v1 > T.h1 || (v1 == T.h1 && r0 > T.h0)));
    return upper ? lo - T.P_lo : lo;                                         // centred representative
}

// key rows: [row][a|b][N] 32-bit Montgomery residues per prime; evaluation e = lane * E + r of a row sits at this word offset,
// so that a team's registers load as coalesced 16-byte quads
template <class W>
__host__ __device__ __forceinline__ int key_perm30(int e) {
    const int lane = e / W::E, r = e % W::E;
    return (r >> 2) * (4 * W::TEAM) + lane * 4 + (r & 3);
}

// one key row (a and b of one limb) as a lane holds it; fetched one limb ahead of its multiply-accumulate (fhew_kernels.hpp: KeyRow)
template <class W>
struct KeyRow30 {
    uint4 a[W::E / 4], b[W::E / 4];
};
template <class W>
__device__ __forceinline__ void load_row30(KeyRow30<W> &kr, const unsigned *__restrict__ row, int lane) {
    static_assert(W::E % 4 == 0, "16-byte key loads");
    const uint4 *ka = reinterpret_cast<const uint4 *>(row);
    const uint4 *kb = reinterpret_cast<const uint4 *>(row + W::N);
#pragma unroll
    for (int r4 = 0; r4 < W::E / 4; ++r4) { kr.a[r4] = ka[r4 * W::TEAM + lane]; kr.b[r4] = kb[r4 * W::TEAM + lane]; }
}
template <class W>
__device__ __forceinline__ void mac_row30(const unsigned (&x)[W::E], unsigned (&sa)[W::E], unsigned (&sb)[W::E], const KeyRow30<W> &kr,
                                          const Arith30::K &k) {
#pragma unroll
    for (int r4 = 0; r4 < W::E / 4; ++r4) {
        const uint4 a = kr.a[r4], b = kr.b[r4];
        sa[4 * r4 + 0] = Arith30::mac(sa[4 * r4 + 0], x[4 * r4 + 0], a.x, k);
        sa[4 * r4 + 1] = Arith30::mac(sa[4 * r4 + 1], x[4 * r4 + 1], a.y, k);
        sa[4 * r4 + 2] = Arith30::mac(sa[4 * r4 + 2], x[4 * r4 + 2], a.z, k);
        sa[4 * r4 + 3] = Arith30::mac(sa[4 * r4 + 3], x[4 * r4 + 3], a.w, k);
        sb[4 * r4 + 0] = Arith30::mac(sb[4 * r4 + 0], x[4 * r4 + 0], b.x, k);
        sb[4 * r4 + 1] = Arith30::mac(sb[4 * r4 + 1], x[4 * r4 + 1], b.y, k);
        sb[4 * r4 + 2] = Arith30::mac(sb[4 * r4 + 2], x[4 * r4 + 2], b.z, k);
        sb[4 * r4 + 3] = Arith30::mac(sb[4 * r4 + 3], x[4 * r4 + 3], b.w, k);
    }
}

// one gadget product pass for ONE 30-bit prime over register-resident operands; outputs canonical residues, coefficient layout
template <class W>
__device__ __forceinline__ void team_torus_gadget30(const u64 (&da)[W::E], const u64 (&db)[W::E], const unsigned *__restrict__ rows,
                                                    const TDecomp &P, int lane, unsigned *lds, const Arith30::K &k, unsigned (&sa)[W::E],
                                                    unsigned (&sb)[W::E]) {
    using A = Arith30;
    constexpr int E = W::E;
    u64 st[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { sa[e] = sb[e] = 0; st[e] = tdecomp_init(da[e], P); }
    KeyRow30<W> kr;
    load_row30<W>(kr, rows, lane);
#pragma unroll 1
    for (int j = 0; j < 2 * P.d; ++j) {
        if (j == P.d) {
#pragma unroll
            for (int e = 0; e < E; ++e) st[e] = tdecomp_init(db[e], P);
        }
        unsigned x[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int dg = (int)(long long)tdecomp_next(st[e], P);      // |digit| <= 2^(log_b-1) < p
            x[e] = dg < 0 ? k.p + (unsigned)dg : (unsigned)dg;
        }
        fwd_run<A, typename W::C, W::LOG_N, W::LOG_E, 0, true, W::WAVE>(x, lane, nullptr, lds, true, k);
        mac_row30<W>(x, sa, sb, kr, k);
        if (j + 1 < 2 * P.d) load_row30<W>(kr, rows + size_t(j + 1) * 2 * W::N, lane);
    }
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
        inv_run<A, typename W::C, W::LOG_N, W::LOG_E, W::LOG_N, true, W::WAVE>(sa, lane, nullptr, lds, true, k);
#pragma unroll
        for (int e = 0; e < E; ++e) { const unsigned t = sa[e]; sa[e] = sb[e]; sb[e] = t; }
    }
}

// rows_k: the key entry's rows for prime k.  (xa, xb) <- external_product(key, (da, db)), exact, mod 2^64
template <class W>
__device__ __forceinline__ void team_torus_external_product30(const u64 (&da)[W::E], const u64 (&db)[W::E], const unsigned *__restrict__ rows0,
                                                              const unsigned *__restrict__ rows1, const unsigned *__restrict__ rows2,
                                                              const TDecomp &P, const Torus30Consts &T, int lane, u64 *lds64,
                                                              u64 (&xa)[W::E], u64 (&xb)[W::E]) {
    constexpr int E = W::E;
    unsigned *lds = reinterpret_cast<unsigned *>(lds64);
    unsigned *park = reinterpret_cast<unsigned *>(lds64 + W::PN);  // [prime 0, 1][a | b][E][TEAM]: each lane its own slots
    unsigned sa[E], sb[E];
    const unsigned *rows[3] = {rows0, rows1, rows2};
#pragma unroll 1
    for (int pr = 0; pr < 3; ++pr) {
        const Arith30::K k = Arith30::make(T.descs[pr], W::LOG_N);
        team_torus_gadget30<W>(da, db, rows[pr], P, lane, lds, k, sa, sb);
        if (pr < 2) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                park[((pr * 2 + 0) * E + e) * W::TEAM + lane] = sa[e];
                park[((pr * 2 + 1) * E + e) * W::TEAM + lane] = sb[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        xa[e] = crt3_mod64(park[((0 * 2 + 0) * E + e) * W::TEAM + lane], park[((1 * 2 + 0) * E + e) * W::TEAM + lane], sa[e], T);
        xb[e] = crt3_mod64(park[((0 * 2 + 1) * E + e) * W::TEAM + lane], park[((1 * 2 + 1) * E + e) * W::TEAM + lane], sb[e], T);
    }
}

template <class W>
__device__ __forceinline__ void team_torus_cmux30(u64 (&ca)[W::E], u64 (&cb)[W::E], unsigned r, const unsigned *__restrict__ rows0,
                                                  const unsigned *__restrict__ rows1, const unsigned *__restrict__ rows2, const TDecomp &P,
                                                  const Torus30Consts &T, int lane, u64 *lds) {
    constexpr int E = W::E;
    if (r == 0) return;  // team-uniform
    u64 da[E], db[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { da[e] = ca[e]; db[e] = cb[e]; }
    team_torus_rotate<W>(da, r, lane, lds);
    team_torus_rotate<W>(db, r, lane, lds);
#pragma unroll
    for (int e = 0; e < E; ++e) { da[e] -= ca[e]; db[e] -= cb[e]; }
    u64 xa[E], xb[E];
    team_torus_external_product30<W>(da, db, rows0, rows1, rows2, P, T, lane, lds, xa, xb);
#pragma unroll
    for (int e = 0; e < E; ++e) { ca[e] += xa[e]; cb[e] += xb[e]; }
}

// as torus_cmux_kernel; rows: [3 primes][count][2d][2][N] u32, `per` words per key entry, `plane` words per prime
template <class W>
__global__ __launch_bounds__(W::THREADS, W::MIN_WAVES) void torus30_cmux_kernel(u64 *__restrict__ acc_a, u64 *__restrict__ acc_b, unsigned batch,
                                                                                const unsigned *__restrict__ rows, size_t plane, TDecomp P,
                                                                                const u64 *__restrict__ rot, size_t rot_stride, Torus30Consts T) {
    constexpr int E = W::E, N = W::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const unsigned ct = blockIdx.x * W::TEAMS + team;
    if (ct >= batch) return;
    u64 *lds = reinterpret_cast<u64 *>(smem_raw) + team * W::TORUS_LDS_WORDS;
    u64 *ga = acc_a + size_t(ct) * N, *gb = acc_b + size_t(ct) * N;
    u64 ca[E], cb[E];
    wave_load<W>(ca, ga, lane);
    wave_load<W>(cb, gb, lane);
    if (rot != nullptr) {
        team_torus_cmux30<W>(ca, cb, unsigned(rot[size_t(ct) * rot_stride]) & (2 * N - 1), rows, rows + plane, rows + 2 * plane, P, T, lane, lds);
    } else {
        u64 xa[E], xb[E];
        team_torus_external_product30<W>(ca, cb, rows, rows + plane, rows + 2 * plane, P, T, lane, lds, xa, xb);
#pragma unroll
        for (int e = 0; e < E; ++e) { ca[e] = xa[e]; cb[e] = xb[e]; }
    }
    wave_store<W>(ca, ga, lane);
    wave_store<W>(cb, gb, lane);
}

template <class W>
__global__ __launch_bounds__(W::THREADS, W::MIN_WAVES) void torus30_blind_rotate_kernel(
    const u64 *__restrict__ v, const u64 *__restrict__ a_tilde, const u64 *__restrict__ b_tilde, unsigned n_lwe, unsigned batch,
    const unsigned *__restrict__ rows, size_t plane, TDecomp P, Torus30Consts T, u64 *__restrict__ out_a, u64 *__restrict__ out_b) {
    constexpr int E = W::E, N = W::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const unsigned ct = blockIdx.x * W::TEAMS + team;
    if (ct >= batch) return;
    u64 *lds = reinterpret_cast<u64 *>(smem_raw) + team * W::TORUS_LDS_WORDS;
    u64 ca[E], cb[E];
    wave_load<W>(cb, v, lane);
#pragma unroll
    for (int e = 0; e < E; ++e) ca[e] = 0;
    team_torus_rotate<W>(cb, (2 * N - (unsigned(b_tilde[ct]) & (2 * N - 1))) & (2 * N - 1), lane, lds);
    const size_t per = size_t(2 * P.d) * 2 * N;
    const u64 *a = a_tilde + size_t(ct) * n_lwe;
#pragma unroll 1
    for (unsigned i = 0; i < n_lwe; ++i) {
        const unsigned r = __builtin_amdgcn_readfirstlane(unsigned(a[i]) & (2 * N - 1));
        const unsigned *k0 = rows + i * per;
        team_torus_cmux30<W>(ca, cb, r, k0, k0 + plane, k0 + 2 * plane, P, T, lane, lds);
    }
    wave_store<W>(ca, out_a + size_t(ct) * N, lane);
    wave_store<W>(cb, out_b + size_t(ct) * N, lane);
}

// ---- the blind rotation with a CMUX's digits computed ONCE (log_b <= 8, 2d <= 8: BASELINE config 5's gadget) --------------------
// The kernels above decompose (da, db) three times, once per prime, and keep (da, db) and the decomposition state in registers
// across all three passes: 48 registers, and a third of the decomposition work repeated twice.  A digit of a base <= 2^8 gadget
// is a signed byte: here the 2d E digits of a lane are computed once, packed four to a dword and parked in a lane-private LDS
// area ([limb][E / 4][TEAM] dwords: consecutive lanes on consecutive banks); each prime's pass unpacks them with one
// v_bfe_i32 each.  The registers that frees hold the multiply-accumulate's sums UNREDUCED (Arith30::mac_close).  Same integers
// mod every prime, same CRT: the outputs are bit-identical to the kernels above (tests/test_torus_gpu.py runs both).
template <class W>
__device__ __forceinline__ void park_digits30(const u64 (&da)[W::E], const u64 (&db)[W::E], const TDecomp &P, int lane, unsigned *dig) {
    constexpr int E = W::E;
    // the gadgets this path serves have log_b d <= 28 bits: after the rounding shift the decomposition state fits one dword, and the
    // recurrence of decompose.rs:124-134 runs on 32-bit words (half the instructions of the u64 form; same digits)
    const unsigned mask = (unsigned)P.mask;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        unsigned st[E];
#pragma unroll
        for (int e = 0; e < E; ++e) st[e] = (unsigned)(((half ? db[e] : da[e]) + P.rnd) >> P.rb);
#pragma unroll 1
        for (int j = 0; j < P.d; ++j) {
#pragma unroll
            for (int r4 = 0; r4 < E / 4; ++r4) {
                unsigned w = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    unsigned &c = st[4 * r4 + b];
                    const unsigned limb = c & mask;
                    c >>= P.log_b;
                    const unsigned carry = (((limb - 1) | c) & limb) >> (P.log_b - 1);
                    c += carry;
                    w |= ((limb - (carry << P.log_b)) & 0xffu) << (8 * b);  // |digit| <= 2^(log_b-1) <= 64: a signed byte
                }
                dig[((half * P.d + j) * (E / 4) + r4) * W::TEAM + lane] = w;
            }
        }
    }
}

template <class W>
__device__ __forceinline__ void team_torus_gadget30_pk(const unsigned *dig, const unsigned *__restrict__ rows, int d2, int lane, unsigned *lds,
                                                       const Arith30::K &k, unsigned (&sa)[W::E], unsigned (&sb)[W::E]) {
    using A = Arith30;
    constexpr int E = W::E;
    u64 acc_a[E], acc_b[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc_a[e] = acc_b[e] = 0;
    KeyRow30<W> kr;
    load_row30<W>(kr, rows, lane);
#pragma unroll 1
    for (int j = 0; j < d2; ++j) {
        const int ln = lane;
        unsigned x[E];
#pragma unroll
        for (int r4 = 0; r4 < E / 4; ++r4) {
            const unsigned w = dig[(j * (E / 4) + r4) * W::TEAM + ln];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned dg = (unsigned)((int)(w << (24 - 8 * b)) >> 24);  // sign-extended byte
                x[4 * r4 + b] = min(dg, dg + k.p);                             // negative digits wrap to huge: + p brings them into [0, p)
            }
        }
        fwd_run<A, typename W::C, W::LOG_N, W::LOG_E, 0, true, W::WAVE>(x, ln, nullptr, lds, true, k);
#pragma unroll
        for (int r4 = 0; r4 < E / 4; ++r4) {
            const uint4 a = kr.a[r4], b = kr.b[r4];
            const unsigned ka[4] = {a.x, a.y, a.z, a.w}, kb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned xe = csub32(x[4 * r4 + i], k.p2);  // lazy forward output [0, 4p) -> [0, 2p)
                acc_a[4 * r4 + i] += (u64)xe * ka[i];
                acc_b[4 * r4 + i] += (u64)xe * kb[i];
            }
        }
        if (j + 1 < d2) load_row30<W>(kr, rows + size_t(j + 1) * 2 * W::N, ln);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) { sa[e] = A::mac_close(acc_a[e], k); sb[e] = A::mac_close(acc_b[e], k); }
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
        const int ln = lane;
        inv_run<A, typename W::C, W::LOG_N, W::LOG_E, W::LOG_N, true, W::WAVE>(sa, ln, nullptr, lds, true, k);
#pragma unroll
        for (int e = 0; e < E; ++e) { const unsigned t = sa[e]; sa[e] = sb[e]; sb[e] = t; }
    }
}

template <class W>
__device__ __forceinline__ void team_torus_cmux30_pk(u64 (&ca)[W::E], u64 (&cb)[W::E], unsigned r, const unsigned *__restrict__ rows0,
                                                     const unsigned *__restrict__ rows1, const unsigned *__restrict__ rows2, const TDecomp &P,
                                                     const Torus30Consts &T, int lane, u64 *lds64) {
    constexpr int E = W::E;
    if (r == 0) return;  // team-uniform
    unsigned *dig = reinterpret_cast<unsigned *>(lds64 + W::TORUS_LDS_WORDS);
    {
        u64 da[E], db[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { da[e] = ca[e]; db[e] = cb[e]; }
        team_torus_rotate<W>(da, r, lane, lds64);
        team_torus_rotate<W>(db, r, lane, lds64);
#pragma unroll
        for (int e = 0; e < E; ++e) { da[e] -= ca[e]; db[e] -= cb[e]; }
        park_digits30<W>(da, db, P, lane, dig);
    }
    unsigned *lds = reinterpret_cast<unsigned *>(lds64);
    unsigned *park = reinterpret_cast<unsigned *>(lds64 + W::PN);
    unsigned sa[E], sb[E];
    const unsigned *rows[3] = {rows0, rows1, rows2};
#pragma unroll 1
    for (int pr = 0; pr < 3; ++pr) {
        const Arith30::K k = Arith30::make(T.descs[pr], W::LOG_N);
        team_torus_gadget30_pk<W>(dig, rows[pr], 2 * P.d, lane, lds, k, sa, sb);
        if (pr < 2) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                park[((pr * 2 + 0) * E + e) * W::TEAM + lane] = sa[e];
                park[((pr * 2 + 1) * E + e) * W::TEAM + lane] = sb[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        ca[e] += crt3_mod64(park[((0 * 2 + 0) * E + e) * W::TEAM + lane], park[((1 * 2 + 0) * E + e) * W::TEAM + lane], sa[e], T);
        cb[e] += crt3_mod64(park[((0 * 2 + 1) * E + e) * W::TEAM + lane], park[((1 * 2 + 1) * E + e) * W::TEAM + lane], sb[e], T);
    }
}

// Compiled for two waves per SIMD like the kernel above (204 registers at N = 2^10: FOUR teams per CU, a batch of 1024 is exactly one
// generation of 256 CUs).  Measured and dropped: three waves per SIMD -- 159 registers once the lane index is made opaque per limb so
// that the transforms' exchange addresses are recomputed instead of living across the loop, and with the digit area sized by the
// gadget (31 232 bytes per team) FIVE teams per CU: 17.1 k gates/s at batch 1024 against 25.5 k (a fifth team per CU leaves 52 CUs of
// that batch idle), 20.8 k against 21.8 k at 1280, 21.8 k against 26.4 k at 4096 -- recomputed addresses cost more than a fifth team hides.
template <class W>
__global__ __launch_bounds__(W::THREADS, W::MIN_WAVES) void torus30_blind_rotate_pk_kernel(
    const u64 *__restrict__ v, const u64 *__restrict__ a_tilde, const u64 *__restrict__ b_tilde, unsigned n_lwe, unsigned batch,
    const unsigned *__restrict__ rows, size_t plane, TDecomp P, Torus30Consts T, u64 *__restrict__ out_a, u64 *__restrict__ out_b) {
    constexpr int E = W::E, N = W::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const unsigned ct = blockIdx.x * W::TEAMS + team;
    if (ct >= batch) return;
    // a team's LDS: exchange image | parking area of two residue pairs | the digits of one CMUX, sized by the gadget (2d limbs): at
    // cfg5 (N = 2^10, d = 3) 31 232 bytes -- FIVE teams per CU where the 8-limb maximum would leave room for four
    u64 *lds = reinterpret_cast<u64 *>(smem_raw) + team * (W::TORUS_LDS_WORDS + W::torus_dig_words(2 * P.d));
    u64 ca[E], cb[E];
    wave_load<W>(cb, v, lane);
#pragma unroll
    for (int e = 0; e < E; ++e) ca[e] = 0;
    team_torus_rotate<W>(cb, (2 * N - (unsigned(b_tilde[ct]) & (2 * N - 1))) & (2 * N - 1), lane, lds);
    const size_t per = size_t(2 * P.d) * 2 * N;
    const u64 *a = a_tilde + size_t(ct) * n_lwe;
#pragma unroll 1
    for (unsigned i = 0; i < n_lwe; ++i) {
        const unsigned r = __builtin_amdgcn_readfirstlane(unsigned(a[i]) & (2 * N - 1));
        const unsigned *k0 = rows + i * per;
        team_torus_cmux30_pk<W>(ca, cb, r, k0, k0 + plane, k0 + 2 * plane, P, T, lane, lds);
    }
    wave_store<W>(ca, out_a + size_t(ct) * N, lane);
    wave_store<W>(cb, out_b + size_t(ct) * N, lane);
}

// key preparation for one prime: signed torus rows [rows][2][N] (a | b) -> Montgomery-form evaluations in key_perm30 layout
template <class W>
__global__ __launch_bounds__(W::THREADS) void torus30_key_prepare_kernel(const u64 *__restrict__ rows_a, const u64 *__restrict__ rows_b, size_t n_rows,
                                                                         const Mod30Desc *__restrict__ desc, unsigned *__restrict__ out) {
    constexpr int E = W::E, N = W::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = W::lane(), team = W::team();
    const size_t job = size_t(blockIdx.x) * W::TEAMS + team;  // (row, a|b)
    if (job >= 2 * n_rows) return;
    unsigned *lds = reinterpret_cast<unsigned *>(reinterpret_cast<u64 *>(smem_raw) + team * W::PN);
    const size_t row = job >> 1;
    const u64 *src = ((job & 1) ? rows_b : rows_a) + row * N;
    const Arith30::K k = Arith30::make(*desc, W::LOG_N);
    unsigned x[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const u64 v = src[coef_index<W>(lane, e)];
        const bool neg = (long long)v < 0;
        const unsigned m = (unsigned)((neg ? 0 - v : v) % k.p);  // |v| <= 2^63
        x[e] = (neg && m) ? k.p - m : m;
    }
    fwd_run<Arith30, typename W::C, W::LOG_N, W::LOG_E, 0, true, W::WAVE>(x, lane, nullptr, lds, true, k);
    unsigned *dst = out + (row * 2 + (job & 1)) * N;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned c = Arith30::canon_fwd(x[e], k);
        // Montgomery form c 2^32 mod p = REDC(c * (2^64 mod p))
        dst[key_perm30<W>(lane * E + e)] = csub32(Arith30::redc((u64)c * desc->r2, k), k.p);
    }
}

}  // namespace fhe
