// Host check of learn-fhe_amd/csrc/pm_dot.hpp (the arithmetic of the RNS base conversions on pseudo-Mersenne moduli) against
// unsigned __int128: every function at its stated bounds, then whole extend / rescale rows the way rns_kernels.hpp composes
// them against the reference formulas (util/src/ring/rns.rs:103-132, 331-345) computed term by term with `% q`.
// Built and run by tests/test_pmdot_cpu.py (g++, no GPU).  Exit code 0 = all checks passed.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../learn-fhe_amd/csrc/pm_dot.hpp"

using namespace fhe::pd;
typedef unsigned __int128 u128;

static u64 rng_state = 0x9e3779b97f4a7c15ull;
static u64 rnd() {  // SplitMix64
    u64 z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static u64 mulmod(u64 a, u64 b, u64 q) { return (u64)((u128)a * b % q); }
static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (fails < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } ++fails; } } while (0)

struct Mod { u64 q; unsigned c, c60; int B; };
static Mod make_mod(int B, u64 c) { return Mod{(u64(1) << B) - c, (unsigned)c, (unsigned)(c << (60 - B)), B}; }

// two-operand form of a fixed multiplier (arith.hpp ArithDS::split)
static void ds_split(u64 w, const Mod &m, unsigned out[4]) {
    const u64 w1 = (u64)((((u128)w) << 32) % m.q), lo = (u64(1) << (m.B - 31)) - 1;
    out[0] = (unsigned)(w & lo); out[1] = (unsigned)(w >> (m.B - 31)); out[2] = (unsigned)(w1 & lo); out[3] = (unsigned)(w1 >> (m.B - 31));
}

int main() {
    const int Bs[] = {60, 55, 54, 45, 34};
    for (int B : Bs) {
        const Uni U = make_uni(B);
        const u64 cmax = u64(1) << (B - 33);
        // moduli: odd c up to the eligibility bound (primality is irrelevant for the arithmetic)
        std::vector<Mod> mods;
        for (int t = 0; t < 6; ++t) mods.push_back(make_mod(B, t == 0 ? 1 : t == 1 ? cmax - 1 : (rnd() % cmax) | 1));
        for (const Mod &m : mods) {
            // --- ds_mul_raw + fold: any 64-bit multiplicand, result congruent and below 2^B + 9c after one fold
            for (int it = 0; it < 2000; ++it) {
                const u64 w = it == 0 ? m.q - 1 : rnd() % m.q;
                const u64 y = it < 4 ? ~u64(0) - it : rnd();
                unsigned d[4];
                ds_split(w, m, d);
                const u64 r = ds_mul_raw(y, d[0], d[1], d[2], d[3], 2 * m.c, U);
                CHECK(r < (u64(1) << (B + 3)), "mul_raw bound B=%d", B);
                const u64 f = fold(r, m.c, U);
                CHECK(f < (u64(1) << B) + 9ull * m.c, "fold bound");
                CHECK(f % m.q == (u64)((u128)w * y % m.q), "mul_raw value B=%d", B);
                const u64 hk = rnd() % m.q;  // the HALF form: + a constant below q before the fold
                const u64 g = fold(r + hk, m.c, U);
                CHECK(g < 2 * m.q && g % m.q == (u64)(((u128)w * y + hk) % m.q), "mul_raw + hk");
            }
            // --- reduce_lazy at the stated bounds
            for (int it = 0; it < 4000; ++it) {
                const bool ext = it < 8;
                const u64 s00 = ext ? 11 * (u64(1) << 60) - 1 : rnd() % (11 * (u64(1) << 60));
                const u64 s11 = ext ? 10 * (u64(1) << 60) - 1 : rnd() % (10 * (u64(1) << 60));
                const u64 s01a = ext ? ~u64(0) : rnd();
                const u64 s01b = ext ? (u64(1) << 62) - 1 : rnd() >> 2;
                const u64 r = reduce_lazy(s00, s01a, s01b, s11, m.c, m.c60, U);
                const u128 X = (u128)s00 + (((u128)s01a + s01b) << 30);
                const u64 want = (u64)((X % m.q + (u128)(s11 % m.q) * ((u64(1) << 60) % m.q)) % m.q);
                CHECK(r < (u64(1) << B) + (u64(1) << 31), "reduce bound");
                CHECK(r < 2 * m.q && r % m.q == want, "reduce value B=%d", B);
            }
        }
        // --- whole rows as the kernels compose them: la source limbs (1, 3, 8; 19 for the grouped form), random and extreme residues
        const int las[] = {1, 3, 8, 19};
        for (int la : las) {
            const Mod m = make_mod(B, (rnd() % cmax) | 1);
            const int stride = (la + 7) / 8 * 8;
            for (int it = 0; it < 400; ++it) {
                std::vector<u64> M(la), vs(la);
                for (int i = 0; i < la; ++i) {
                    M[i] = it == 0 ? m.q - 1 : rnd() % m.q;
                    vs[i] = it < 2 ? (u64(1) << B) - 1 - (u64)it : rnd() >> (64 - B);  // canonical residues of OTHER B-bit moduli: below 2^B
                }
                const unsigned u = it == 0 ? (unsigned)la : (unsigned)(rnd() % (la + 1));
                const u64 Uc = it == 0 ? m.q - 1 : rnd() % m.q, X = it == 0 ? m.q - 1 : rnd() % m.q, x = it == 0 ? m.q - 1 : rnd() % m.q,
                          kc = it == 0 ? m.q - 1 : rnd() % m.q;
                for (int xterm = 0; xterm < 2; ++xterm) {
                    // reference value, term by term
                    u128 ref = 0;
                    for (int i = 0; i < la; ++i) ref += (u128)mulmod(M[i], vs[i] % m.q, m.q);
                    ref += (u128)mulmod(Uc, u, m.q);
                    if (xterm) ref += (u128)mulmod(X, x, m.q) + kc;
                    const u64 want = (u64)(ref % m.q);
                    // the kernels' way (rns_kernels.hpp pm_row)
                    u64 total = 0;
                    const int G = (stride) / 8;
                    for (int g = 0; g < G; ++g) {
                        u64 s00 = 0, s11 = 0, sk = 0;
                        for (int i = g * 8; i < g * 8 + 8 && i < la; ++i) {
                            const Y3 y = split30(vs[i]), k = split30(M[i]);
                            s00 += (u64)k.y0 * y.y0; s11 += (u64)k.y1 * y.y1; sk += (u64)k.yk * y.yk;
                        }
                        const u64 s01a = sk - s00 - s11;
                        u64 s01b = 0;
                        if (g == 0) {
                            const Y3 uc = split30(Uc);
                            s00 += (u64)uc.y0 * u; s01b = (u64)uc.y1 * u;
                            if (xterm) {
                                const Y3 xm = split30(X), xv = split30(x);
                                s00 += (u64)xm.y0 * xv.y0 + kc; s11 += (u64)xm.y1 * xv.y1; s01b += (u64)xm.y0 * xv.y1 + (u64)xm.y1 * xv.y0;
                            }
                        }
                        CHECK(s00 < 11 * (u64(1) << 60) && s11 < 10 * (u64(1) << 60) && s01b < (u64(1) << 62), "row bounds");
                        total += reduce_lazy(s00, s01a, s01b, s11, m.c, m.c60, U);
                    }
                    if (G > 1) total = fold(total, m.c, U);
                    CHECK(total < 2 * m.q, "row lazy bound");
                    CHECK(csub(total, m.q) == want, "row value B=%d la=%d xterm=%d", B, la, xterm);
                }
            }
        }
    }
    if (fails) { printf("%d checks failed\n", fails); return 1; }
    printf("pm_dot host checks passed\n");
    return 0;
}
