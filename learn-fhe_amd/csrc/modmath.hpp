// Host-side modular arithmetic used at context/key set-up time (never on the timed path).
// Mathematically defined functions only: any correct implementation matches the reference's
// num-bigint / num-integer based ones (util/src/zq.rs:99-126, 325-342) bit for bit.
#pragma once
#include <cstdint>
#include <vector>

namespace fhe {

using u64 = unsigned long long;
using u128 = unsigned __int128;

inline u64 mulmod(u64 a, u64 b, u64 q) { return (u64)((u128)a * b % q); }
inline u64 addmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a + b) % q); }
inline u64 submod(u64 a, u64 b, u64 q) { return (u64)(((u128)a + q - (b % q)) % q); }

inline u64 powmod(u64 v, u64 e, u64 q) {
    u64 r = 1 % q;
    v %= q;
    while (e) {
        if (e & 1) r = mulmod(r, v, q);
        v = mulmod(v, v, q);
        e >>= 1;
    }
    return r;
}

// modular inverse by the extended Euclidean algorithm (q need not be prime); 0 if none
inline u64 invmod(u64 v, u64 q) {
    __int128 r0 = q, r1 = v % q, t0 = 0, t1 = 1;
    while (r1 != 0) {
        __int128 k = r0 / r1, tmp;
        tmp = r0 - k * r1; r0 = r1; r1 = tmp;
        tmp = t0 - k * t1; t0 = t1; t1 = tmp;
    }
    if (r0 != 1) return 0;
    t0 %= (__int128)q;
    if (t0 < 0) t0 += q;
    return (u64)t0;
}

// deterministic Miller-Rabin, exact for all 64-bit inputs (bases 2..37)
inline bool is_prime_u64(u64 n) {
    static const u64 bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return false;
    for (u64 p : bases)
        if (n % p == 0) return n == p;
    u64 d = n - 1;
    int r = 0;
    while (!(d & 1)) { d >>= 1; ++r; }
    for (u64 a : bases) {
        u64 x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool composite = true;
        for (int j = 1; j < r; ++j) {
            x = mulmod(x, x, n);
            if (x == n - 1) { composite = false; break; }
        }
        if (composite) return false;
    }
    return true;
}

// smallest g >= 1 with g^((q-1)/2) == q-1  (util/src/zq.rs:99-105)
inline u64 smallest_nonresidue(u64 q) {
    for (u64 g = 1; g < q - 1; ++g)  // zq.rs:99-105: `(1..order)`, order = q - 1 excluded (q = 3 has no candidate: the reference panics)
        if (powmod(g, (q - 1) >> 1, q) == q - 1) return g;
    return 0;
}

// floor(w * 2^64 / q): Shoup companion of a fixed multiplicand w < q
inline u64 shoup(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }

inline unsigned bitrev(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

}  // namespace fhe
