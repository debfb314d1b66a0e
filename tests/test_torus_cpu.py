"""Row T oracle (oracle/pyref.py): the exact torus arithmetic against the reference's own test properties with fixed seeds --
decode-level TGGSW external product (scheme/tfhe/src/tggsw.rs:150-181) and the full gate bootstrap with LUTs identity /
double / parity over every message (scheme/tfhe/src/bootstrapping.rs:139-165), at a toy ring size."""
import random

from oracle import pyref as P


def test_torus_decompose_recomposes():
    """digits are in [-B/2, B/2] and recompose to the rounded value (decompose.rs:114-135)"""
    rnd = random.Random(1)
    for log_b, d in [(23, 1), (4, 5), (8, 3), (16, 4), (7, 9)]:
        dec = P.TorusDecomposor(log_b, d)
        for _ in range(200):
            v = rnd.getrandbits(64)
            digs = dec.decompose_scalar(v)
            sd = [P.t64_to_i64(x) for x in digs]
            assert all(-(1 << (log_b - 1)) <= x <= (1 << (log_b - 1)) for x in sd)
            rec = sum(x << (dec.rounding_bits + j * log_b) for j, x in enumerate(sd)) % P.M64
            err = P.t64_to_i64((rec - v) % P.M64)
            assert abs(err) <= (1 << dec.rounding_bits) >> 1 if dec.rounding_bits else err == 0


def test_monomial_is_a_product():
    rnd = random.Random(2)
    n = 16
    a = [rnd.getrandbits(64) for _ in range(n)]
    for k in (0, 1, 5, 15, 16, 17, 31, -3):
        mono = [0] * n
        kk = k % (2 * n)
        if kk < n:
            mono[kk] = 1
        else:
            mono[kk - n] = P.M64 - 1
        assert P.torus_monomial_mul(a, k) == P.torus_mul_exact(a, mono)


def test_gate_bootstrap_decode_level_toy():
    rnd = random.Random(3)
    n, n_lwe, log_p, padding = 64, 8, 3, 1
    dec, ksdec = P.TorusDecomposor(12, 3), P.TorusDecomposor(4, 5)
    z = [rnd.randint(0, 1) for _ in range(n_lwe)]
    s = [rnd.randint(0, 1) for _ in range(n)]
    brk = [P.tggsw_sk_encrypt(dec, s, [zi] + [0] * (n - 1), rnd, noise=2) for zi in z]
    ksa, ksb = P.tlwe_ksk_gen(ksdec, z, s, rnd, noise=2)
    log_delta = 64 - (log_p + padding)
    p, m_ = 1 << log_p, n >> log_p

    def table(f):  # scheme/tfhe/src/bootstrapping.rs:116-127
        t = [f(v) % p for v in range(p)]
        out = [t[0]] * (m_ // 2)
        for x in t[1:]:
            out += [x] * m_
        return out + [(-t[0]) % p] * (m_ // 2)

    for f in (lambda v: v, lambda v: 2 * v, lambda v: v % 2):
        v = [(x << log_delta) % P.M64 for x in table(f)]
        for msg in range(p):
            a, b = P.tlwe_sk_encrypt(z, (msg << log_delta) % P.M64, rnd, noise=2)
            acc = P.tfhe_blind_rotate(dec, brk, v, P.tfhe_mod_switch(a, n), P.tfhe_mod_switch([b], n)[0])
            ea, eb = P.tglwe_sample_extract(acc[0], acc[1], 0)
            oa, ob = P.tlwe_key_switch(ksdec, ksa, ksb, ea, eb)
            mu = ((P.tlwe_phase(z, oa, ob) + (1 << (log_delta - 1))) % P.M64) >> log_delta
            assert mu % p == f(msg) % p


# ---- the C restatement of row T (oracle/ref_ring.c) against the Python big-integer restatement ----

def _rows(rnd, count, n):
    return [[rnd.getrandbits(64) for _ in range(n)] for _ in range(count)]


def test_c_oracle_torus_matches_python_oracle():
    import numpy as np
    from oracle import cref
    rnd = random.Random(40)
    U = lambda x: np.array(x, dtype=np.uint64)  # noqa: E731
    L = lambda x: [int(v) for v in np.asarray(x).ravel()]  # noqa: E731
    for log_b, d in [(23, 1), (4, 5), (8, 3), (7, 9), (32, 2), (16, 4)]:
        dec = P.TorusDecomposor(log_b, d)
        v = [0, 1, P.M64 - 1, 1 << 63, (1 << 63) - 1] + [rnd.getrandbits(64) for _ in range(59)]
        assert cref.torus_decompose(log_b, d, U(v)).tolist() == dec.decompose(v)
    for n in (1, 2, 16, 64):
        a, b = [rnd.getrandbits(64) for _ in range(n)], [rnd.getrandbits(64) for _ in range(n)]
        assert L(cref.torus_mul_exact(U(a), U(b))) == P.torus_mul_exact(a, b)
        for k in (0, 1, n - 1, n, n + 1, 2 * n - 1, 2 * n, -1, -n - 3, 5 * n + 2):
            assert L(cref.torus_monomial_mul(U(a), k)) == P.torus_monomial_mul(a, k)
    n, n_lwe, log_b, d = 32, 5, 10, 2
    dec, ksdec = P.TorusDecomposor(log_b, d), P.TorusDecomposor(4, 5)
    brk = [(_rows(rnd, 2 * d, n), _rows(rnd, 2 * d, n)) for _ in range(n_lwe)]
    ca, cb = [rnd.getrandbits(64) for _ in range(n)], [rnd.getrandbits(64) for _ in range(n)]
    ea, eb = cref.tggsw_external_product(log_b, d, U(brk[0][0]), U(brk[0][1]), U(ca), U(cb))
    assert (L(ea), L(eb)) == P.tggsw_external_product(dec, brk[0][0], brk[0][1], ca, cb)
    v = [rnd.getrandbits(64) for _ in range(n)]
    a_raw, b_raw = [[rnd.getrandbits(64) for _ in range(n_lwe)] for _ in range(3)], [rnd.getrandbits(64) for _ in range(3)]
    a_raw[1][2] = 0
    at, bt = cref.tfhe_mod_switch(U(a_raw), n), cref.tfhe_mod_switch(U(b_raw), n)
    assert [L(r) for r in at] == [P.tfhe_mod_switch(r, n) for r in a_raw] and L(bt) == P.tfhe_mod_switch(b_raw, n)
    oa, ob = cref.tfhe_blind_rotate(log_b, d, U([k[0] for k in brk]), U([k[1] for k in brk]), U(v), at, bt, threads=2)
    ksa, ksb = _rows(rnd, n * 5, n_lwe), [rnd.getrandbits(64) for _ in range(n * 5)]
    ga, gb = cref.tfhe_bootstrap(log_b, d, 4, 5, U([k[0] for k in brk]), U([k[1] for k in brk]), U(ksa), U(ksb), U(v), U(a_raw), U(b_raw), threads=2)
    for i in range(3):
        acc = P.tfhe_blind_rotate(dec, brk, v, L(at[i]), int(bt[i]))
        assert L(oa[i]) == acc[0] and L(ob[i]) == acc[1]
        xa, xb = P.tglwe_sample_extract(acc[0], acc[1], 0)
        sa, sb = cref.tglwe_sample_extract(oa[i], ob[i], 0)
        assert (L(sa), sb) == (xa, xb)
        assert (L(cref.tglwe_sample_extract(oa[i], ob[i], 7)[0]), cref.tglwe_sample_extract(oa[i], ob[i], 7)[1]) == P.tglwe_sample_extract(acc[0], acc[1], 7)
        ya, yb = P.tlwe_key_switch(ksdec, ksa, ksb, xa, xb)
        ka, kb = cref.tlwe_key_switch(4, 5, U(ksa), U(ksb), sa, sb)
        assert (L(ka), kb) == (ya, yb) and (L(ga[i]), int(gb[i])) == (ya, yb)


def test_reference_f64_product_is_within_its_own_bound_of_the_exact_one():
    """util/src/ring/fft/c64.rs:186-208 (`precision`): |fft - schoolbook| <= 2^(64 + log_b + log_n - 53) for n 2^8..2^11 and
    b 2^12..2^17, re-run on the C restatement of the reference's floating-point product (the CPU baseline) with fixed seeds; and
    c64.rs:169-184: small operands come out exact.  The GPU path equals the exact product, so it is inside this bound by definition."""
    import numpy as np
    from oracle import cref
    for log_n in range(8, 12):
        n = 1 << log_n
        for log_b in range(12, 18):
            rng = np.random.Generator(np.random.PCG64(log_n * 100 + log_b))
            a = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
            b = rng.integers(-(1 << log_b) + 1, 1 << log_b, size=n, dtype=np.int64).view(np.uint64)
            exact, fft = cref.torus_mul_exact(a, b), cref.torus_mul_fft64(a, b)
            diff = (fft - exact).view(np.int64)
            bound = 1 << (64 + log_b + log_n - 53)
            assert int(np.abs(diff).max()) <= bound, (log_n, log_b, int(np.abs(diff).max()), bound)
    rng = np.random.Generator(np.random.PCG64(5))
    for n in (2, 16, 256, 1024):
        a = rng.integers(-(1 << 15), 1 << 15, size=n, dtype=np.int64).view(np.uint64)
        b = rng.integers(-(1 << 15), 1 << 15, size=n, dtype=np.int64).view(np.uint64)
        assert np.array_equal(cref.torus_mul_fft64(a, b), cref.torus_mul_exact(a, b))


# ---- any TGLWE rank k: the reference's own TGLWE / TGGSW tests run at k = 2 (tglwe.rs:138-166, tggsw.rs:134-181) ----

def _negacyclic_mod(a, b, p):
    n = len(a)
    c = [0] * n
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            if i + j < n:
                c[i + j] += x * y
            else:
                c[i + j - n] -= x * y
    return [v % p for v in c]


def test_rank2_reference_tests_decode_level_on_the_oracle():
    """tglwe.rs:138-166 `encrypt_decrypt`, `sample_extract`; tggsw.rs:150-181 `external_product`, `cmux`: (log_p, padding, big_n, n)
    = (8, 1, N, 2), base 2^8, d = 8 -- at a smaller ring so that pure Python finishes, small integer noise in place of tdg(1e-8)"""
    rnd = random.Random(77)
    k, n, log_p, padding = 2, 32, 8, 1
    dec = P.TorusDecomposor(8, 8)
    p, log_delta = 1 << log_p, 64 - (log_p + padding)
    s = [rnd.randint(0, 1) for _ in range(k * n)]
    enc = lambda m: [(x << log_delta) % P.M64 for x in m]  # noqa: E731  tlwe.rs `encode`
    dec_msg = lambda mu: [(((x + (1 << (log_delta - 1))) % P.M64) >> log_delta) % p for x in mu]  # noqa: E731  tlwe.rs `decode` of `round`
    for _ in range(3):
        m0, m1 = [rnd.randrange(p) for _ in range(n)], [rnd.randrange(p) for _ in range(n)]
        ct1 = P.tglwek_sk_encrypt(k, s, enc(m1), rnd, noise=1 << 20)
        assert dec_msg(P.tglwek_phase(k, s, ct1)) == m1
        for i in (0, 1, n - 1):
            la, lb = P.tglwek_sample_extract(k, ct1, i)
            assert dec_msg([P.tlwe_phase(s, la, lb)])[0] == m1[i]
        gg = P.tggswk_sk_encrypt(k, dec, s, m0, rnd, noise=1 << 20)  # Tggsw::encode: the message itself
        assert dec_msg(P.tglwek_phase(k, s, P.tggswk_external_product(k, dec, gg, ct1))) == _negacyclic_mod(m0, m1, p)
        ct0 = P.tglwek_sk_encrypt(k, s, enc(m0), rnd, noise=1 << 20)
        for bit, want in ((0, m0), (1, m1)):
            sel = P.tggswk_sk_encrypt(k, dec, s, [bit] + [0] * (n - 1), rnd, noise=1 << 20)
            assert dec_msg(P.tglwek_phase(k, s, P.tggswk_cmux(k, dec, sel, ct0, ct1))) == want


def test_c_oracle_rank_k_matches_python_oracle():
    import numpy as np
    from oracle import cref
    rnd = random.Random(41)
    U = lambda x: np.array(x, dtype=np.uint64)  # noqa: E731
    for k, n, log_b, d, n_lwe in [(1, 16, 10, 2, 3), (2, 16, 8, 8, 3), (3, 8, 7, 3, 2)]:
        dec = P.TorusDecomposor(log_b, d)
        k1 = k + 1
        brk = [[[[rnd.getrandbits(64) for _ in range(n)] for _ in range(k1)] for _ in range(k1 * d)] for _ in range(n_lwe)]
        ct0 = [[rnd.getrandbits(64) for _ in range(n)] for _ in range(k1)]
        ct1 = [[rnd.getrandbits(64) for _ in range(n)] for _ in range(k1)]
        assert cref.tggswk_external_product(k, log_b, d, U(brk[0]), U(ct0)).tolist() == P.tggswk_external_product(k, dec, brk[0], ct0)
        assert cref.tggswk_cmux(k, log_b, d, U(brk[1]), U(ct0), U(ct1)).tolist() == P.tggswk_cmux(k, dec, brk[1], ct0, ct1)
        if k == 1:  # the rank-1 case is the k = 1 oracle in the other layout
            ea, eb = P.tggsw_external_product(dec, [r[0] for r in brk[0]], [r[1] for r in brk[0]], ct0[0], ct0[1])
            assert [ea, eb] == P.tggswk_external_product(k, dec, brk[0], ct0)
        v = [rnd.getrandbits(64) for _ in range(n)]
        a_raw, b_raw = [[rnd.getrandbits(64) for _ in range(n_lwe)] for _ in range(2)], [rnd.getrandbits(64) for _ in range(2)]
        at, bt = cref.tfhe_mod_switch(U(a_raw), n), cref.tfhe_mod_switch(U(b_raw), n)
        acc = cref.tfhek_blind_rotate(k, log_b, d, U(brk), U(v), at, bt, threads=2)
        ksa = [[rnd.getrandbits(64) for _ in range(n_lwe)] for _ in range(k * n * 5)]
        ksb = [rnd.getrandbits(64) for _ in range(k * n * 5)]
        ga, gb = cref.tfhek_bootstrap(k, log_b, d, 4, 5, U(brk), U(ksa), U(ksb), U(v), U(a_raw), U(b_raw), threads=2)
        for c in range(2):
            want = P.tfhek_blind_rotate(k, dec, brk, v, [int(x) for x in at[c]], int(bt[c]))
            assert acc[c].tolist() == want
            for i in (0, n - 1, 5):
                la, lb = cref.tglwek_sample_extract(k, acc[c], i)
                assert (la.tolist(), lb) == P.tglwek_sample_extract(k, want, i)
            xa, xb = P.tglwek_sample_extract(k, want, 0)
            ya, yb = P.tlwe_key_switch(P.TorusDecomposor(4, 5), ksa, ksb, xa, xb)
            assert (ga[c].tolist(), int(gb[c])) == (ya, yb)
