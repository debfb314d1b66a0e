"""The oracle against the committed golden vectors, the two restatements against each other, and the
reference's own test properties (util/src/ring/fft/zq.rs:94-116, ring.rs:443-452, ring/rns.rs:373-386)."""
import itertools
import math
import os
import random

import numpy as np
import pytest

from oracle import cref as R
from oracle import pyref as P
from conftest import GOLDEN, load_golden

L = lambda x: [int(v) for v in np.asarray(x).ravel()]  # noqa: E731


def test_moduli_golden():
    g = load_golden("moduli.json")
    # values recorded in SURVEY.md section 8(a) row a2/a3
    by = {m["label"]: m for m in g["moduli"]}
    assert by["cfg1"]["q"] == 1073707009 and by["cfg1"]["omega"] == 110668061 and by["cfg1"]["g"] == 13
    assert by["cfg2"]["q"] == 1152921504606748673 and by["cfg2"]["s"] == 15 and by["cfg2"]["g"] == 3
    assert by["cfg2"]["omega"] == 641000223749548346
    assert by["cfg3"]["q"] == 18014398509404161 and by["cfg3"]["s"] == 12 and by["cfg3"]["g"] == 11
    for m in g["moduli"]:
        q = m["q"]
        assert next(P.two_adic_primes(m["bits"], m["log_n"])) == q
        assert R.two_adic_primes(m["bits"], m["log_n"], 1) == [q]
        tw, twi = P.twiddle(q)
        assert tw[:16] == m["tw"] and twi[:16] == m["twi"]
        s, gg, w, ctw, ctwi = R.twiddle_info(q, 16)
        assert (s, gg, w) == (m["s"], m["g"], m["omega"])
        assert L(ctw) == m["tw"] and L(ctwi) == m["twi"]
    assert R.two_adic_primes(60, 16, 16) == g["cfg4_qs"] + g["cfg4_ps"]


def test_ntt_golden():
    g = load_golden("ntt.json")
    for v in g["ntt"]:
        q, n = v["q"], v["n"]
        assert P.nega_cyclic_ntt(q, v["a"]) == v["ntt"]
        assert L(R.ntt_fwd(q, v["a"], n)) == v["ntt"]
        assert P.nega_cyclic_intt(q, v["ntt"]) == v["a"]
        assert L(R.ntt_inv(q, v["ntt"], n)) == v["a"]
    for v in g["mul"]:
        assert L(R.ntt_mul(v["q"], v["a"], v["b"], v["n"])) == v["c"]
        assert L(R.schoolbook_mul(v["q"], v["a"], v["b"])) == v["c"]
    z = np.load(os.path.join(GOLDEN, "ntt_2p14.npz"))
    q = int(z["q"])
    assert np.array_equal(R.ntt_fwd(q, z["a"], 1 << 14), z["ntt"])
    assert np.array_equal(R.ntt_inv(q, z["ntt"], 1 << 14), z["a"])


def test_ntt_output_order_and_root():
    """out[k] = a(psi^(2*bitrev(k)+1)) with psi = omega^(2^(s-1-logN))."""
    q, n = 1073707009, 16
    rng = P.SplitMix64(9)
    a = rng.uniform(q, n)
    s = 11
    psi = pow(P.two_adic_generator(q, s), 1 << (s - 1 - 4), q)
    out = P.nega_cyclic_ntt(q, a)
    for k in range(n):
        br = int(format(k, "04b")[::-1], 2)
        x = pow(psi, 2 * br + 1, q)
        assert out[k] == sum(c * pow(x, i, q) for i, c in enumerate(a)) % q


@pytest.mark.parametrize("log_n", range(0, 10))
def test_reference_properties(log_n):
    """round_trip + nega_cyclic_mul of util/src/ring/fft/zq.rs:94-116, with fixed seeds."""
    n = 1 << log_n
    rng = P.SplitMix64(100 + log_n)
    for q in itertools.islice(P.two_adic_primes(45, log_n + 1), 3 if log_n > 6 else 10):
        a, b = rng.uniform(q, n), rng.uniform(q, n)
        fa = R.ntt_fwd(q, a, n)
        assert L(R.ntt_inv(q, fa, n)) == a
        c = R.ntt_mul(q, a, b, n)
        assert np.array_equal(c, R.schoolbook_mul(q, a, b))
        if log_n <= 7:
            assert P.nega_cyclic_ntt(q, a) == L(fa)
            assert P.nega_cyclic_ntt_mul(q, a, b) == L(c)


def test_decompose_golden_and_cross():
    for v in load_golden("decompose.json"):
        dec = P.Base2Decomposor(v["q"], v["log_b"], v["d"])
        assert dec.rounding_bits == v["rounding_bits"]
        assert dec.decompose(v["in"]) == v["digits"]
        assert [L(r) for r in R.decompose(v["q"], v["log_b"], v["d"], v["in"])] == v["digits"]
        # digits are in [-B/2, B/2] mod q
        half = 1 << (v["log_b"] - 1)
        for row in v["digits"]:
            assert all(x <= half or x >= v["q"] - half for x in row)
    rnd = random.Random(1)
    for bits, log_n, log_b, d in [(28, 10, 7, 4), (54, 10, 6, 9), (45, 10, 5, 9), (55, 12, 11, 5), (60, 15, 12, 5)]:
        q = next(P.two_adic_primes(bits, log_n))
        a = [rnd.randrange(q) for _ in range(256)]
        assert [L(r) for r in R.decompose(q, log_b, d, a)] == P.Base2Decomposor(q, log_b, d).decompose(a)


def test_decompose_recomposition():
    """sum_j digit_j * B^j 2^rb == v up to rounding, for non-negative centred values; the reference's
    negative-representative quirk is kept, so only the v < q/2 half is asserted exact."""
    q = 18014398509404161
    dec = P.Base2Decomposor(q, 6, 9)
    rnd = random.Random(2)
    for _ in range(200):
        v = rnd.randrange(q >> 1)
        digs = dec.decompose_scalar(v)
        rec = sum(P.zq_to_i64(q, dg) << (dec.rounding_bits + 6 * j) for j, dg in enumerate(digs))
        assert abs(rec - v) <= (1 << dec.rounding_bits)


def test_automorphism_monomial_golden():
    g = load_golden("automorphism.json")
    for v in g["automorphism"]:
        assert P.automorphism(v["q"], v["in"], v["t"]) == v["out"]
        assert L(R.automorphism(v["q"], v["t"], v["in"])) == v["out"]
    for v in g["monomial"]:
        assert P.monomial_mul(v["q"], v["in"], v["k"]) == v["out"]
        assert L(R.monomial_mul(v["q"], v["k"], v["in"])) == v["out"]
    # X^k product == schoolbook product with the monomial
    q, n = 35184372060161, 16
    a = P.SplitMix64(3).uniform(q, n)
    for k in (0, 3, 15, 16, 21, 31):
        mono = [0] * n
        if k < n:
            mono[k] = 1
        else:
            mono[k - n] = q - 1
        assert P.monomial_mul(q, a, k) == P.nega_cyclic_schoolbook_mul(q, a, mono)


def test_rlwe_golden():
    v = load_golden("rlwe.json")
    q, lb, d = v["q"], v["log_b"], v["d"]
    dec = P.Base2Decomposor(q, lb, d)
    assert P.rgsw_external_product(q, dec, v["rgsw_a"], v["rgsw_b"], v["ct_a"], v["ct_b"]) == (v["ext_a"], v["ext_b"])
    xa, xb = R.external_product(q, lb, d, v["rgsw_a"], v["rgsw_b"], v["ct_a"], v["ct_b"])
    assert L(xa) == v["ext_a"] and L(xb) == v["ext_b"]
    sa, sb = R.rlwe_key_switch(q, lb, d, v["ksk_a"], v["ksk_b"], v["ct_a"], v["ct_b"])
    assert L(sa) == v["ks_a"] and L(sb) == v["ks_b"]
    ta, tb = R.rlwe_automorphism(q, lb, d, v["auto_t"], v["ksk_a"], v["ksk_b"], v["ct_a"], v["ct_b"])
    assert L(ta) == v["auto_a"] and L(tb) == v["auto_b"]


@pytest.mark.parametrize("log_n", [0, 2, 5])
def test_external_product_decrypt_level(log_n):
    """scheme/fhew/src/rgsw.rs:198-211 and rlwe.rs:379-415 with fixed seeds: decrypt(ct0 [x] ct1) == m0*m1."""
    rnd = random.Random(40 + log_n)
    n, p, log_b, d = 1 << log_n, 16, 5, 9
    q = next(P.two_adic_primes(45, log_n + 1))
    dec = P.Base2Decomposor(q, log_b, d)
    delta = q / p
    enc = lambda m: [P.zq_from_f64(q, float(x) * delta) for x in m]  # noqa: E731
    decd = lambda pt: [P.zq_from_f64(p, float(P.zq_to_i64(q, x)) / delta) for x in pt]  # noqa: E731
    sk = [rnd.randint(-3, 3) for _ in range(n)]
    m0, m1 = [rnd.randrange(p) for _ in range(n)], [rnd.randrange(p) for _ in range(n)]
    ra, rb = P.rgsw_encrypt(q, dec, sk, m0, rnd)
    ca, cb = P.rlwe_sk_encrypt(q, sk, enc(m1), rnd)
    oa, ob = R.external_product(q, log_b, d, ra, rb, ca, cb)
    assert decd(P.rlwe_decrypt(q, sk, L(oa), L(ob))) == P.nega_cyclic_schoolbook_mul(p, m0, m1)
    for t in (5, -5):
        ka, kb = P.rlwe_ak_gen(q, dec, t, sk, rnd)
        ta, tb = R.rlwe_automorphism(q, log_b, d, t, ka, kb, ca, cb)
        assert decd(P.rlwe_decrypt(q, sk, L(ta), L(tb))) == P.automorphism(p, m1, t)
    la, lb_ = R.sample_extract(q, ca, cb, n - 1)
    assert P.rlwe_sample_extract(q, ca, cb, n - 1) == (L(la), lb_)


def test_blind_rotate_golden():
    v = load_golden("blind_rotate.json")
    q, n, w, lb, d = v["q"], v["n"], v["w"], v["log_b"], v["d"]
    sched = [(k, i) for k, i in v["schedule"]]
    assert P.blind_rotate_schedule(n, w, v["lwe_a"]) == sched
    assert R.blind_rotate_schedule(n, w, v["lwe_a"]) == sched
    assert P.ak_t(n, w) == v["ak_t"]
    oa, ob = R.blind_rotate(q, n, w, lb, d, lb, d, np.array(v["brk"], dtype=np.uint64),
                            np.array(v["ak"], dtype=np.uint64), v["ak_t"], v["f"], v["lwe_a"], v["lwe_b"])
    assert L(oa) == v["out_a"] and L(ob) == v["out_b"]


def test_blind_rotate_decrypt_level():
    """constant term of the rotated accumulator == f at the LWE phase (negacyclic), as Fhew::op relies on
    (scheme/fhew/src/fhew.rs:31-40, bootstrapping.rs:149-169)."""
    rnd = random.Random(11)
    log_n, log_b, d, w, n_lwe = 5, 5, 9, 3, 6
    n = 1 << log_n
    q = next(P.two_adic_primes(45, log_n + 1))
    dec = P.Base2Decomposor(q, log_b, d)
    z = [rnd.randint(-1, 1) for _ in range(n)]
    s = [rnd.randint(-1, 1) for _ in range(n_lwe)]
    one = [1] + [0] * (n - 1)
    brk = [P.rgsw_encrypt(q, dec, z, P.monomial_mul(q, one, sj), rnd) for sj in s]
    ts = P.ak_t(n, w)
    ak = [(t,) + P.rlwe_ak_gen(q, dec, t, z, rnd) for t in ts]
    f = [rnd.randrange(q >> 4) << 3 for _ in range(n)]
    brk_arr = np.array([[a, b] for a, b in brk], dtype=np.uint64)
    ak_arr = np.array([[a, b] for _, a, b in ak], dtype=np.uint64)
    for trial in range(4):
        a = [rnd.randrange(n) * 2 + 1 for _ in range(n_lwe)]
        if trial == 1:
            a[3] = 0
        b = rnd.randrange(2 * n)
        mu = (b - sum(x * y for x, y in zip(a, s))) % (2 * n)
        oa, ob = R.blind_rotate(q, n, w, log_b, d, log_b, d, brk_arr, ak_arr, ts, f, a, b)
        pt = P.rlwe_decrypt(q, z, L(oa), L(ob))
        exp = f[mu] if mu < n else P.zq_neg(q, f[mu - n])
        assert abs(P.zq_to_i64(q, (pt[0] - exp) % q)) < (1 << 30)
        if trial == 0:
            assert P.blind_rotate(q, n, w, dec, dec, brk, ak, f, a, b) == (L(oa), L(ob))


def test_rns_golden_and_crt_property():
    v = load_golden("rns.json")
    qs, ps = v["qs"], v["ps"]
    assert P.rns_extend_bases(qs, v["limbs"], ps) == v["extended"]
    assert [L(r) for r in R.rns_extend_bases(qs, ps, v["limbs"])] == v["extended"]
    assert [L(r) for r in R.rns_rescale_k(qs + ps, 2, v["full"])] == v["rescale_k2"]
    assert [L(r) for r in R.rns_rescale_k((qs + ps)[:3], 1, v["full"][:3])] == v["rescale_k1"]
    ob, oa = R.ckks_key_switch(qs, ps, v["ksk_b"], v["ksk_a"], v["ct_b"], v["ct_a"])
    assert [L(r) for r in ob] == v["ks_b"] and [L(r) for r in oa] == v["ks_a"]
    # util/src/ring/rns.rs:373-386: extend_bases preserves the CRT-reconstructed integer
    rnd = random.Random(5)
    for log_n in (0, 3, 6):
        n = 1 << log_n
        gen = P.two_adic_primes(55, log_n + 1)
        qs = [next(gen) for _ in range(8)]
        ps = [next(gen) for _ in range(8)]
        limbs = [[rnd.randrange(qi) for _ in range(n)] for qi in qs]
        ext = [L(r) for r in R.rns_extend_bases(qs, ps, limbs)]
        r0, r1 = P.Rns(qs), P.Rns(qs + ps)
        for i in range(n):
            assert r0.reconstruct([l[i] for l in limbs]) == r1.reconstruct([l[i] for l in limbs] + [l[i] for l in ext])
        for k in (1, 3, 8):
            qps = qs + ps[:k]
            lm = [[rnd.randrange(qi) for _ in range(n)] for qi in qps]
            o = [L(r) for r in R.rns_rescale_k(qps, k, lm)]
            assert o == P.rns_rescale_k(qps, lm, k)
            rq, rr, pp, qq = P.Rns(qps), P.Rns(qs), math.prod(ps[:k]), math.prod(qs)
            for i in range(n):
                x, y = rq.reconstruct([l[i] for l in lm]), rr.reconstruct([l[i] for l in o])
                diff = (y - (x + pp // 2) // pp) % qq
                assert min(diff, qq - diff) <= 1


def test_lwe_golden():
    v = load_golden("lwe.json")
    for m in v["mod_switch"]:
        assert R.mod_switch(v["big_q"], m["v"], 1 << 16) == m["to_2p16"] == P.zq_mod_switch(v["big_q"], m["v"], 1 << 16)
    for m in v["mod_switch_odd"]:
        assert R.mod_switch_odd(v["q_ks"], m["v"], 2048) == m["to_2048"]
        assert m["to_2048"] % 2 == 1 or m["v"] * 2048 < (1 << 16)  # floor == 0 -> plain rounding (zq.rs:135-136)
    rnd = random.Random(8)
    q, dec = 1 << 16, P.Base2Decomposor(1 << 16, 4, 4)
    s0, s1 = [rnd.randint(-2, 2) for _ in range(10)], [rnd.randint(-2, 2) for _ in range(16)]
    ka, kb = P.lwe_ksk_gen(q, dec, s0, s1, rnd)
    a, b = P.lwe_sk_encrypt(q, s1, 12345, rnd)
    oa, ob = P.lwe_key_switch(q, dec, ka, kb, a, b)
    xa, xb = R.lwe_key_switch(q, 4, 4, np.array(ka, dtype=np.uint64), kb, a, b)
    assert L(xa) == oa and xb == ob
    assert abs(P.zq_to_i64(q, (P.lwe_decrypt(q, s0, oa, ob) - 12345) % q)) < 2048


def test_ckks_mul_rotate_c_vs_python():
    """the two oracles agree on `Ckks::mul` / `rotate` (scheme/ckks/src/ckks.rs:250-282)"""
    import numpy as np
    from oracle import cref, pyref as P
    log_n, bits, big_l, big_k = 4, 50, 3, 2
    n = 1 << log_n
    primes = cref.two_adic_primes(bits, log_n + 1, big_l + big_k)
    qs, ps = primes[:big_l], primes[big_l:]
    rng = np.random.Generator(np.random.PCG64(5))
    limbs = lambda mods: np.stack([rng.integers(0, m, size=n, dtype=np.uint64) for m in mods])  # noqa: E731
    kb, ka = limbs(qs + ps), limbs(qs + ps)
    cts = [limbs(qs) for _ in range(4)]
    ints = lambda a: [[int(v) for v in row] for row in a]  # noqa: E731
    eb, ea = cref.ckks_mul(qs, ps, kb, ka, *cts)
    pb, pa = P.ckks_mul(qs, ps, ints(kb), ints(ka), *[ints(c) for c in cts])
    assert ints(eb) == pb and ints(ea) == pa
    for t in (25, -1):
        eb, ea = cref.ckks_rotate(qs, ps, kb, ka, t, cts[0], cts[1])
        pb, pa = P.ckks_rotate(qs, ps, ints(kb), ints(ka), t, ints(cts[0]), ints(cts[1]))
        assert ints(eb) == pb and ints(ea) == pa, t
