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
