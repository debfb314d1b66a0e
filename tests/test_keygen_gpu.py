"""key material produced on the device -- samplers (util/src/misc/distribution.rs, zq.rs:91-97),
`power_up` (decompose.rs:35-40), `Rlwe::sk_encrypt` (rlwe.rs:146-156), `Rgsw::sk_encrypt` (rgsw.rs:84-105), `Rlwe::ksk_gen` /
`ak_gen` (rlwe.rs:109-132).  The reference's draws come from `thread_rng()`: no parity exists for them, so -- exactly as the
reference tests its own key generation -- the producers are checked at DECRYPT level (the reference's rgsw.rs:198-211 and
rlwe.rs:379-415 tests, run with keys made on the device), statistically, and for determinism; `power_up` is bit-exact."""
import math
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

L = lambda x: [int(v) for v in np.asarray(x).ravel()]  # noqa: E731
U = lambda x: np.array(x, dtype=np.uint64)  # noqa: E731


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint64)


def test_samplers(fhe, torch_cuda):
    like = dev(torch_cuda, U([0]))
    q = 18014398509404161
    n = 1 << 16
    u = host(fhe.sample_uniform(q, 7, 1, like, (n,)))
    assert u.max() < q and abs(float(u.astype(np.float64).mean()) / q - 0.5) < 0.01
    assert len(np.unique(u)) > n - 4
    assert np.array_equal(u, host(fhe.sample_uniform(q, 7, 1, like, (n,))))                # reproducible
    assert not np.array_equal(u, host(fhe.sample_uniform(q, 7, 2, like, (n,))))            # another stream
    assert np.array_equal(u[:1000], host(fhe.sample_uniform(q, 7, 1, like, (1000,))))      # order independent: a prefix is a prefix
    assert np.array_equal(u[:77], fhe.sample_uniform(q, 7, 1, U([0]), (77,)))              # host-memory call: same values
    small = host(fhe.sample_uniform(5, 3, 0, like, (50000,)))
    assert small.max() == 4 and all(abs(int((small == v).sum()) - 10000) < 600 for v in range(5))
    t = host(fhe.sample_torus(9, 0, like, (n,)))
    assert abs(float((t >> np.uint64(63)).mean()) - 0.5) < 0.01 and len(np.unique(t)) == n
    # dg(3.2, 6): support [-19, 19], variance ~ 3.2^2, symmetric; as Zq values and as plain integers
    e = host(fhe.sample_dg(0, 3.2, 6, 11, 0, like, (n,))).view(np.int64)
    assert e.min() >= -19 and e.max() <= 19
    assert abs(float(e.mean())) < 0.05 and abs(float(e.var()) - 3.2 ** 2) < 0.3
    ez = host(fhe.sample_dg(q, 3.2, 6, 11, 0, like, (n,)))
    assert np.array_equal(ez, np.where(e < 0, q - (-e), e).astype(np.uint64))
    # the weights are the reference's: cdf differences with the A&S 7.1.26 erf
    def erf_as(x):
        p, a1, a2, a3, a4, a5 = 0.3275911, 0.254829592, -0.284496736, 1.421413741, -1.453152027, 1.061405429
        tt = 1.0 / (1.0 + p * abs(x))
        pos = 1.0 - (((((a5 * tt + a4) * tt) + a3) * tt + a2) * tt + a1) * tt * math.exp(-x * x)
        return pos if x >= 0 else -pos
    cdf = lambda x: (1.0 + erf_as(x / (3.2 * math.sqrt(2)))) / 2.0  # noqa: E731
    w = [cdf(i + 0.5) - cdf(i - 0.5) for i in range(-19, 20)]
    big = host(fhe.sample_dg(0, 3.2, 6, 12, 0, like, (1 << 20,))).view(np.int64)
    for i in (-6, -1, 0, 1, 3, 9):
        exp = w[i + 19] / sum(w) * (1 << 20)
        assert abs(int((big == i).sum()) - exp) < 6 * math.sqrt(exp) + 5, i


def test_power_up_bit_exact(fhe, torch_cuda):
    from oracle import pyref as P
    rnd = random.Random(5)
    for q, log_b, d in [(18014398509404161, 6, 9), (268409857, 7, 4), (35184372065281, 5, 9), (1 << 16, 4, 4)]:
        dec = P.Base2Decomposor(q, log_b, d)
        v = [[rnd.randrange(q) for _ in range(16)] for _ in range(3)]
        out = host(fhe.power_up(q, log_b, d, dev(torch_cuda, U(v)), 16)).reshape(3, d, 16)
        for p in range(3):
            assert out[p].tolist() == dec.power_up_poly(v[p])


def test_device_made_keys_decrypt_level(fhe, torch_cuda):
    """rgsw.rs:198-211 (external product), rlwe.rs:345-351 (encrypt/decrypt), rlwe.rs:379-415 (key switch, automorphism) with every
    ciphertext and key produced by the device-side generators; decryption by the oracle"""
    from oracle import pyref as P
    rnd = random.Random(31)
    log_n, p, log_b, d = 7, 16, 5, 9
    n = 1 << log_n
    q = next(P.two_adic_primes(45, log_n + 1))
    delta = q / p
    enc = lambda m: [P.zq_from_f64(q, float(x) * delta) for x in m]  # noqa: E731
    decd = lambda pt: [P.zq_from_f64(p, float(P.zq_to_i64(q, x)) / delta) for x in pt]  # noqa: E731
    ctx = fhe.NttContext(q)
    like = dev(torch_cuda, U([0]))
    # secret keys: dg(3.2, 6) on the device (rlwe.rs:94-96)
    sk_z = fhe.sample_dg(q, 3.2, 6, 100, 0, like, (n,))
    sk2_z = fhe.sample_dg(q, 3.2, 6, 100, 1, like, (n,))
    sk = [P.zq_to_i64(q, v) for v in L(host(sk_z))]
    sk2 = [P.zq_to_i64(q, v) for v in L(host(sk2_z))]
    assert max(abs(v) for v in sk) <= 19
    m0, m1 = [rnd.randrange(p) for _ in range(n)], [rnd.randrange(p) for _ in range(n)]
    # encrypt / decrypt
    ca, cb = fhe.rlwe_sk_encrypt(ctx, sk_z, dev(torch_cuda, U([enc(m1), enc(m0)])), n, 2, 200, 0)
    assert decd(P.rlwe_decrypt(q, sk, L(host(ca)[0]), L(host(cb)[0]))) == m1 and decd(P.rlwe_decrypt(q, sk, L(host(ca)[1]), L(host(cb)[1]))) == m0
    za, zb = fhe.rlwe_sk_encrypt(ctx, sk_z, None, n, 3, 201, 0)   # encryptions of zero: the phase is the dg noise
    for i in range(3):
        ph = [P.zq_to_i64(q, x) for x in P.rlwe_decrypt(q, sk, L(host(za)[i]), L(host(zb)[i]))]
        assert max(abs(x) for x in ph) <= 19 and any(ph)
    assert not np.array_equal(host(za)[0], host(za)[1])
    # RGSW(m0) made on the device, external product with RLWE(m1)
    ra, rb = fhe.rgsw_encrypt(ctx, log_b, d, sk_z, dev(torch_cuda, U([m0, m1])), n, 300, 0)
    rgsw = fhe.GadgetKey(ctx, log_b, d, ra, rb, n, rgsw=True)
    a, b = ca[:1].clone(), cb[:1].clone()
    rgsw.external_product_(0, a, b)
    assert decd(P.rlwe_decrypt(q, sk, L(host(a)), L(host(b)))) == P.nega_cyclic_schoolbook_mul(p, m0, m1)
    a, b = ca[:1].clone(), cb[:1].clone()
    rgsw.external_product_(1, a, b)
    assert decd(P.rlwe_decrypt(q, sk, L(host(a)), L(host(b)))) == P.nega_cyclic_schoolbook_mul(p, m1, m1)
    # key switch sk2 -> sk: a ciphertext under sk2 decrypts under sk afterwards (rlwe.rs:379-391)
    ka, kb = fhe.rlwe_ksk_gen(ctx, log_b, d, sk_z, sk2_z, 0, n, 400, 0)
    ksk = fhe.GadgetKey(ctx, log_b, d, ka, kb, n, rgsw=False)
    c2a, c2b = fhe.rlwe_sk_encrypt(ctx, sk2_z, dev(torch_cuda, U([enc(m1)])), n, 1, 401, 0)
    ksk.key_switch_(0, c2a, c2b)
    assert decd(P.rlwe_decrypt(q, sk, L(host(c2a)), L(host(c2b)))) == m1
    # automorphism keys (rlwe.rs:401-415)
    for t in (5, -5):
        aa, ab = fhe.rlwe_ksk_gen(ctx, log_b, d, sk_z, None, t, n, 500 + t, 0)
        ak = fhe.GadgetKey(ctx, log_b, d, aa, ab, n, rgsw=False)
        a, b = ca[:1].clone(), cb[:1].clone()
        ak.automorphism_(0, t, a, b)
        assert decd(P.rlwe_decrypt(q, sk, L(host(a)), L(host(b)))) == P.automorphism(p, m1, t)
    # host-memory operands take the same path
    ha, hb = fhe.rgsw_encrypt(ctx, log_b, d, host(sk_z), U([m0, m1]), n, 300, 0)
    assert np.array_equal(ha, host(ra).reshape(ha.shape)) and np.array_equal(hb, host(rb).reshape(hb.shape))


def test_whole_bootstrapping_key_made_on_the_device(fhe, torch_cuda):
    """`Bootstrapping::key_gen` (scheme/fhew/src/bootstrapping.rs:122-146) with every key made by the device-side producers -- LWE
    key-switching key (lwe.rs:108-119), brk = RGSW(X^s_j) (rgsw.rs:84-105), ak = automorphism keys for ak_t (rlwe.rs:122-132), the
    secret keys themselves from the device's dg(3.2, 6) -- then the reference's own gate test (fhew/boolean.rs:256-290, its
    `single_key_testing_param`) at decode level, inputs encrypted on the device as well (lwe.rs:128-139)."""
    from oracle import pyref as P
    log_q, log_n, log_b, d, w = 28, 9, 7, 4, 10
    n_lwe, q_ks, kb, kd = 100, 1 << 16, 4, 4
    n = 1 << log_n
    q = next(P.two_adic_primes(log_q, log_n + 1))
    like = dev(torch_cuda, U([0]))
    ctx = fhe.NttContext(q)
    z_i = host(fhe.sample_dg(0, 3.2, 6, 900, 0, like, (n,))).view(np.int64)            # ring key z (rlwe.rs:94-96)
    s_i = host(fhe.sample_dg(0, 3.2, 6, 900, 1, like, (n_lwe,))).view(np.int64)        # LWE key s (lwe.rs:103-106)
    zq = lambda v, m: dev(torch_cuda, U([int(x) % m for x in v]))  # noqa: E731
    z_q, z_ks, s_ks = zq(z_i, q), zq(z_i, q_ks), zq(s_i, q_ks)
    # brk_j = RGSW(X^{s_j}): the monomial plaintexts (bootstrapping.rs:131-136)
    mono = np.zeros((n_lwe, n), dtype=np.uint64)
    for j, sj in enumerate(s_i):
        e = int(sj) % (2 * n)
        mono[j, e % n] = 1 if e < n else q - 1
    ra, rb = fhe.rgsw_encrypt(ctx, log_b, d, z_q, dev(torch_cuda, mono), n, 901, 0)
    ts = P.ak_t(n, w)
    aks = [fhe.rlwe_ksk_gen(ctx, log_b, d, z_q, None, t, n, 902, i) for i, t in enumerate(ts)]
    ak_a = torch_cuda.stack([k[0] for k in aks]).contiguous()
    ak_b = torch_cuda.stack([k[1] for k in aks]).contiguous()
    ksk_a, ksk_b = fhe.lwe_ksk_gen(q_ks, kb, kd, s_ks, z_ks, 903, 0)                   # Lwe::ksk_gen(lwe_s, &s, z)
    gk = fhe.GadgetKey(ctx, log_b, d, ra, rb, n, rgsw=True)
    ga = fhe.GadgetKey(ctx, log_b, d, ak_a, ak_b, n, rgsw=False)
    bk = fhe.BootstrapKey(ctx, gk, ga, ts, w)
    ev = fhe.Fhew(bk, q_ks, kb, kd, ksk_a, ksk_b)
    delta = q / 4.0
    z = [int(v) for v in z_i]
    stream = [0]

    def encrypt(bits):
        stream[0] += 1
        pt = dev(torch_cuda, U([P.zq_from_f64(q, float(m) * delta) for m in bits]))
        return fhe.lwe_sk_encrypt(q, z_q, pt, n, len(bits), 904, stream[0])

    def decrypt(ct):
        a, b = host(ct[0]).reshape(-1, n), host(ct[1])
        out = []
        for i in range(a.shape[0]):
            m = P.zq_from_f64(4, float(P.lwe_decrypt(q, z, L(a[i]), int(b[i]))) / delta)
            assert m in (0, 1), m
            out.append(m)
        return out

    m0 = [(m >> 0) & 1 for m in range(4)]
    m1 = [(m >> 1) & 1 for m in range(4)]
    c0, c1 = encrypt(m0), encrypt(m1)
    assert decrypt(c0) == m0 and decrypt(c1) == m1
    assert decrypt(ev.nand(c0, c1)) == [1 - (x & y) for x, y in zip(m0, m1)]
    assert decrypt(ev.and_(c0, c1)) == [x & y for x, y in zip(m0, m1)]
    assert decrypt(ev.or_(c0, c1)) == [x | y for x, y in zip(m0, m1)]
    assert decrypt(ev.xor(c0, c1)) == [x ^ y for x, y in zip(m0, m1)]
    assert decrypt(ev.xor(ev.nand(c0, c1), ev.or_(c0, c1))) == [(1 - (x & y)) ^ (x | y) for x, y in zip(m0, m1)]
    # `FhewU8::wrapping_add` (fhew/uint8.rs:65-90, its `op` test 306-340): a ripple adder of `FhewBool::overflowing_add` / `carrying_add`
    # (fhew/boolean.rs:139-150: t = a ^ b, sum = t ^ c, carry = (a & b) | (t & c)) -- 37 gate bootstraps in sequence on 12 pairs of bytes
    rnd = random.Random(9)
    xs, ys = [rnd.randrange(256) for _ in range(12)], [rnd.randrange(256) for _ in range(12)]
    xs[0], ys[0], xs[1], ys[1] = 255, 1, 255, 255
    xb, yb = [encrypt([(x >> i) & 1 for x in xs]) for i in range(8)], [encrypt([(y >> i) & 1 for y in ys]) for i in range(8)]
    out_bits, carry = [], None
    for i in range(8):
        t = ev.xor(xb[i], yb[i])
        if carry is None:
            out_bits.append(t)
            carry = ev.and_(xb[i], yb[i])
        else:
            out_bits.append(ev.xor(t, carry))
            if i < 7:
                carry = ev.or_(ev.and_(xb[i], yb[i]), ev.and_(t, carry))
    got = [sum(b << i for i, b in enumerate(bits)) for bits in zip(*[decrypt(c) for c in out_bits])]
    assert got == [(x + y) & 255 for x, y in zip(xs, ys)]


def test_rq_sum(fhe, torch_cuda):
    """util/src/ring.rs:328-341 `Rq: Sum`, prime and non-prime moduli, against exact integers"""
    rng = np.random.Generator(np.random.PCG64(77))
    for q, count, n in [(18014398509404161, 19, 256), (1 << 16, 7, 64), (1152921504606748673, 5, 1024)]:
        a = rng.integers(0, q, size=(count, n), dtype=np.uint64)
        exp = [sum(int(a[k, i]) for k in range(count)) % q for i in range(n)]
        assert L(host(fhe.rq_sum(q, dev(torch_cuda, a), n))) == exp
        assert L(fhe.rq_sum(q, a, n)) == exp


def test_tfhe_bootstrap_with_device_made_keys(fhe, torch_cuda):
    """The reference's own `bootstrap` test (scheme/tfhe/src/bootstrapping.rs:139-165) with ITS parameter set -- big_n = 2048,
    k = 1, base 2^23 x 1, n_lwe = 1024, key switch (4, 5), log_p 4, padding 1, std_dev 1.34e-7 (TLWE) / 2.85e-15 (TGGSW) -- and every
    key and ciphertext made by the device-side producers: `Tlwe::sk_gen` (binary), `Bootstrapping::key_gen` (bootstrapping.rs:59-76:
    brk_i = TGGSW(z_i) under s, ksk = Tlwe::ksk_gen(z, s)), `Tlwe::sk_encrypt`; LUTs identity / double / parity over all 16 messages
    through the single-call gate, decoded on the host (tlwe.rs:134-142)."""
    from oracle import pyref as P
    n, n_lwe, log_p, padding, log_b, d, ks_lb, ks_d = 2048, 1024, 4, 1, 23, 1, 4, 5
    sd_lwe, sd_glwe = 1.339775301998614e-7, 2.845267479601915e-15
    p, log_delta = 1 << log_p, 64 - (log_p + padding)
    like = dev(torch_cuda, U([0]))
    t = fhe.TorusContext()
    z, s = fhe.sample_binary(700, 0, like, n_lwe), fhe.sample_binary(700, 1, like, n)
    zh = L(host(z))
    assert set(zh) == {0, 1} and 0.4 < sum(zh) / n_lwe < 0.6
    pt = np.zeros((n_lwe, n), dtype=np.uint64)
    pt[:, 0] = host(z)                                            # Rt::constant(z_i) (bootstrapping.rs:66-67)
    ra, rb = fhe.tggsw_encrypt(t, log_b, d, s, dev(torch_cuda, pt), n, sd_glwe, 701, 0)
    key = fhe.TggswKey(t, log_b, d, ra, rb, n)
    ksa, ksb = fhe.tlwe_ksk_gen(ks_lb, ks_d, z, s, sd_lwe, 702, 0)

    def table(f):
        m_ = n >> log_p
        tt = [f(v) % p for v in range(p)]
        out = [tt[0]] * (m_ // 2)
        for x in tt[1:]:
            out += [x] * m_
        return out + [(-tt[0]) % p] * (m_ // 2)

    for li, f in enumerate((lambda v: v, lambda v: 2 * v, lambda v: v % 2)):
        v = dev(torch_cuda, U([(x << log_delta) % P.M64 for x in table(f)]))
        msgs = dev(torch_cuda, U([(m << log_delta) % P.M64 for m in range(p)]))
        ca, cb = fhe.tlwe_sk_encrypt(z, msgs, n_lwe, p, sd_lwe, 703, li)
        for m in range(p):                                         # the inputs decrypt to their messages (tlwe.rs:134-142)
            mu = ((P.tlwe_phase(zh, L(host(ca)[m]), int(host(cb)[m])) + (1 << (log_delta - 1))) % P.M64) >> log_delta
            assert mu % p == m
        oa, ob = key.bootstrap(ks_lb, ks_d, ksa, ksb, v, ca, cb)
        for m in range(p):
            mu = ((P.tlwe_phase(zh, L(host(oa)[m]), int(host(ob)[m])) + (1 << (log_delta - 1))) % P.M64) >> log_delta
            assert mu % p == f(m) % p, (li, m, mu)


def test_tdg_and_tglwe_encrypt(fhe, torch_cuda):
    """`tdg` (distribution.rs:49-54): centred, the standard deviation asked for (as a fraction of the torus); `Tglwe::sk_encrypt`
    (tglwe.rs:91-103): b - a s - pt is that noise, with a s the exact product of row T"""
    from oracle import cref
    like = dev(torch_cuda, U([0]))
    sd = 2.0 ** -20
    e = host(fhe.sample_tdg(sd, 9, 0, like, 1 << 16)).view(np.int64).astype(np.float64) / 2.0 ** 64
    assert abs(e.mean()) < 4 * sd / 256 and 0.97 * sd < e.std() < 1.03 * sd
    assert np.array_equal(host(fhe.sample_tdg(sd, 9, 0, like, 100)), host(fhe.sample_tdg(sd, 9, 0, like, 1 << 16))[:100])
    n, rows = 512, 3
    t = fhe.TorusContext()
    s = fhe.sample_binary(10, 0, like, n)
    rng = np.random.Generator(np.random.PCG64(3))
    pt = rng.integers(0, 1 << 63, size=(rows, n), dtype=np.uint64)
    a, b = fhe.tglwe_sk_encrypt(t, s, dev(torch_cuda, pt), n, rows, sd, 11, 0)
    for r in range(rows):
        noise = (host(b)[r] - cref.torus_mul_exact(host(a)[r], host(s)) - pt[r]).view(np.int64).astype(np.float64) / 2.0 ** 64
        assert abs(noise).max() < 6 * sd and noise.std() > 0.5 * sd


def test_rng_domain_separation_and_stream_ids(fhe, torch_cuda):
    """(generator, stream id) reused ACROSS entry points must not share keystream -- the public mask of an encryption
    made with the pair a secret key was sampled with reveals nothing of that key -- and the documented contract holds: the same
    entry point with the same pair reproduces its draw, FHE_STREAM_AUTO never does, a 256-bit key is honoured in full."""
    like = dev(torch_cuda, U([0]))
    n, rows = 1024, 4
    rng = fhe.Rng(key=bytes(range(32)))
    # raw keystream words under (rng, 7) as fhe_sample_torus sees them, the binary secret key, and the masks of three encryptions
    words = set(L(host(fhe.sample_torus(rng, 7, like, (4 * n * rows,)))))
    sk = fhe.sample_binary(rng, 7, like, n)
    a_tlwe, _ = fhe.tlwe_sk_encrypt(sk, None, n, rows, 2.0 ** -30, rng, 7)
    t = fhe.TorusContext()
    a_tglwe, _ = fhe.tglwe_sk_encrypt(t, sk, None, n, rows, 2.0 ** -30, rng, 7)
    assert not words & set(L(host(a_tlwe))) and not words & set(L(host(a_tglwe))) and not set(L(host(a_tlwe))) & set(L(host(a_tglwe)))
    # the key's bits are not the low (or any fixed) bit of the mask words drawn with the same pair
    skb = host(sk).astype(np.uint64) & np.uint64(1)
    for bit in range(64):
        agree = float((((host(a_tlwe)[0] >> np.uint64(bit)) & np.uint64(1)) == skb).mean())
        assert 0.4 < agree < 0.6, (bit, agree)
    # same entry point, same pair: reproducible; another stream id or another purpose: independent
    assert np.array_equal(host(fhe.sample_binary(rng, 7, like, n)), host(sk))
    assert not np.array_equal(host(fhe.sample_binary(rng, 8, like, n)), host(sk))
    q = 18014398509404161
    assert not np.array_equal(host(fhe.sample_uniform(q, rng, 7, like, (n,))), host(fhe.sample_torus(rng, 7, like, (n,))) % np.uint64(q))
    # FHE_STREAM_AUTO: a fresh stream per call
    x1, x2 = host(fhe.sample_torus(rng, fhe.STREAM_AUTO, like, (n,))), host(fhe.sample_torus(rng, fhe.STREAM_AUTO, like, (n,)))
    assert not np.array_equal(x1, x2)
    # all 256 key bits matter: flipping the last one changes every draw; operating-system keys differ from each other
    other = fhe.Rng(key=bytes(range(31)) + b"\x9f")
    assert not set(L(host(fhe.sample_torus(other, 7, like, (n,))))) & words
    assert not np.array_equal(host(fhe.sample_torus(fhe.Rng(), 0, like, (n,))), host(fhe.sample_torus(fhe.Rng(), 0, like, (n,))))


def test_reference_tlwe_tests_decode_level(fhe, torch_cuda):
    """The reference's own TLWE tests at ITS parameters on device-made material: scheme/tfhe/src/tlwe.rs:162-174 `encrypt_decrypt`
    ((log_p, padding, n, std_dev) = (8, 1, 256, 1e-8), every message 0 .. 255) and 176-192 `key_switch` (decomposor base 2^8 x 8: sk0 -> sk1,
    `ksk_gen(param1, sk1, sk0)`, every message decodes under sk1)."""
    log_p, padding, n, sd, log_b, d = 8, 1, 256, 1.0e-8, 8, 8
    p, log_delta = 1 << log_p, 64 - (log_p + padding)
    like = dev(torch_cuda, U([0]))
    sk0, sk1 = fhe.sample_binary(950, 0, like, n), fhe.sample_binary(950, 1, like, n)
    s0, s1 = host(sk0).astype(np.uint64), host(sk1).astype(np.uint64)
    msgs = np.arange(p, dtype=np.uint64)
    ca, cb = fhe.tlwe_sk_encrypt(sk0, dev(torch_cuda, msgs << np.uint64(log_delta)), n, p, sd, 951, 0)

    def decode(a, b, s):  # tlwe.rs:134-142: mu* = b - <a, s>, rounded at log_delta, then `decode`
        ph = b - (a * s[None, :]).sum(axis=1, dtype=np.uint64)
        return ((ph + np.uint64(1 << (log_delta - 1))) >> np.uint64(log_delta)) % np.uint64(p)

    assert np.array_equal(decode(host(ca), host(cb), s0), msgs)
    ksa, ksb = fhe.tlwe_ksk_gen(log_b, d, sk1, sk0, sd, 952, 0)      # rows encrypt -sk0_i 2^(8 j) under sk1 (tlwe.rs:100-111)
    oa, ob = fhe.tlwe_key_switch(log_b, d, ksa, ksb, ca, cb, n, n)
    assert np.array_equal(decode(host(oa).reshape(p, n), host(ob).reshape(p), s1), msgs)
    assert not np.array_equal(decode(host(oa).reshape(p, n), host(ob).reshape(p), s0), msgs)   # and no longer under sk0


@pytest.mark.parametrize("log_n", [1, 5, 9])
def test_reference_multi_key_encrypt_decrypt(fhe, torch_cuda, log_n):
    """The reference's `multi_key_encrypt_decrypt` (scheme/fhew/src/rlwe.rs:434-459: three parties, q ~ 2^45, p = 16) on the device
    entries: every party's public-key share is `share_encrypt` of zero on the common reference string a (`pk_share_gen`, rlwe.rs:217-225),
    the merged key (a, sum of shares) encrypts (`pk_encrypt`, 158-170), every party contributes a decryption share a' s_i + e_i
    (`share_decrypt`, 260-269) and b' minus their sum decodes to the message (`decryption_share_merge`, 271-274).  Also `Rlwe::decrypt`
    (172-175) under the summed key, and pk_gen / pk_encrypt / decrypt of one party (rlwe.rs:345-360 `encrypt_decrypt`)."""
    from oracle import pyref as P
    parties, p = 3, 16
    n = 1 << log_n
    q = next(P.two_adic_primes(45, log_n + 1))
    delta = q / p
    enc = lambda m: [P.zq_from_f64(q, float(x) * delta) for x in m]  # noqa: E731
    decd = lambda pt: [P.zq_from_f64(p, float(P.zq_to_i64(q, int(x))) / delta) for x in pt]  # noqa: E731
    ctx = fhe.NttContext(q)
    like = dev(torch_cuda, U([0]))
    rnd = random.Random(60 + log_n)
    a = fhe.sample_uniform(q, 600, 0, like, (n,))                                   # the common reference string
    sks = [fhe.sample_dg(q, 3.2, 6, 601, i, like, (n,)) for i in range(parties)]   # rlwe.rs:94-96 `sk_gen`
    shares = torch_cuda.stack([fhe.rlwe_share_encrypt(ctx, a, sk, None, n, 1, 602, i)[0] for i, sk in enumerate(sks)])
    pk_b = fhe.rq_sum(q, shares.contiguous(), n)                                    # `pk_share_merge`: (a, sum of the shares)
    batch = 4
    msgs = [[rnd.randrange(p) for _ in range(n)] for _ in range(batch)]
    ca, cb = fhe.rlwe_pk_encrypt(ctx, a, pk_b, dev(torch_cuda, U([enc(m) for m in msgs])), n, batch, 603, 0)
    assert not np.array_equal(host(ca)[0], host(ca)[1])                             # a fresh u per ciphertext
    dsh = torch_cuda.stack([fhe.rlwe_share_encrypt(ctx, ca, sk, None, n, batch, 604, i) for i, sk in enumerate(sks)])   # [parties][batch][n]
    for c in range(batch):
        tot = fhe.rq_sum(q, dsh[:, c].contiguous(), n)
        pt = (host(cb)[c].astype(object) - host(tot).astype(object)) % q            # b - sum of the shares
        assert decd(pt) == msgs[c], c
    # the same through Rlwe::decrypt under the summed key (no fresh decryption noise)
    sk_sum = fhe.rq_sum(q, torch_cuda.stack(sks).contiguous(), n)
    pts = host(fhe.rlwe_decrypt(ctx, sk_sum, ca, cb, n))
    for c in range(batch):
        assert decd(L(pts[c])) == msgs[c], c
    # one party alone: pk_gen (an encryption of zero), pk_encrypt, decrypt in place of ct_b
    za, zb = fhe.rlwe_sk_encrypt(ctx, sks[0], None, n, 1, 605, 0)
    ea, eb = fhe.rlwe_pk_encrypt(ctx, za[0].contiguous(), zb[0].contiguous(), dev(torch_cuda, U([enc(m) for m in msgs])), n, batch, 606, 0)
    one = host(fhe.rlwe_decrypt(ctx, sks[0], ea, eb, n))
    for c in range(batch):
        assert decd(L(one[c])) == msgs[c], c
    noise = [P.zq_to_i64(q, (int(x) - y) % q) for x, y in zip(one[0], enc(msgs[0]))]
    assert 0 < max(abs(v) for v in noise) <= 19 * 19 * n + 19 + 19 * n                  # e u + e1 + e0 s with |e|, |s| <= 19, |u| <= 1


def test_reference_rgsw_tests_with_public_key_encryption(fhe, torch_cuda):
    """The reference's RGSW tests encrypt with the PUBLIC key (scheme/fhew/src/rgsw.rs:162-230: `key_gen`, `pk_encrypt`): `external_product`
    (198-211: RGSW_pk(m0) x RLWE_pk(m1) decrypts to m0 m1) and `internal_product` (214-230: RGSW_pk(m0) x RGSW_pk(m1), then the external
    product of the result with an encryption of 1... here: with RLWE_pk(m2), decrypting to m0 m1 m2) -- keys and ciphertexts from
    fhe_rlwe_sk_encrypt (pk_gen), fhe_rlwe_pk_encrypt, fhe_rgsw_pk_encrypt, decrypted by fhe_rlwe_decrypt."""
    from oracle import pyref as P
    rnd = random.Random(71)
    log_n, p, log_b, d = 7, 16, 5, 9
    n = 1 << log_n
    q = next(P.two_adic_primes(45, log_n + 1))
    delta = q / p
    enc = lambda m: [P.zq_from_f64(q, float(x) * delta) for x in m]  # noqa: E731
    decd = lambda pt: [P.zq_from_f64(p, float(P.zq_to_i64(q, int(x))) / delta) for x in pt]  # noqa: E731
    ctx = fhe.NttContext(q)
    like = dev(torch_cuda, U([0]))
    sk = fhe.sample_dg(q, 3.2, 6, 700, 0, like, (n,))
    za, zb = fhe.rlwe_sk_encrypt(ctx, sk, None, n, 1, 701, 0)                       # rgsw.rs:50-52 `key_gen`: pk = an encryption of zero
    pk_a, pk_b = za[0].contiguous(), zb[0].contiguous()
    m0, m1, m2 = ([rnd.randrange(2) for _ in range(n)] for _ in range(3))           # small messages: the products stay below p
    m0 = [0] * n; m0[3] = 1                                                         # a monomial, as the blind rotation's RGSW messages are
    ra, rb = fhe.rgsw_pk_encrypt(ctx, log_b, d, pk_a, pk_b, dev(torch_cuda, U([m0, m1])), n, 702, 0)
    key = fhe.GadgetKey(ctx, log_b, d, ra, rb, n, rgsw=True)
    ca, cb = fhe.rlwe_pk_encrypt(ctx, pk_a, pk_b, dev(torch_cuda, U([enc(m1), enc(m2)])), n, 2, 703, 0)
    a, b = ca[:1].clone(), cb[:1].clone()
    key.external_product_(0, a, b)                                                  # RGSW(m0) x RLWE(m1)
    assert decd(L(host(fhe.rlwe_decrypt(ctx, sk, a, b, n)))) == P.nega_cyclic_schoolbook_mul(p, m0, m1)
    rows_a, rows_b = ra[1:2].clone(), rb[1:2].clone()                               # RGSW(m0) x RGSW(m1) -> RGSW(m0 m1) (rgsw.rs:130-150)
    key.internal_product_(0, rows_a, rows_b)
    prod = fhe.GadgetKey(ctx, log_b, d, rows_a, rows_b, n, rgsw=True)
    a, b = ca[1:2].clone(), cb[1:2].clone()
    prod.external_product_(0, a, b)
    want = P.nega_cyclic_schoolbook_mul(p, P.nega_cyclic_schoolbook_mul(p, m0, m1), m2)
    assert decd(L(host(fhe.rlwe_decrypt(ctx, sk, a, b, n)))) == want


def test_reference_multi_key_gates(fhe, torch_cuda):
    """The reference's `multi_key_op` (scheme/fhew/src/fhew/boolean.rs:337-386) at ITS `multi_key_testing_param` (q ~ 2^54, N = 512, base 2^6 x 9,
    LWE n = 100 over 2^16 with (4, 4), w = 10, p = 4) with THREE parties, every share and every merge on the device entries:
    `Bootstrapping::crs_gen` (bootstrapping.rs:256-273) = uniform draws; `Rlwe::pk_share_gen` / `pk_share_merge` (rlwe.rs:217-235);
    `key_share_gen` (bootstrapping.rs:275-298): a fresh LWE key per party, `Lwe::ksk_share_gen` on the common masks (lwe.rs:214-226),
    brk rows = `Rgsw::pk_encrypt(X^{s_j})` under the merged public key, `Rlwe::ak_share_gen` per automorphism (rlwe.rs:305-314) =
    share_encrypt of power_up(-z(X^t)); `key_share_merge` (300-320): sums of the shares and `Rgsw::internal_product` across the parties;
    inputs `FhewBool::pk_encrypt` (boolean.rs:31-40: public-key RLWE encryption of the constant, sample_extract(0)); outputs decrypted by
    `share_decrypt` / `decryption_share_merge` (lwe.rs:197-212).  Gates not / and / nand / or / nor / xor / xnor / majority over all inputs."""
    from oracle import pyref as P
    parties, p = 3, 4
    log_q, log_n, log_b, d, w = 54, 9, 6, 9, 10
    n_lwe, q_ks, kb, kd = 100, 1 << 16, 4, 4
    n = 1 << log_n
    q = next(P.two_adic_primes(log_q, log_n + 1))
    ctx = fhe.NttContext(q)
    like = dev(torch_cuda, U([0]))
    ts = P.ak_t(n, w)
    zq = lambda v, m: dev(torch_cuda, U([int(x) % m for x in v]))  # noqa: E731
    # common reference string
    crs_pk = fhe.sample_uniform(q, 800, 0, like, (n,))
    crs_ksk = fhe.sample_uniform(q_ks, 800, 1, like, (n * kd, n_lwe))
    crs_ak = fhe.sample_uniform(q, 800, 2, like, (len(ts), d, n))
    # the parties' ring keys and the merged public key
    z_i = [host(fhe.sample_dg(0, 3.2, 6, 801, i, like, (n,))).view(np.int64) for i in range(parties)]
    z_q = [zq(z, q) for z in z_i]
    pk_b = fhe.rq_sum(q, torch_cuda.stack([fhe.rlwe_share_encrypt(ctx, crs_pk, z_q[i], None, n, 1, 802, i)[0] for i in range(parties)]).contiguous(), n)
    # bootstrapping-key shares
    ksk_shares, brk_rows, ak_shares = [], [], []
    for i in range(parties):
        s = host(fhe.sample_dg(0, 3.2, 6, 803, i, like, (n_lwe,))).view(np.int64)      # the party's own LWE key
        ksk_shares.append(fhe.lwe_ksk_share_gen(q_ks, kb, kd, crs_ksk, zq(s, q_ks), zq(z_i[i], q_ks), 804, i))
        mono = np.zeros((n_lwe, n), dtype=np.uint64)
        for j, sj in enumerate(s):
            e = int(sj) % (2 * n)
            mono[j, e % n] = 1 if e < n else q - 1
        brk_rows.append(fhe.rgsw_pk_encrypt(ctx, log_b, d, crs_pk, pk_b, dev(torch_cuda, mono), n, 805, i))
        shares_t = []
        for ti, t in enumerate(ts):
            z_auto = fhe.automorphism(q, t, z_q[i].reshape(1, n), n)
            pt = fhe.power_up(q, log_b, d, fhe.rq_neg(q, z_auto), n).reshape(d, n).contiguous()   # power_up(-z(X^t)) (rlwe.rs:283)
            shares_t.append(fhe.rlwe_share_encrypt(ctx, crs_ak[ti].contiguous(), z_q[i], pt, n, d, 806, i * 100 + ti))
        ak_shares.append(torch_cuda.stack(shares_t))                                         # [|ts|][d][n]
    # merge
    ksk_b = fhe.rq_sum(q_ks, torch_cuda.stack(ksk_shares).contiguous(), n * kd)
    acc_a, acc_b = brk_rows[0]
    for i in range(1, parties):                                                              # reduce(|acc, item| internal_product(acc, item))
        key = fhe.GadgetKey(ctx, log_b, d, acc_a, acc_b, n, rgsw=True)
        nxt_a, nxt_b = brk_rows[i][0].clone(), brk_rows[i][1].clone()
        for j in range(n_lwe):
            key.internal_product_(j, nxt_a[j:j + 1], nxt_b[j:j + 1])
        acc_a, acc_b = nxt_a, nxt_b
    ak_b = torch_cuda.stack([fhe.rq_sum(q, torch_cuda.stack([ak_shares[i][ti] for i in range(parties)]).contiguous(), d * n).reshape(d, n)
                             for ti in range(len(ts))]).contiguous()
    bk = fhe.BootstrapKey(ctx, fhe.GadgetKey(ctx, log_b, d, acc_a, acc_b, n, rgsw=True), fhe.GadgetKey(ctx, log_b, d, crs_ak.contiguous(), ak_b, n, rgsw=False), ts, w)
    ev = fhe.Fhew(bk, q_ks, kb, kd, crs_ksk, ksk_b)
    delta = q / float(p)
    stream = [0]

    def encrypt(bits):  # FhewBool::pk_encrypt
        stream[0] += 1
        pt = np.zeros((len(bits), n), dtype=np.uint64)
        pt[:, 0] = [P.zq_from_f64(q, float(m) * delta) for m in bits]
        ca, cb = fhe.rlwe_pk_encrypt(ctx, crs_pk, pk_b, dev(torch_cuda, pt), n, len(bits), 807, stream[0])
        return fhe.rlwe_sample_extract(q, ca, cb, n, 0)

    def decrypt(ct):    # share_decrypt per party, decryption_share_merge, Fhew::decode
        stream[0] += 1
        a, b = ct
        shares = torch_cuda.stack([fhe.lwe_share_encrypt(q, a.reshape(-1, n).contiguous(), z_q[i], None, n, 808, stream[0] * 10 + i) for i in range(parties)])
        tot = fhe.rq_sum(q, shares.contiguous(), shares.shape[1])
        out = []
        for bv, sv in zip(L(host(b)), L(host(tot))):
            m = P.zq_from_f64(p, float(P.zq_to_i64(q, (bv - sv) % q)) / delta)
            assert m in (0, 1), m
            out.append(m)
        return out

    bit = lambda k: [(m >> k) & 1 for m in range(8)]  # noqa: E731
    m0, m1, m2 = bit(0), bit(1), bit(2)
    c0, c1, c2 = encrypt(m0), encrypt(m1), encrypt(m2)
    assert decrypt(c0) == m0 and decrypt(c2) == m2
    assert decrypt(ev.not_(c0)) == [1 - x for x in m0]
    assert decrypt(ev.and_(c0, c1)) == [x & y for x, y in zip(m0, m1)]
    assert decrypt(ev.nand(c0, c1)) == [1 - (x & y) for x, y in zip(m0, m1)]
    assert decrypt(ev.or_(c0, c1)) == [x | y for x, y in zip(m0, m1)]
    assert decrypt(ev.nor(c0, c1)) == [1 - (x | y) for x, y in zip(m0, m1)]
    assert decrypt(ev.xor(c0, c1)) == [x ^ y for x, y in zip(m0, m1)]
    assert decrypt(ev.xnor(c0, c1)) == [1 - (x ^ y) for x, y in zip(m0, m1)]
    assert decrypt(ev.majority(c0, c1, c2)) == [(x & y) | (y & z) | (z & x) for x, y, z in zip(m0, m1, m2)]
