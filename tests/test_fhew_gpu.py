"""GPU parity tests for SURVEY.md section 8(a) rows a8-a13 through the C ABI, bit-exact against the oracle
(oracle/cref.py, oracle/pyref.py) on identical seeded inputs and against the committed golden vectors."""
import random

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

L = lambda x: [int(v) for v in np.asarray(x).ravel()]  # noqa: E731


def rand_u64(seed, q, shape):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, q, size=shape, dtype=np.uint64)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint64)


def test_decompose_golden_and_oracle(fhe, cref, torch_cuda):
    for v in load_golden("decompose.json"):
        a = np.array(v["in"], dtype=np.uint64)
        n = a.size
        out = fhe.decompose(v["q"], v["log_b"], v["d"], dev(torch_cuda, a), n)
        assert host(out).reshape(v["d"], n).tolist() == v["digits"]
        out_h = fhe.decompose(v["q"], v["log_b"], v["d"], a, n)  # host-memory entry path
        assert out_h.reshape(v["d"], n).tolist() == v["digits"]
    for bits, log_n, log_b, d in [(28, 10, 7, 4), (54, 10, 6, 9), (45, 10, 5, 9), (55, 12, 11, 5), (60, 15, 12, 5)]:
        q = cref.two_adic_primes(bits, log_n, 1)[0]
        n, polys = 256, 5
        a = rand_u64(bits, q, (polys, n))
        a[0, :6] = [0, 1, q - 1, q >> 1, (q >> 1) - 1, (q >> 1) + 1]
        out = host(fhe.decompose(q, log_b, d, dev(torch_cuda, a), n))
        for p in range(polys):
            assert np.array_equal(out[p], cref.decompose(q, log_b, d, a[p]))
    q = 1 << 16  # non-prime modulus (LWE key switch)
    a = rand_u64(3, q, (1, 100))
    assert np.array_equal(host(fhe.decompose(q, 4, 4, dev(torch_cuda, a), 100))[0], cref.decompose(q, 4, 4, a[0]))


def test_automorphism_monomial(fhe, cref, torch_cuda):
    g = load_golden("automorphism.json")
    for v in g["automorphism"]:
        a = np.array(v["in"], dtype=np.uint64)
        assert L(host(fhe.automorphism(v["q"], v["t"], dev(torch_cuda, a), a.size))) == v["out"]
    for v in g["monomial"]:
        a = np.array(v["in"], dtype=np.uint64)
        assert L(host(fhe.monomial_mul(v["q"], v["k"], dev(torch_cuda, a), a.size))) == v["out"]
    q = 18014398509404161
    for n in (1, 2, 64, 1024):
        a = rand_u64(n, q, (3, n))
        a[0, 0] = 0
        for t in (5, -5, 25, 1, -1, 2 * n - 1, 3, 2 * n + 5):
            out = host(fhe.automorphism(q, t, dev(torch_cuda, a), n))
            for p in range(3):
                assert np.array_equal(out[p], cref.automorphism(q, t, a[p])), (n, t)
        for k in (0, 1, -1, n, n - 1, n + 1, 2 * n - 1, -n, 3 * n + 5, -7):
            out = host(fhe.monomial_mul(q, k, dev(torch_cuda, a), n))
            for p in range(3):
                assert np.array_equal(out[p], cref.monomial_mul(q, k, a[p])), (n, k)


def test_gadget_products_golden(fhe, torch_cuda):
    v = load_golden("rlwe.json")
    q, n, lb, d = v["q"], v["n"], v["log_b"], v["d"]
    ctx = fhe.NttContext(q)
    U = lambda x: np.array(x, dtype=np.uint64)  # noqa: E731
    rgsw = fhe.GadgetKey(ctx, lb, d, dev(torch_cuda, U(v["rgsw_a"])), dev(torch_cuda, U(v["rgsw_b"])), n, rgsw=True)
    ksk = fhe.GadgetKey(ctx, lb, d, U(v["ksk_a"]), U(v["ksk_b"]), n, rgsw=False)  # host-memory key rows
    a, b = dev(torch_cuda, U(v["ct_a"])), dev(torch_cuda, U(v["ct_b"]))
    rgsw.external_product_(0, a, b)
    assert L(host(a)) == v["ext_a"] and L(host(b)) == v["ext_b"]
    a, b = dev(torch_cuda, U(v["ct_a"])), dev(torch_cuda, U(v["ct_b"]))
    ksk.key_switch_(0, a, b)
    assert L(host(a)) == v["ks_a"] and L(host(b)) == v["ks_b"]
    a, b = U(v["ct_a"]).copy(), U(v["ct_b"]).copy()  # host-memory ciphertext
    ksk.automorphism_(0, v["auto_t"], a, b)
    assert L(a) == v["auto_a"] and L(b) == v["auto_b"]
    with pytest.raises(fhe.FheError):  # an RGSW key is not a key-switching key
        rgsw.key_switch_(0, dev(torch_cuda, U(v["ct_a"])), dev(torch_cuda, U(v["ct_b"])))


@pytest.mark.parametrize("log_n,bits,log_b,d", [(7, 45, 5, 9), (8, 28, 7, 4), (9, 28, 7, 4), (9, 54, 6, 9), (10, 54, 6, 9), (11, 55, 11, 5),
                                                (11, 54, 9, 6)])
def test_gadget_products_vs_oracle(fhe, cref, torch_cuda, log_n, bits, log_b, d):
    """reference parameter sets (rgsw.rs:164-227 (5,9)@45; boolean.rs:225-239 (7,4)@28; cfg3 (6,9)@54; example (11,5)@55);
    uniform-random key rows, ragged batch, two key entries"""
    n = 1 << log_n
    q = cref.two_adic_primes(bits, log_n + 1, 1)[0]
    ctx = fhe.NttContext(q)
    batch, count = 7, 2
    ra, rb = rand_u64(1, q, (count, 2 * d, n)), rand_u64(2, q, (count, 2 * d, n))
    ka, kb = rand_u64(3, q, (count, d, n)), rand_u64(4, q, (count, d, n))
    ca, cb = rand_u64(5, q, (batch, n)), rand_u64(6, q, (batch, n))
    ca[0, :4] = [0, q - 1, q >> 1, (q >> 1) + 1]
    rgsw = fhe.GadgetKey(ctx, log_b, d, dev(torch_cuda, ra), dev(torch_cuda, rb), n, rgsw=True)
    ksk = fhe.GadgetKey(ctx, log_b, d, dev(torch_cuda, ka), dev(torch_cuda, kb), n, rgsw=False)
    for idx in range(count):
        a, b = dev(torch_cuda, ca), dev(torch_cuda, cb)
        rgsw.external_product_(idx, a, b)
        ha, hb = host(a), host(b)
        for i in range(batch):
            ea, eb = cref.external_product(q, log_b, d, ra[idx], rb[idx], ca[i], cb[i])
            assert np.array_equal(ha[i], ea) and np.array_equal(hb[i], eb), (idx, i)
        a, b = dev(torch_cuda, ca), dev(torch_cuda, cb)
        ksk.key_switch_(idx, a, b)
        ha, hb = host(a), host(b)
        for i in range(0, batch, 3):
            ea, eb = cref.rlwe_key_switch(q, log_b, d, ka[idx], kb[idx], ca[i], cb[i])
            assert np.array_equal(ha[i], ea) and np.array_equal(hb[i], eb)
        for t in (5, -5, 25):
            a, b = dev(torch_cuda, ca), dev(torch_cuda, cb)
            ksk.automorphism_(idx, t, a, b)
            ha, hb = host(a), host(b)
            ea, eb = cref.rlwe_automorphism(q, log_b, d, t, ka[idx], kb[idx], ca[1], cb[1])
            assert np.array_equal(ha[1], ea) and np.array_equal(hb[1], eb)


def test_external_product_decrypt_level(fhe, torch_cuda):
    """scheme/fhew/src/rgsw.rs:198-211 on the GPU path: decrypt(rgsw(m0) [x] rlwe(m1)) == m0 * m1 (valid keys from the
    oracle's key generator), and rlwe.rs:401-415 for the automorphism."""
    from oracle import pyref as P
    rnd = random.Random(21)
    log_n, p, log_b, d = 7, 16, 5, 9
    n = 1 << log_n
    q = next(P.two_adic_primes(45, log_n + 1))
    dec = P.Base2Decomposor(q, log_b, d)
    delta = q / p
    enc = lambda m: [P.zq_from_f64(q, float(x) * delta) for x in m]  # noqa: E731
    decd = lambda pt: [P.zq_from_f64(p, float(P.zq_to_i64(q, x)) / delta) for x in pt]  # noqa: E731
    sk = [rnd.randint(-3, 3) for _ in range(n)]
    m0, m1 = [rnd.randrange(p) for _ in range(n)], [rnd.randrange(p) for _ in range(n)]
    ra, rb = P.rgsw_encrypt(q, dec, sk, m0, rnd)
    ca, cb = P.rlwe_sk_encrypt(q, sk, enc(m1), rnd)
    ctx = fhe.NttContext(q)
    U = lambda x: np.array(x, dtype=np.uint64)  # noqa: E731
    rgsw = fhe.GadgetKey(ctx, log_b, d, U(ra), U(rb), n, rgsw=True)
    a, b = dev(torch_cuda, U(ca)), dev(torch_cuda, U(cb))
    rgsw.external_product_(0, a, b)
    assert decd(P.rlwe_decrypt(q, sk, L(host(a)), L(host(b)))) == P.nega_cyclic_schoolbook_mul(p, m0, m1)
    for t in (5, -5):
        ka, kb = P.rlwe_ak_gen(q, dec, t, sk, rnd)
        ak = fhe.GadgetKey(ctx, log_b, d, U(ka), U(kb), n, rgsw=False)
        a, b = dev(torch_cuda, U(ca)), dev(torch_cuda, U(cb))
        ak.automorphism_(0, t, a, b)
        assert decd(P.rlwe_decrypt(q, sk, L(host(a)), L(host(b)))) == P.automorphism(p, m1, t)


def test_rgsw_internal_product(fhe, torch_cuda):
    """scheme/fhew/src/rgsw.rs:130-150 `Rgsw::internal_product` through fhe_rgsw_internal_product: (i) every row of the product
    bit-equal to the oracle's evaluation-domain restatement, for two right-hand ciphertexts in one call and two key entries;
    (ii) the reference's own test (rgsw.rs:214-227, its parameters log_q 45, p 16, log_b 5, d 9): decrypt(ct0 x ct1) == m0 * m1."""
    from oracle import pyref as P
    rnd = random.Random(23)
    log_n, p, log_b, d = 7, 16, 5, 9
    n = 1 << log_n
    q = next(P.two_adic_primes(45, log_n + 1))
    dec = P.Base2Decomposor(q, log_b, d)
    sk = [rnd.randint(-3, 3) for _ in range(n)]
    ms = [[rnd.randrange(p) for _ in range(n)] for _ in range(4)]
    cts = [P.rgsw_encrypt(q, dec, sk, m, rnd) for m in ms]          # (rows_a, rows_b), each [2d][n]
    U = lambda x: np.array(x, dtype=np.uint64)  # noqa: E731
    ctx = fhe.NttContext(q)
    key = fhe.GadgetKey(ctx, log_b, d, dev(torch_cuda, U([cts[0][0], cts[1][0]])), dev(torch_cuda, U([cts[0][1], cts[1][1]])), n, rgsw=True)
    last_base_bits = dec.rounding_bits + (d - 1) * log_b
    for idx in (0, 1):
        a, b = dev(torch_cuda, U([cts[2][0], cts[3][0]])), dev(torch_cuda, U([cts[2][1], cts[3][1]]))
        key.internal_product_(idx, a, b)
        ha, hb = host(a).reshape(2, 2 * d, n), host(b).reshape(2, 2 * d, n)
        for r, rhs in enumerate((2, 3)):
            ea, eb = P.rgsw_internal_product(q, dec, cts[idx][0], cts[idx][1], cts[rhs][0], cts[rhs][1])
            assert ha[r].tolist() == ea and hb[r].tolist() == eb, (idx, rhs)
            # Rgsw::decrypt (rgsw.rs:107-114): phase of the LAST row, rounding_shr by the last base, then mod p
            phase = P.rlwe_decrypt(q, sk, L(ha[r][-1]), L(hb[r][-1]))
            got = [(((v + ((1 << last_base_bits) >> 1)) % q) >> last_base_bits) % p for v in phase]
            assert got == P.nega_cyclic_schoolbook_mul(p, ms[idx], ms[rhs]), (idx, rhs)
    with pytest.raises(fhe.FheError):  # a key-switching key is not an RGSW ciphertext
        ksk = fhe.GadgetKey(ctx, log_b, d, U(cts[0][0][:d]), U(cts[0][1][:d]), n, rgsw=False)
        ksk.internal_product_(0, dev(torch_cuda, U(cts[2][0])), dev(torch_cuda, U(cts[2][1])))


def _make_bk(fhe, torch_cuda, q, n, log_b, d, ks_log_b, ks_d, w, n_lwe, seed):
    from oracle import pyref as P
    brk = rand_u64(seed, q, (n_lwe, 2, 2 * d, n))          # [key][a|b][row][n] as the oracle takes it
    ak = rand_u64(seed + 1, q, (w + 1, 2, ks_d, n))
    ctx = fhe.NttContext(q)
    gk = fhe.GadgetKey(ctx, log_b, d, dev(torch_cuda, brk[:, 0]), dev(torch_cuda, brk[:, 1]), n, rgsw=True)
    ga = fhe.GadgetKey(ctx, ks_log_b, ks_d, dev(torch_cuda, ak[:, 0]), dev(torch_cuda, ak[:, 1]), n, rgsw=False)
    ts = P.ak_t(n, w)
    return ctx, fhe.BootstrapKey(ctx, gk, ga, ts, w), brk, ak, ts


@pytest.mark.parametrize("q,log_n", [(18014398509404161, 10), (18014398509404161, 11), (35184372060161, 10), (35184372060161, 11)])
def test_throughput_and_small_batch_shapes_agree(fhe, cref, torch_cuda, q, log_n):
    """N >= 1024 has two instantiations of the fused kernels (8 coefficients per lane above 512 ciphertexts, 4 below, each with
    its own key layout; at N = 1024 the 8-per-lane form serves batches of 769 .. 1024 only, fhew_api.hip `small_shape`): the same ciphertext must come out bit-identical from both, and equal to the oracle"""
    n, lb, d, w, n_lwe, big = 1 << log_n, 6, 3, 3, 3, 800  # 769 .. 1024: the batches that run the 8-per-lane form at N = 1024
    assert cref.is_prime(q) and (q - 1) % (2 * n) == 0
    ctx, bk, brk, ak, ts = _make_bk(fhe, torch_cuda, q, n, lb, d, 5, 4, w, n_lwe, seed=90 + log_n)
    gk, ga = bk.brk, bk.ak
    ca, cb = rand_u64(91, q, (big, n)), rand_u64(92, q, (big, n))
    for kind in ("ep", "ks", "auto"):
        outs = []
        for cnt in (big, 3):
            a, b = dev(torch_cuda, ca[:cnt]), dev(torch_cuda, cb[:cnt])
            if kind == "ep":
                gk.external_product_(1, a, b)
            elif kind == "ks":
                ga.key_switch_(2, a, b)
            else:
                ga.automorphism_(1, ts[1], a, b)
            outs.append((host(a), host(b)))
        assert np.array_equal(outs[0][0][:3], outs[1][0]) and np.array_equal(outs[0][1][:3], outs[1][1]), kind
        assert outs[0][0].shape == (big, n) and not np.array_equal(outs[0][0][big - 1], ca[big - 1]), kind  # the large batch ran to its end
        if kind == "ep":
            ea, eb = cref.external_product(q, lb, d, brk[1, 0], brk[1, 1], ca[big - 1], cb[big - 1])
            assert np.array_equal(outs[0][0][big - 1], ea) and np.array_equal(outs[0][1][big - 1], eb)
    rng = np.random.Generator(np.random.PCG64(93))
    lwe_a = (rng.integers(0, n, size=(big, n_lwe), dtype=np.uint64) * 2 + 1)
    lwe_b = rng.integers(0, 2 * n, size=big, dtype=np.uint64)
    f = rand_u64(94, q, n)
    oa, ob = bk.blind_rotate(dev(torch_cuda, lwe_a), dev(torch_cuda, lwe_b), dev(torch_cuda, f))
    sa, sb = bk.blind_rotate(dev(torch_cuda, lwe_a[:2]), dev(torch_cuda, lwe_b[:2]), dev(torch_cuda, f))
    assert np.array_equal(host(oa)[:2], host(sa)) and np.array_equal(host(ob)[:2], host(sb))
    ea, eb = cref.blind_rotate(q, n, w, lb, d, 5, 4, brk, ak, ts, f, lwe_a[big - 1], int(lwe_b[big - 1]))
    assert np.array_equal(host(oa)[big - 1], ea) and np.array_equal(host(ob)[big - 1], eb)


@pytest.mark.parametrize("batch", [1024, 1600])
def test_blind_rotate_large_batch_is_deterministic(fhe, cref, torch_cuda, batch):
    """cfg3 ring, 1024 ciphertexts (every SIMD busy with two-wave teams sharing LDS images) and 1600 (the 4-per-lane form over
    several generations of workgroups): two runs agree bit for bit, a sample agrees with the small-batch instantiation and one
    ciphertext with the oracle"""
    q, n, lb, d, w, n_lwe = 18014398509404161, 1024, 6, 9, 10, 24
    ctx, bk, brk, ak, ts = _make_bk(fhe, torch_cuda, q, n, lb, d, lb, d, w, n_lwe, seed=110)
    rng = np.random.Generator(np.random.PCG64(111))
    lwe_a = (rng.integers(0, n, size=(batch, n_lwe), dtype=np.uint64) * 2 + 1)
    lwe_b = rng.integers(0, 2 * n, size=batch, dtype=np.uint64)
    f = rand_u64(112, q, n)
    da, db, df = dev(torch_cuda, lwe_a), dev(torch_cuda, lwe_b), dev(torch_cuda, f)
    o1 = bk.blind_rotate(da, db, df)
    o2 = bk.blind_rotate(da, db, df)
    assert torch_cuda.equal(o1[0], o2[0]) and torch_cuda.equal(o1[1], o2[1])
    pick = [0, 511, batch - 1]
    s = bk.blind_rotate(dev(torch_cuda, lwe_a[pick]), dev(torch_cuda, lwe_b[pick]), df)
    assert np.array_equal(host(o1[0])[pick], host(s[0])) and np.array_equal(host(o1[1])[pick], host(s[1]))
    ea, eb = cref.blind_rotate(q, n, w, lb, d, lb, d, brk, ak, ts, f, lwe_a[777], int(lwe_b[777]))
    assert np.array_equal(host(o1[0])[777], ea) and np.array_equal(host(o1[1])[777], eb)


def test_blind_rotate_golden(fhe, torch_cuda):
    v = load_golden("blind_rotate.json")
    q, n, w, lb, d = v["q"], v["n"], v["w"], v["log_b"], v["d"]
    U = lambda x: np.array(x, dtype=np.uint64)  # noqa: E731
    brk, ak = U(v["brk"]), U(v["ak"])  # [key][a|b][row][n]
    ctx = fhe.NttContext(q)
    gk = fhe.GadgetKey(ctx, lb, d, dev(torch_cuda, brk[:, 0]), dev(torch_cuda, brk[:, 1]), n, rgsw=True)
    ga = fhe.GadgetKey(ctx, lb, d, dev(torch_cuda, ak[:, 0]), dev(torch_cuda, ak[:, 1]), n, rgsw=False)
    bk = fhe.BootstrapKey(ctx, gk, ga, v["ak_t"], w)
    oa, ob, sched = bk.blind_rotate(dev(torch_cuda, U([v["lwe_a"]])), dev(torch_cuda, U([v["lwe_b"]])), dev(torch_cuda, U(v["f"])),
                                    want_schedule=True)
    assert sched[0] == [(k, i) for k, i in v["schedule"]]
    assert L(host(oa)) == v["out_a"] and L(host(ob)) == v["out_b"]


def test_blind_rotate_vs_oracle_small(fhe, cref, torch_cuda):
    """N = 128, n_lwe = 6, w = 3: batch of 6 with zero coefficients, repeated buckets, per-ciphertext LUTs"""
    q, n, lb, d, w, n_lwe, batch = 18014398509404161, 128, 6, 3, 3, 6, 6
    ctx, bk, brk, ak, ts = _make_bk(fhe, torch_cuda, q, n, lb, d, 5, 4, w, n_lwe, seed=50)
    rng = np.random.Generator(np.random.PCG64(9))
    lwe_a = (rng.integers(0, n, size=(batch, n_lwe), dtype=np.uint64) * 2 + 1)
    lwe_a[1, 2] = 0
    lwe_a[2, :] = lwe_a[2, 0]          # all in one bucket
    lwe_a[3, :] = 0                    # nothing to rotate by: only the automorphism walk
    lwe_a[4, :3] = 2 * n - lwe_a[4, 3:6]  # +- pairs
    lwe_b = rng.integers(0, 2 * n, size=batch, dtype=np.uint64)
    f = rand_u64(77, q, (batch, n))
    oa, ob, sched = bk.blind_rotate(dev(torch_cuda, lwe_a), dev(torch_cuda, lwe_b), dev(torch_cuda, f), want_schedule=True)
    ha, hb = host(oa), host(ob)
    for i in range(batch):
        assert sched[i] == cref.blind_rotate_schedule(n, w, lwe_a[i]), i
        ea, eb = cref.blind_rotate(q, n, w, lb, d, 5, 4, brk, ak, ts, f[i], lwe_a[i], int(lwe_b[i]))
        assert np.array_equal(ha[i], ea) and np.array_equal(hb[i], eb), i
    # shared LUT
    oa2, ob2 = bk.blind_rotate(dev(torch_cuda, lwe_a[:2]), dev(torch_cuda, lwe_b[:2]), dev(torch_cuda, f[0]))
    assert np.array_equal(host(oa2)[0], ha[0]) and np.array_equal(host(ob2)[0], hb[0])
    # an even (non-zero) LWE coefficient is `unreachable!()` in the reference (bootstrapping.rs:221)
    bad = lwe_a.copy()
    bad[0, 0] = 4
    bk.check(dev(torch_cuda, lwe_b))                       # nothing wrong so far
    with pytest.raises(fhe.FheError):                      # host memory: the call itself reports it
        bk.blind_rotate(bad, lwe_b, f)
    d_bad = dev(torch_cuda, bad)
    bk.blind_rotate(d_bad, dev(torch_cuda, lwe_b), dev(torch_cuda, f))  # device memory: asynchronous, the key's status word records it
    with pytest.raises(fhe.FheError):
        bk.check(d_bad)
    bk.check(d_bad)                                        # cleared by the previous query


@pytest.mark.parametrize("w", [1, 2, 5, 31])
def test_blind_rotate_schedule_random(fhe, cref, torch_cuda, w):
    """the device-side LMKCDEY walk (sorted-run restatement) against the oracle's level-by-level loop, many random LWE vectors:
    dense and sparse buckets, zeros, every window size w up to N/4 - 1"""
    q, n, lb, d, n_lwe, batch = 18014398509404161, 128, 6, 3, 40, 48
    ctx, bk, brk, ak, ts = _make_bk(fhe, torch_cuda, q, n, lb, d, 5, 4, w, n_lwe, seed=80 + w)
    rng = np.random.Generator(np.random.PCG64(100 + w))
    lwe_a = (rng.integers(0, n, size=(batch, n_lwe), dtype=np.uint64) * 2 + 1)
    lwe_a[rng.random(size=lwe_a.shape) < 0.2] = 0          # rotations by zero are skipped
    lwe_a[1, :] = 1                                          # everything in bucket (plus, 0)
    lwe_a[2, :] = 2 * n - 1                                  # everything in bucket (minus, 0)
    lwe_a[3, :] = 0
    lwe_a[4, :] = 5                                          # (plus, 1): the level next to the end of the walk
    lwe_a[5, :20] = pow(5, n // 2 - 1, 2 * n)                # (plus, top level)
    lwe_a[5, 20:] = 2 * n - pow(5, n // 2 - 1, 2 * n)        # (minus, top level)
    lwe_b = rng.integers(0, 2 * n, size=batch, dtype=np.uint64)
    f = rand_u64(79, q, n)
    oa, ob, sched = bk.blind_rotate(dev(torch_cuda, lwe_a), dev(torch_cuda, lwe_b), dev(torch_cuda, f), want_schedule=True)
    for i in range(batch):
        assert sched[i] == cref.blind_rotate_schedule(n, w, lwe_a[i]), (w, i)
    for i in (0, 5):
        ea, eb = cref.blind_rotate(q, n, w, lb, d, 5, 4, brk, ak, ts, f, lwe_a[i], int(lwe_b[i]))
        assert np.array_equal(host(oa)[i], ea) and np.array_equal(host(ob)[i], eb), i


def test_blind_rotate_cfg3(fhe, cref, torch_cuda):
    """BASELINE config 3: N = 2^10, q = 18014398509404161, base 2^6, d = 9, n = 100, w = 10 -- the full CMUX loop,
    bit-exact against the oracle on 2 ciphertexts (uniform-random keys: validity is irrelevant for parity)."""
    q, n, lb, d, w, n_lwe, batch = 18014398509404161, 1024, 6, 9, 10, 100, 2
    ctx, bk, brk, ak, ts = _make_bk(fhe, torch_cuda, q, n, lb, d, lb, d, w, n_lwe, seed=60)
    rng = np.random.Generator(np.random.PCG64(10))
    lwe_a = (rng.integers(0, n, size=(batch, n_lwe), dtype=np.uint64) * 2 + 1)
    lwe_b = rng.integers(0, 2 * n, size=batch, dtype=np.uint64)
    f = rand_u64(78, q, n)
    oa, ob, sched = bk.blind_rotate(dev(torch_cuda, lwe_a), dev(torch_cuda, lwe_b), dev(torch_cuda, f), want_schedule=True)
    for i in range(batch):
        assert sched[i] == cref.blind_rotate_schedule(n, w, lwe_a[i])
        n_ep = sum(1 for k, _ in sched[i] if k == "ep")
        assert n_ep == n_lwe and len(sched[i]) > 200
        ea, eb = cref.blind_rotate(q, n, w, lb, d, lb, d, brk, ak, ts, f, lwe_a[i], int(lwe_b[i]))
        assert np.array_equal(host(oa)[i], ea) and np.array_equal(host(ob)[i], eb), i


def test_blind_rotate_decrypt_level(fhe, torch_cuda):
    """valid keys: the constant term of the rotated accumulator decrypts to f at the LWE phase"""
    from oracle import pyref as P
    rnd = random.Random(31)
    log_n, log_b, d, w, n_lwe = 7, 5, 9, 3, 8
    n = 1 << log_n
    q = next(P.two_adic_primes(45, log_n + 1))
    dec = P.Base2Decomposor(q, log_b, d)
    z = [rnd.randint(-1, 1) for _ in range(n)]
    s = [rnd.randint(-1, 1) for _ in range(n_lwe)]
    one = [1] + [0] * (n - 1)
    brk = [P.rgsw_encrypt(q, dec, z, P.monomial_mul(q, one, sj), rnd) for sj in s]
    ts = P.ak_t(n, w)
    ak = [P.rlwe_ak_gen(q, dec, t, z, rnd) for t in ts]
    U = lambda x: np.array(x, dtype=np.uint64)  # noqa: E731
    ctx = fhe.NttContext(q)
    gk = fhe.GadgetKey(ctx, log_b, d, U([k[0] for k in brk]), U([k[1] for k in brk]), n, rgsw=True)
    ga = fhe.GadgetKey(ctx, log_b, d, U([k[0] for k in ak]), U([k[1] for k in ak]), n, rgsw=False)
    bk = fhe.BootstrapKey(ctx, gk, ga, ts, w)
    f = [rnd.randrange(q >> 4) << 3 for _ in range(n)]
    batch = 5
    a = [[rnd.randrange(n) * 2 + 1 for _ in range(n_lwe)] for _ in range(batch)]
    b = [rnd.randrange(2 * n) for _ in range(batch)]
    oa, ob = bk.blind_rotate(dev(torch_cuda, U(a)), dev(torch_cuda, U(b)), dev(torch_cuda, U(f)))
    for i in range(batch):
        mu = (b[i] - sum(x * y for x, y in zip(a[i], s))) % (2 * n)
        pt = P.rlwe_decrypt(q, z, L(host(oa)[i]), L(host(ob)[i]))
        exp = f[mu] if mu < n else P.zq_neg(q, f[mu - n])
        assert abs(P.zq_to_i64(q, (pt[0] - exp) % q)) < (1 << 30)


# ---- the LWE side of the gate and the whole gate bootstrap ---------------------------------


def test_lwe_mod_switch(fhe, cref, torch_cuda):
    """util/src/zq.rs:128-140 through scheme/fhew/src/lwe.rs:90-99: the f64 rule, bit for bit"""
    rng = np.random.Generator(np.random.PCG64(5))
    for q, qp, odd in [(268369921, 1 << 16, False), (1 << 16, 1024, True), (18014398509404161, 1 << 16, False), (1 << 16, 2048, True),
                       (1152921504606748673, 1 << 20, False), (1 << 20, 4096, True), (1073707009, 12289, False)]:
        v = rng.integers(0, q, size=3000, dtype=np.uint64)
        v[:8] = [0, 1, 2, q - 1, q - 2, q // 2, q // 2 + 1, q // 3]
        if odd:
            v[8:40] = np.arange(32, dtype=np.uint64)          # floor(x) == 0 branch and its neighbours
        out = host(fhe.lwe_mod_switch(q, qp, dev(torch_cuda, v), odd=odd))
        ref = cref.mod_switch_odd if odd else cref.mod_switch
        assert L(out) == [ref(q, int(x), qp) for x in v], (q, qp, odd)
    v = rng.integers(0, 1 << 16, size=37, dtype=np.uint64)
    assert L(fhe.lwe_mod_switch(1 << 16, 1024, v, odd=True)) == [cref.mod_switch_odd(1 << 16, int(x), 1024) for x in v]  # host memory
    assert fhe.lwe_mod_switch(1 << 16, 1024, np.zeros(0, dtype=np.uint64)).size == 0


def test_lwe_key_switch_and_sample_extract(fhe, cref, torch_cuda):
    """scheme/fhew/src/lwe.rs:151-160 and rlwe.rs:193-202"""
    for q, lb, d, n_in, n_out, batch in [(1 << 16, 4, 4, 128, 20, 5), (1 << 20, 5, 4, 64, 33, 3), (12289, 3, 4, 32, 7, 2)]:
        ksk_a, ksk_b = rand_u64(1, q, (d * n_in, n_out)), rand_u64(2, q, d * n_in)
        ct_a, ct_b = rand_u64(3, q, (batch, n_in)), rand_u64(4, q, batch)
        ct_a[0, :4] = [0, q - 1, q // 2, 1]
        oa, ob = fhe.lwe_key_switch(q, lb, d, dev(torch_cuda, ksk_a), dev(torch_cuda, ksk_b), dev(torch_cuda, ct_a), dev(torch_cuda, ct_b),
                                    n_in, n_out)
        for i in range(batch):
            ea, eb = cref.lwe_key_switch(q, lb, d, ksk_a, ksk_b, ct_a[i], int(ct_b[i]))
            assert np.array_equal(host(oa)[i], ea) and int(host(ob)[i]) == eb, (q, i)
    with pytest.raises(fhe.FheError):
        fhe.lwe_key_switch(1 << 40, 4, 4, ksk_a, ksk_b, ct_a, ct_b, n_in, n_out)  # q >= 2^32 is outside this entry's range
    # ragged batches that take the 2- and 4-ciphertext tiles of the tiled kernel, more output columns than threads of a block
    for q, lb, d, n_in, n_out, batch in [(1 << 16, 4, 4, 32, 7, 515), (12289, 3, 4, 16, 150, 2050), (1 << 20, 5, 4, 8, 3, 2049)]:
        ksk_a, ksk_b = rand_u64(21, q, (d * n_in, n_out)), rand_u64(22, q, d * n_in)
        ct_a, ct_b = rand_u64(23, q, (batch, n_in)), rand_u64(24, q, batch)
        oa, ob = fhe.lwe_key_switch(q, lb, d, dev(torch_cuda, ksk_a), dev(torch_cuda, ksk_b), dev(torch_cuda, ct_a), dev(torch_cuda, ct_b),
                                    n_in, n_out)
        ha, hb = host(oa), host(ob)
        for i in list(range(0, batch, 37)) + [batch - 2, batch - 1]:
            ea, eb = cref.lwe_key_switch(q, lb, d, ksk_a, ksk_b, ct_a[i], int(ct_b[i]))
            assert np.array_equal(ha[i], ea) and int(hb[i]) == eb, (q, i)
    q, n, batch = 18014398509404161, 256, 4
    a, b = rand_u64(5, q, (batch, n)), rand_u64(6, q, (batch, n))
    a[0, :3] = 0
    for idx in (0, 1, 100, n - 1):
        oa, ob = fhe.rlwe_sample_extract(q, dev(torch_cuda, a), dev(torch_cuda, b), n, idx)
        for i in range(batch):
            ea, eb = cref.sample_extract(q, a[i], b[i], idx)
            assert np.array_equal(host(oa)[i], ea) and int(host(ob)[i]) == eb
    oa, ob = fhe.rlwe_sample_extract(q, a, b, n, 0, addend=q - 1)
    assert L(ob) == [(int(x) + q - 1) % q for x in b[:, 0]]


def test_lwe_lincomb(fhe, torch_cuda):
    q = 18014398509404161
    xs = [rand_u64(10 + i, q, (3, 50)) for i in range(3)]
    xs[0][0, :2] = [0, q - 1]
    for coefs, addend in [([1, 1], 0), ([2, -2], 0), ([-1], q // 4), ([1, 1, 1], 5), ([7, -3, 2], q - 1)]:
        out = fhe.lwe_lincomb(q, coefs, [dev(torch_cuda, x) for x in xs[:len(coefs)]], addend)
        exp = [(sum(c * int(x.ravel()[i]) for c, x in zip(coefs, xs)) + addend) % q for i in range(150)]
        assert L(host(out)) == exp, coefs


def test_fhew_bootstrap_vs_oracle(fhe, cref, torch_cuda):
    """scheme/fhew/src/bootstrapping.rs:149-155, the whole chain on uniform-random keys, bit-exact against the oracle's steps"""
    q, n, lb, d, w, n_lwe, batch = 18014398509404161, 128, 6, 3, 3, 6, 4
    q_ks, kb, kd = 1 << 16, 4, 4
    ctx, bk, brk, ak, ts = _make_bk(fhe, torch_cuda, q, n, lb, d, 5, 4, w, n_lwe, seed=70)
    ksk_a, ksk_b = rand_u64(71, q_ks, (kd * n, n_lwe)), rand_u64(72, q_ks, kd * n)
    ct_a, ct_b = rand_u64(73, q, (batch, n)), rand_u64(74, q, batch)
    f = rand_u64(75, q, n)
    addend = q // 8
    oa, ob = bk.bootstrap(q_ks, kb, kd, dev(torch_cuda, ksk_a), dev(torch_cuda, ksk_b), dev(torch_cuda, f), dev(torch_cuda, ct_a),
                          dev(torch_cuda, ct_b), addend=addend)
    for i in range(batch):
        a1 = np.array([cref.mod_switch(q, int(x), q_ks) for x in ct_a[i]], dtype=np.uint64)
        b1 = cref.mod_switch(q, int(ct_b[i]), q_ks)
        a2, b2 = cref.lwe_key_switch(q_ks, kb, kd, ksk_a, ksk_b, a1, b1)
        a3 = np.array([cref.mod_switch_odd(q_ks, int(x), 2 * n) for x in a2], dtype=np.uint64)
        b3 = cref.mod_switch_odd(q_ks, int(b2), 2 * n)
        ra, rb = cref.blind_rotate(q, n, w, lb, d, 5, 4, brk, ak, ts, f, a3, b3)
        ea, eb = cref.sample_extract(q, ra, rb, 0)
        assert np.array_equal(host(oa)[i], ea) and int(host(ob)[i]) == (eb + addend) % q, i
    # host-memory entry
    ha, hb = bk.bootstrap(q_ks, kb, kd, ksk_a, ksk_b, f, ct_a[:1], ct_b[:1], addend=addend)
    assert np.array_equal(ha, host(oa)[:1]) and np.array_equal(hb, host(ob)[:1])


def test_fhew_gates_decrypt(fhe, torch_cuda):
    """The reference's own gate test (scheme/fhew/src/fhew/boolean.rs:256-290) with its `single_key_testing_param`
    (boolean.rs:225-239): every gate on every input combination decrypts to the truth table."""
    from oracle import pyref as P
    rnd = random.Random(2024)
    log_q, log_n, log_b, d, w = 28, 9, 7, 4, 10
    n_lwe, q_ks, kb, kd = 100, 1 << 16, 4, 4
    n = 1 << log_n
    q = next(P.two_adic_primes(log_q, log_n + 1))
    U = lambda x: np.array(x, dtype=np.uint64)  # noqa: E731
    dec, dec_ks = P.Base2Decomposor(q, log_b, d), P.Base2Decomposor(q_ks, kb, kd)
    z = [rnd.randint(-1, 1) for _ in range(n)]
    s = [rnd.randint(-1, 1) for _ in range(n_lwe)]
    one = [1] + [0] * (n - 1)
    brk = [P.rgsw_encrypt(q, dec, z, P.monomial_mul(q, one, sj), rnd) for sj in s]
    ts = P.ak_t(n, w)
    ak = [P.rlwe_ak_gen(q, dec, t, z, rnd) for t in ts]
    ksk_a, ksk_b = P.lwe_ksk_gen(q_ks, dec_ks, s, z, rnd)
    ctx = fhe.NttContext(q)
    gk = fhe.GadgetKey(ctx, log_b, d, U([k[0] for k in brk]), U([k[1] for k in brk]), n, rgsw=True)
    ga = fhe.GadgetKey(ctx, log_b, d, U([k[0] for k in ak]), U([k[1] for k in ak]), n, rgsw=False)
    bk = fhe.BootstrapKey(ctx, gk, ga, ts, w)
    ev = fhe.Fhew(bk, q_ks, kb, kd, dev(torch_cuda, U(ksk_a)), dev(torch_cuda, U(ksk_b)))
    delta = q / 4.0

    def encrypt(bits):
        cts = [P.lwe_sk_encrypt(q, z, P.zq_from_f64(q, float(m) * delta), rnd) for m in bits]
        return dev(torch_cuda, U([c[0] for c in cts])), dev(torch_cuda, U([c[1] for c in cts]))

    def decrypt(ct):
        a, b = host(ct[0]), host(ct[1])
        out = []
        for i in range(a.shape[0]):
            pt = P.lwe_decrypt(q, z, L(a[i]), int(b[i]))
            m = P.zq_from_f64(4, float(pt) / delta)
            assert m in (0, 1), m
            out.append(m)
        return out

    m0 = [(m >> 0) & 1 for m in range(4)]
    m1 = [(m >> 1) & 1 for m in range(4)]
    c0, c1 = encrypt(m0), encrypt(m1)
    assert decrypt(c0) == m0 and decrypt(ev.not_(c0)) == [1 - x for x in m0]
    assert decrypt(ev.and_(c0, c1)) == [x & y for x, y in zip(m0, m1)]
    assert decrypt(ev.nand(c0, c1)) == [1 - (x & y) for x, y in zip(m0, m1)]
    assert decrypt(ev.or_(c0, c1)) == [x | y for x, y in zip(m0, m1)]
    assert decrypt(ev.nor(c0, c1)) == [1 - (x | y) for x, y in zip(m0, m1)]
    assert decrypt(ev.xor(c0, c1)) == [x ^ y for x, y in zip(m0, m1)]
    assert decrypt(ev.xnor(c0, c1)) == [1 - (x ^ y) for x, y in zip(m0, m1)]
    t0 = [(m >> 0) & 1 for m in range(8)]
    t1 = [(m >> 1) & 1 for m in range(8)]
    t2 = [(m >> 2) & 1 for m in range(8)]
    assert decrypt(ev.majority(encrypt(t0), encrypt(t1), encrypt(t2))) == [(x & y) | (y & u) | (u & x) for x, y, u in zip(t0, t1, t2)]
    # a two-level circuit: outputs of one gate are valid inputs of the next (noise refreshed)
    assert decrypt(ev.xor(ev.nand(c0, c1), ev.or_(c0, c1))) == [(1 - (x & y)) ^ (x | y) for x, y in zip(m0, m1)]


def test_blind_rotate_is_asynchronous_on_device_memory(fhe, torch_cuda):
    """Device-memory blind rotations return when their work is enqueued (the data-dependent input check goes to the key's status
    word, fhe_bootstrap_key_status): the call's host time is a fraction of the kernel's, and two streams overlap -- two batch-64
    blind rotations (64 of 256 CUs each) on two streams finish in well under twice the time of one."""
    import time
    torch = torch_cuda
    q, n, lb, d, w, n_lwe, batch = 18014398509404161, 1024, 6, 9, 10, 100, 64
    ctx, bk, brk, ak, ts = _make_bk(fhe, torch_cuda, q, n, lb, d, lb, d, w, n_lwe, seed=160)
    rng = np.random.Generator(np.random.PCG64(161))
    lwe_a = dev(torch, rng.integers(0, n, size=(batch, n_lwe), dtype=np.uint64) * 2 + 1)
    lwe_b = dev(torch, rng.integers(0, 2 * n, size=batch, dtype=np.uint64))
    f = dev(torch, rand_u64(162, q, n))
    ref_a, ref_b = bk.blind_rotate(lwe_a, lwe_b, f)  # warm-up (module load, LDS attribute)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    oa, ob = bk.blind_rotate(lwe_a, lwe_b, f)
    t_call = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_one = time.perf_counter() - t0
    assert t_call < 0.5 * t_one, (t_call, t_one)           # the call returned long before the GPU finished
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for s_ in (s1, s2):  # first use of a stream: its creation and its first workspace from the pool are not what is measured
        with torch.cuda.stream(s_):
            bk.blind_rotate(lwe_a, lwe_b, f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        o1 = bk.blind_rotate(lwe_a, lwe_b, f)
    with torch.cuda.stream(s2):
        o2 = bk.blind_rotate(lwe_a, lwe_b, f)
    torch.cuda.synchronize()
    t_two = time.perf_counter() - t0
    assert t_two < 1.6 * t_one, (t_two, t_one)
    for o in (o1, o2, (oa, ob)):
        assert torch.equal(o[0], ref_a) and torch.equal(o[1], ref_b)
    bk.check(lwe_a)                                        # every input was an odd residue: status clean
