"""GPU parity tests for SURVEY.md section 8(a) row T (TFHE torus path, k = 1) through the C ABI.
Acceptance rule of row T: (i) decode-level equality, (ii) per-coefficient distance to the exact negacyclic product mod 2^64
within the reference's own bound (util/src/ring/fft/c64.rs:186-208).  The GPU path is exact, so (ii) is asserted as
equality with the exact oracle (oracle/pyref.py), which is inside the reference's bound by definition."""
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


def test_torus_decompose(fhe, torch_cuda):
    from oracle import pyref as P
    rnd = random.Random(1)
    for log_b, d in [(23, 1), (4, 5), (8, 3), (16, 4), (7, 9), (32, 2)]:
        dec = P.TorusDecomposor(log_b, d)
        v = [0, 1, P.M64 - 1, 1 << 63, (1 << 63) - 1, (1 << 40) - 1] + [rnd.getrandbits(64) for _ in range(58)]
        out = host(fhe.torus_decompose(log_b, d, dev(torch_cuda, U(v)), len(v)))
        assert out.reshape(d, len(v)).tolist() == dec.decompose(v)


@pytest.mark.parametrize("n", [2, 64, 1024, 4096])
def test_torus_mul_exact(fhe, torch_cuda, n):
    """`Rt * Rt` (ring.rs:315-320): exact product, hence within the reference's precision bound (c64.rs:186-208) for the
    operand sizes that test uses (b up to 2^17)"""
    from oracle import pyref as P
    rnd = random.Random(n)
    t = fhe.TorusContext()
    for log_bound in (1, 12, 17, 23):
        a = [rnd.getrandbits(64) for _ in range(n)]
        b = [rnd.randint(-(1 << log_bound) + 1, (1 << log_bound) - 1) % P.M64 for _ in range(n)]
        da, db = dev(torch_cuda, U(a)), dev(torch_cuda, U(b))
        t.mul_(da, db, log_bound, n)
        if n <= 1024:
            assert L(host(da)) == P.torus_mul_exact(a, b), (n, log_bound)
        else:  # full schoolbook is slow in Python: check 8 output coefficients
            sa, sb = [P.t64_to_i64(x) for x in a], [P.t64_to_i64(x) for x in b]
            got = host(da)
            for k in (0, 1, n // 2, n - 1, 17, 1000, 2049, 4000):
                exp = sum(sa[i] * sb[k - i] for i in range(k + 1)) - sum(sa[i] * sb[n + k - i] for i in range(k + 1, n))
                assert int(got[k]) == exp % P.M64


def test_tggsw_external_product_vs_oracle(fhe, torch_cuda):
    """uniform-random TGGSW rows (validity is irrelevant for parity), N = 256, three gadget shapes, batch 3"""
    from oracle import pyref as P
    rnd = random.Random(7)
    n, batch = 256, 3
    t = fhe.TorusContext()
    for log_b, d in [(23, 1), (8, 3), (4, 5)]:
        dec = P.TorusDecomposor(log_b, d)
        ra = [[[rnd.getrandbits(64) for _ in range(n)] for _ in range(2 * d)] for _ in range(2)]
        rb = [[[rnd.getrandbits(64) for _ in range(n)] for _ in range(2 * d)] for _ in range(2)]
        key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, U(ra)), dev(torch_cuda, U(rb)), n)
        ca = [[rnd.getrandbits(64) for _ in range(n)] for _ in range(batch)]
        cb = [[rnd.getrandbits(64) for _ in range(n)] for _ in range(batch)]
        ca[0][:3] = [0, P.M64 - 1, 1 << 63]
        for idx in range(2):
            a, b = dev(torch_cuda, U(ca)), dev(torch_cuda, U(cb))
            key.external_product_(idx, a, b)
            for i in range(batch):
                ea, eb = P.tggsw_external_product(dec, ra[idx], rb[idx], ca[i], cb[i])
                assert L(host(a)[i]) == ea and L(host(b)[i]) == eb, (log_b, d, idx, i)


def test_tggsw_external_product_n2048(fhe, torch_cuda):
    """the reference's ring (big_n = 2048, log_b = 23, d = 1): one external product, bit-exact against the exact oracle"""
    from oracle import pyref as P
    rnd = random.Random(8)
    n, log_b, d = 2048, 23, 1
    dec = P.TorusDecomposor(log_b, d)
    t = fhe.TorusContext()
    ra = [[rnd.getrandbits(64) for _ in range(n)] for _ in range(2 * d)]
    rb = [[rnd.getrandbits(64) for _ in range(n)] for _ in range(2 * d)]
    key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, U([ra])), dev(torch_cuda, U([rb])), n)
    ca, cb = [rnd.getrandbits(64) for _ in range(n)], [rnd.getrandbits(64) for _ in range(n)]
    a, b = dev(torch_cuda, U([ca])), dev(torch_cuda, U([cb]))
    key.external_product_(0, a, b)
    ea, eb = P.tggsw_external_product(dec, ra, rb, ca, cb)
    assert L(host(a)) == ea and L(host(b)) == eb


def test_blind_rotate_and_gate_vs_oracle(fhe, torch_cuda):
    """N = 256, n_lwe = 6: CMUX chain, sample extract and TLWE key switch bit-exact against the exact oracle"""
    from oracle import pyref as P
    rnd = random.Random(9)
    n, n_lwe, batch, log_b, d = 256, 6, 3, 10, 2
    dec, ksdec = P.TorusDecomposor(log_b, d), P.TorusDecomposor(4, 5)
    t = fhe.TorusContext()
    brk = [([[rnd.getrandbits(64) for _ in range(n)] for _ in range(2 * d)], [[rnd.getrandbits(64) for _ in range(n)] for _ in range(2 * d)])
           for _ in range(n_lwe)]
    key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, U([k[0] for k in brk])), dev(torch_cuda, U([k[1] for k in brk])), n)
    v = [rnd.getrandbits(64) for _ in range(n)]
    a_raw = [[rnd.getrandbits(64) for _ in range(n_lwe)] for _ in range(batch)]
    b_raw = [rnd.getrandbits(64) for _ in range(batch)]
    a_raw[1][2] = 0  # rotation by zero: the CMUX leaves the accumulator untouched
    at = host(fhe.TorusContext.mod_switch(dev(torch_cuda, U(a_raw)), n))
    bt = host(fhe.TorusContext.mod_switch(dev(torch_cuda, U(b_raw)), n))
    for i in range(batch):
        assert L(at[i]) == P.tfhe_mod_switch(a_raw[i], n)
    assert L(bt) == P.tfhe_mod_switch(b_raw, n)
    oa, ob = key.blind_rotate(dev(torch_cuda, at), dev(torch_cuda, bt), dev(torch_cuda, U(v)))
    ksa = [[rnd.getrandbits(64) for _ in range(n_lwe)] for _ in range(n * 5)]
    ksb = [rnd.getrandbits(64) for _ in range(n * 5)]
    ea, eb = fhe.tglwe_sample_extract(oa, ob, n, 0)
    ka, kb = fhe.tlwe_key_switch(4, 5, dev(torch_cuda, U(ksa)), dev(torch_cuda, U(ksb)), ea, eb, n, n_lwe)
    for i in range(batch):
        acc = P.tfhe_blind_rotate(dec, brk, v, L(at[i]), int(bt[i]))
        assert L(host(oa)[i]) == acc[0] and L(host(ob)[i]) == acc[1], i
        xa, xb = P.tglwe_sample_extract(acc[0], acc[1], 0)
        assert L(host(ea)[i]) == xa and int(host(eb)[i]) == xb
        ya, yb = P.tlwe_key_switch(ksdec, ksa, ksb, xa, xb)
        assert L(host(ka)[i]) == ya and int(host(kb)[i]) == yb
    xa7, xb7 = fhe.tglwe_sample_extract(oa, ob, n, 7)
    assert (L(host(xa7)[0]), int(host(xb7)[0])) == P.tglwe_sample_extract(L(host(oa)[0]), L(host(ob)[0]), 7)


def test_gate_bootstrap_decode_level_reference_parameters(fhe, torch_cuda):
    """scheme/tfhe/src/bootstrapping.rs:139-165 on the GPU path, the reference's only parameter set: big_n = 2048, k = 1,
    log_b = 23, d = 1, n_lwe = 1024 (binary key), key switch (4, 5), log_p = 4, padding 1; LUTs identity / double / parity
    over all 16 messages in ONE batch of 48 ciphertexts.  Keys come from the oracle's key generator (noise as wide as the
    reference's tdg draws would be, a few units of the last place at these standard deviations)."""
    from oracle import pyref as P
    rnd = random.Random(11)
    n, n_lwe, log_p, padding, log_b, d = 2048, 1024, 4, 1, 23, 1
    p, log_delta = 1 << log_p, 64 - (log_p + padding)
    dec, ksdec = P.TorusDecomposor(log_b, d), P.TorusDecomposor(4, 5)
    z = [rnd.randint(0, 1) for _ in range(n_lwe)]
    s = [rnd.randint(0, 1) for _ in range(n)]
    t = fhe.TorusContext()
    # TGGSW(z_i) under s: rows = TGLWE zeros + z_i * base on a (rows 0..d) / on b (rows d..2d).  Encryptions of zero are
    # built with the exact GPU product (b = a * s + e): one batched torus_mul instead of n_lwe * 2d Python schoolbooks.
    rows = n_lwe * 2 * d
    rng = np.random.Generator(np.random.PCG64(5))
    A = rng.integers(0, 1 << 63, size=(rows, n), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(rows, n), dtype=np.uint64)
    S = np.tile(np.array(s, dtype=np.uint64), (rows, 1))
    dB = dev(torch_cuda, A.copy())
    t.mul_(dB, dev(torch_cuda, S), 1, n)
    Bm = host(dB) + rng.integers(0, 5, size=(rows, n), dtype=np.uint64) - np.uint64(2)  # e in [-2, 2]
    ra, rb = A.reshape(n_lwe, 2 * d, n).copy(), Bm.reshape(n_lwe, 2 * d, n).copy()
    for i, zi in enumerate(z):
        for j, base in enumerate(dec.bases):
            ra[i, j, 0] += np.uint64(zi * base % P.M64)
            rb[i, d + j, 0] += np.uint64(zi * base % P.M64)
    key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, ra), dev(torch_cuda, rb), n)
    ksa, ksb = P.tlwe_ksk_gen(ksdec, z, s, rnd, noise=2)

    def table(f):
        m_ = n >> log_p
        tt = [f(v) % p for v in range(p)]
        out = [tt[0]] * (m_ // 2)
        for x in tt[1:]:
            out += [x] * m_
        return out + [(-tt[0]) % p] * (m_ // 2)

    luts = [lambda v: v, lambda v: 2 * v, lambda v: v % 2]
    for f in luts:
        v = [(x << log_delta) % P.M64 for x in table(f)]
        cts = [P.tlwe_sk_encrypt(z, (m << log_delta) % P.M64, rnd, noise=2) for m in range(p)]
        a_raw, b_raw = U([c[0] for c in cts]), U([c[1] for c in cts])
        at = fhe.TorusContext.mod_switch(dev(torch_cuda, a_raw), n)
        bt = fhe.TorusContext.mod_switch(dev(torch_cuda, b_raw), n)
        oa, ob = key.blind_rotate(at, bt, dev(torch_cuda, U(v)))
        ea, eb = fhe.tglwe_sample_extract(oa, ob, n, 0)
        ka, kb = fhe.tlwe_key_switch(4, 5, dev(torch_cuda, U(ksa)), dev(torch_cuda, U(ksb)), ea, eb, n, n_lwe)
        for m in range(p):
            mu = ((P.tlwe_phase(z, L(host(ka)[m]), int(host(kb)[m])) + (1 << (log_delta - 1))) % P.M64) >> log_delta
            assert mu % p == f(m) % p, (m, mu)


def test_tlwe_key_switch_tiles(fhe, torch_cuda):
    """scheme/tfhe/src/tlwe.rs:144-153 on ragged batches that take the 2- and 4-ciphertext tiles of the tiled kernel"""
    from oracle import pyref as P
    rnd = random.Random(21)
    # the last two shapes take the wide-output kernel (>= 256 columns split over the grid, 8 ciphertexts per block, byte digits)
    for log_b, d, n_in, n_out, batch in [(4, 5, 16, 5, 514), (7, 3, 8, 140, 2051), (4, 5, 16, 300, 70), (7, 2, 32, 257, 129)]:
        dec = P.TorusDecomposor(log_b, d)
        ksa = [[rnd.getrandbits(64) for _ in range(n_out)] for _ in range(n_in * d)]
        ksb = [rnd.getrandbits(64) for _ in range(n_in * d)]
        a = np.random.Generator(np.random.PCG64(batch)).integers(0, 1 << 63, size=(batch, n_in), dtype=np.uint64) * np.uint64(2)
        b = np.random.Generator(np.random.PCG64(batch + 1)).integers(0, 1 << 63, size=batch, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
        a[0, :4] = [0, (1 << 64) - 1, 1 << 63, (1 << 63) - 1]
        ka, kb = fhe.tlwe_key_switch(log_b, d, dev(torch_cuda, U(ksa)), dev(torch_cuda, U(ksb)), dev(torch_cuda, a), dev(torch_cuda, b), n_in, n_out)
        ha, hb = host(ka), host(kb)
        for i in list(range(0, batch, 101)) + [batch - 2, batch - 1]:
            ya, yb = P.tlwe_key_switch(dec, ksa, ksb, L(a[i]), int(b[i]))
            assert L(ha[i]) == ya and int(hb[i]) == yb, (batch, i)


def test_tlwe_key_switch_split_grid_shapes(fhe, cref, torch_cuda):
    """The wide TLWE key switch splits ciphertexts x output columns x input coefficients over the grid and adds partial sums with 64-bit
    atomics (csrc/lwe_kernels.hpp `tlwe_key_switch_split`): every split the host picks -- small batches (many coefficient chunks), ragged
    tiles, a batch large enough that one block keeps ALL coefficients (the digit image near its LDS cap) -- gives the oracle's bits."""
    n_in, n_out, log_b, d = 512, 300, 4, 5
    rng = np.random.Generator(np.random.PCG64(77))
    r64 = lambda *s: rng.integers(0, 1 << 63, size=s, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=s, dtype=np.uint64)  # noqa: E731
    ksa, ksb = r64(n_in * d, n_out), r64(n_in * d)
    dka, dkb = dev(torch_cuda, ksa), dev(torch_cuda, ksb)
    for batch in (64, 77, 1000, 30000):
        ca, cb = r64(batch, n_in), r64(batch)
        oa, ob = fhe.tlwe_key_switch(log_b, d, dka, dkb, dev(torch_cuda, ca), dev(torch_cuda, cb), n_in, n_out)
        ha, hb = host(oa).reshape(batch, n_out), host(ob).reshape(batch)
        for i in sorted({0, 1, 15, 16, batch // 2, batch - 1}):
            wa, wb = cref.tlwe_key_switch(log_b, d, ksa, ksb, ca[i], int(cb[i]))
            assert np.array_equal(ha[i], wa) and int(hb[i]) == wb, (batch, i)


def test_blind_rotate_large_batch_is_deterministic(fhe, torch_cuda):
    """cfg5 ring (N = 1024, base 2^7, d = 3), 1024 ciphertexts: every SIMD busy with multi-wave teams sharing LDS images; two runs
    agree bit for bit, and a sample agrees with the same ciphertexts run as a small batch"""
    n, n_lwe, log_b, d, batch = 1024, 12, 7, 3, 1024
    gen = torch_cuda.Generator(device="cuda")
    gen.manual_seed(7)
    rnd = lambda *shape: torch_cuda.randint(-(1 << 63), (1 << 63) - 1, shape, dtype=torch_cuda.int64, device="cuda", generator=gen)  # noqa: E731
    t = fhe.TorusContext()
    key = fhe.TggswKey(t, log_b, d, rnd(n_lwe, 2 * d, n), rnd(n_lwe, 2 * d, n), n)
    v = rnd(n)
    at = fhe.TorusContext.mod_switch(rnd(batch, n_lwe), n)
    bt = fhe.TorusContext.mod_switch(rnd(batch), n)
    at[5, :] = 0                                   # a ciphertext every CMUX of which short-circuits
    o1 = key.blind_rotate(at, bt, v)
    o2 = key.blind_rotate(at, bt, v)
    assert torch_cuda.equal(o1[0], o2[0]) and torch_cuda.equal(o1[1], o2[1])
    pick = [0, 5, 1023]
    s = key.blind_rotate(at[pick].contiguous(), bt[pick].contiguous(), v)
    assert torch_cuda.equal(o1[0][pick], s[0]) and torch_cuda.equal(o1[1][pick], s[1])
    assert int(o1[0][5].abs().max()) == 0          # (0, v X^-b) untouched: a stays zero


@pytest.mark.parametrize("log_b,d", [(15, 2), (16, 2)])
def test_external_product_at_the_path_boundary(fhe, torch_cuda, log_b, d):
    """N = 256: bound 2d N 2^(62 + log_b) = 2^88 (base 2^15: three 30-bit primes) and 2^89 (base 2^16: two 60-bit primes), with
    WORST-CASE operands -- every key coefficient -2^63 and every digit at its extreme -- so the exact integer coefficients sit as
    close to P/2 as this shape can bring them; both paths bit-equal to the exact oracle"""
    from oracle import pyref as P
    n = 256
    dec = P.TorusDecomposor(log_b, d)
    t = fhe.TorusContext()
    big = 1 << 63
    ra = [[[big] * n for _ in range(2 * d)]]
    rb = [[[big if (i + r) % 3 else (1 << 63) - 1 for i in range(n)] for r in range(2 * d)]]
    key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, U(ra)), dev(torch_cuda, U(rb)), n)
    # torus values whose every digit is -2^(log_b-1) (the most negative balanced digit): sum_j (-B/2) B^j scaled to the top bits
    half = 1 << (log_b - 1)
    v = (-sum(half << (64 - log_b * (j + 1)) for j in range(d))) % (1 << 64)
    rnd = random.Random(31)
    ca = [[v] * n, [rnd.getrandbits(64) for _ in range(n)]]
    cb = [[v] * n, [rnd.getrandbits(64) for _ in range(n)]]
    a, b = dev(torch_cuda, U(ca)), dev(torch_cuda, U(cb))
    key.external_product_(0, a, b)
    for i in range(2):
        ea, eb = P.tggsw_external_product(dec, ra[0], rb[0], ca[i], cb[i])
        assert L(host(a)[i]) == ea and L(host(b)[i]) == eb, (log_b, i)


# ---- cfg5's ring, N = 2^10: every instantiation bench.py times, against the exact oracle's C restatement (oracle/ref_ring.c) ----

@pytest.mark.parametrize("log_b,d,path", [(7, 3, "three key pieces through f64 transforms (the cfg5 gadget)"), (10, 2, "three 30-bit primes"),
                                           (23, 1, "two 60-bit primes"), (16, 2, "two 60-bit primes")])
def test_tggsw_external_product_n1024_vs_oracle(fhe, cref, torch_cuda, log_b, d, path):
    """scheme/tfhe/src/tggsw.rs:100-112 at N = 1024 on EVERY exact path (fhe_tggsw_prepare picks: 2d N 2^log_b <= 2^21 and base <= 2^7:
    three key pieces through f64 transforms, (7,3); else from the bound 2d N 2^(62 + log_b): (10,2) -> 2^85: three 30-bit primes;
    (23,1) -> 2^97, (16,2) -> 2^91: two 60-bit primes),
    two key entries, a ragged batch with extreme torus values; bit-equal to the exact product"""
    n, batch, count = 1024, 5, 2
    rng = np.random.Generator(np.random.PCG64(1000 + log_b))
    r64 = lambda *shape: rng.integers(0, 1 << 63, size=shape, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape, dtype=np.uint64)  # noqa: E731
    ra, rb = r64(count, 2 * d, n), r64(count, 2 * d, n)
    ra[0, 0, :] = np.uint64(1 << 63)                     # a key row of all -2^63
    ca, cb = r64(batch, n), r64(batch, n)
    ca[0, :5] = [0, (1 << 64) - 1, 1 << 63, (1 << 63) - 1, 1]
    half = 1 << (log_b - 1)
    cb[1, :] = np.uint64((-sum(half << (64 - log_b * (j + 1)) for j in range(d))) % (1 << 64))  # every digit at its extreme
    t = fhe.TorusContext()
    key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, ra), dev(torch_cuda, rb), n)
    for idx in range(count):
        a, b = dev(torch_cuda, ca), dev(torch_cuda, cb)
        key.external_product_(idx, a, b)
        ha, hb = host(a), host(b)
        for i in range(batch):
            ea, eb = cref.tggsw_external_product(log_b, d, ra[idx], rb[idx], ca[i], cb[i])
            assert np.array_equal(ha[i], ea) and np.array_equal(hb[i], eb), (path, idx, i)


@pytest.mark.parametrize("log_n,log_b,d", [(10, 7, 3), (10, 7, 4), (10, 6, 4), (9, 7, 4), (8, 7, 4), (10, 4, 7), (8, 2, 8)])
def test_exact_f64_path_equals_the_integer_paths_on_worst_case_operands(fhe, cref, torch_cuda, log_n, log_b, d):
    """The three-piece f64 path (csrc/torusf_kernels.hpp TorusX3: key words cut into signed pieces of 22 / 21 / 21 bits, half-size complex
    transforms, every product rounded to the integer it is) must be EXACT, not close: here against the oracle and against the three-prime
    integer path (lab switch NO_F64_EXACT) on the operands that make the rounding error largest -- every digit at +-2^(log_b - 1) and every
    key piece at its extreme (key words 0x7fff..., 0x8000..., and the word whose three pieces are all at their negative end), constant and
    alternating in sign, besides uniform ones; the largest gadget the path takes ((10,7,4): 2d N 2^log_b = 2^20 of the 2^21 the host
    admits), cfg5's (10,7,3), long ones ((10,4,7): 14 limbs, (8,2,8): 16 limbs); external product, CMUX and a short blind rotation."""
    n, batch = 1 << log_n, 6
    rng = np.random.Generator(np.random.PCG64(7000 + log_n * 10 + d))
    r64 = lambda *s: rng.integers(0, 1 << 63, size=s, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=s, dtype=np.uint64)  # noqa: E731
    n_lwe = 4
    ra, rb = r64(n_lwe, 2 * d, n), r64(n_lwe, 2 * d, n)
    low = ((-(1 << 21)) + ((-(1 << 20)) << 22) + ((-(1 << 20)) << 43)) % (1 << 64)       # all three pieces at their negative end
    ra[0, :, :] = np.uint64((1 << 63) - 1)
    rb[0, :, :] = np.uint64(1 << 63)
    ra[1, :, :] = np.uint64(low)
    rb[1, :, 0::2] = np.uint64((1 << 63) - 1)
    rb[1, :, 1::2] = np.uint64(1 << 63)
    half = 1 << (log_b - 1)
    neg = (-sum(half << (64 - log_b * (j + 1)) for j in range(d) if 64 - log_b * (j + 1) >= 0)) % (1 << 64)   # every digit -2^(log_b-1)
    pos = (-neg) % (1 << 64)
    ca, cb = r64(batch, n), r64(batch, n)
    ca[0, :], cb[0, :] = np.uint64(neg), np.uint64(pos)
    ca[1, 0::2], ca[1, 1::2], cb[1, :] = np.uint64(neg), np.uint64(pos), np.uint64(neg)
    t = fhe.TorusContext()
    res = []
    for off in (0, 1):  # the f64 path | the integer path
        fhe.set_option("NO_F64_EXACT", off)
        try:
            key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, ra), dev(torch_cuda, rb), n)
        finally:
            fhe.set_option("NO_F64_EXACT", 0)
        outs = []
        for idx in range(2):
            a, b = dev(torch_cuda, ca), dev(torch_cuda, cb)
            key.external_product_(idx, a, b)
            outs += [a, b]
        outs += list(key.cmux(1, dev(torch_cuda, ca), dev(torch_cuda, cb), dev(torch_cuda, cb), dev(torch_cuda, ca)))
        a_t = rng.integers(1, 2 * n, size=(batch, n_lwe), dtype=np.uint64) if off == 0 else a_t  # noqa: F821
        b_t = rng.integers(0, 2 * n, size=batch, dtype=np.uint64) if off == 0 else b_t          # noqa: F821
        outs += list(key.blind_rotate(dev(torch_cuda, a_t), dev(torch_cuda, b_t), dev(torch_cuda, ca[0])))
        res.append(outs)
    for x, y in zip(*res):
        assert torch_cuda.equal(x, y)
    for idx in range(2):
        for i in range(batch):
            ea, eb = cref.tggsw_external_product(log_b, d, ra[idx], rb[idx], ca[i], cb[i])
            assert np.array_equal(host(res[0][2 * idx])[i], ea) and np.array_equal(host(res[0][2 * idx + 1])[i], eb), (idx, i)
    ea, eb = cref.tfhe_blind_rotate(log_b, d, ra, rb, ca[0], a_t, b_t, threads=8)
    assert np.array_equal(host(res[0][6]).reshape(batch, n), ea) and np.array_equal(host(res[0][7]).reshape(batch, n), eb)


def test_exact_f64_path_whole_cfg5_blind_rotations_equal_the_integer_path(fhe, torch_cuda):
    """BASELINE config 5 at full length on both exact paths: 630 CMUXes x 512 ciphertexts with uniform 64-bit key words -- 660 million
    coefficients, each rounded three times on the f64 path; one rounding the wrong way anywhere would re-randomise everything after it.
    Every accumulator of the three-piece f64 path is bit-identical to the three-prime integer path's (which the oracle tests pin)."""
    n, n_lwe, log_b, d, batch = 1024, 630, 7, 3, 512
    gen = torch_cuda.Generator(device="cuda")
    gen.manual_seed(55)
    rnd = lambda *shape: torch_cuda.randint(-(1 << 63), (1 << 63) - 1, shape, dtype=torch_cuda.int64, device="cuda", generator=gen)  # noqa: E731
    ra, rb, v = rnd(n_lwe, 2 * d, n), rnd(n_lwe, 2 * d, n), rnd(n)
    a_t = torch_cuda.randint(0, 2 * n, (batch, n_lwe), dtype=torch_cuda.int64, device="cuda", generator=gen)
    b_t = torch_cuda.randint(0, 2 * n, (batch,), dtype=torch_cuda.int64, device="cuda", generator=gen)
    t = fhe.TorusContext()
    outs = []
    for off in (0, 1):
        fhe.set_option("NO_F64_EXACT", off)
        try:
            key = fhe.TggswKey(t, log_b, d, ra, rb, n)
        finally:
            fhe.set_option("NO_F64_EXACT", 0)
        outs.append(key.blind_rotate(a_t, b_t, v))
        del key
    assert torch_cuda.equal(outs[0][0], outs[1][0]) and torch_cuda.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("log_b,d,n_lwe,batch", [(7, 3, 10, 7), (23, 1, 8, 7), (7, 3, 4, 1100), (23, 1, 3, 600)])
def test_blind_rotate_n1024_vs_oracle(fhe, cref, torch_cuda, log_b, d, n_lwe, batch):
    """scheme/tfhe/src/bootstrapping.rs:84-104 at N = 1024 (cfg5's ring): mod switch, the whole CMUX chain, sample extract and the
    TLWE key switch, bit-equal to the exact oracle on both prime paths; small batches and batches that fill the GPU (every
    ciphertext of the small ones, a spread sample of the large ones, their first and last included)"""
    n, ks_log_b, ks_d = 1024, 4, 5
    rng = np.random.Generator(np.random.PCG64(2000 + log_b + batch))
    r64 = lambda *shape: rng.integers(0, 1 << 63, size=shape, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape, dtype=np.uint64)  # noqa: E731
    bra, brb, v = r64(n_lwe, 2 * d, n), r64(n_lwe, 2 * d, n), r64(n)
    a_raw, b_raw = r64(batch, n_lwe), r64(batch)
    a_raw[batch - 1, 1] = 0                               # a rotation by zero: that CMUX short-circuits
    a_raw[0, 0] = np.uint64((1 << 64) - 1)                # rounds up to 2N = 0 (mod 2N)
    t = fhe.TorusContext()
    key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, bra), dev(torch_cuda, brb), n)
    at = fhe.TorusContext.mod_switch(dev(torch_cuda, a_raw), n)
    bt = fhe.TorusContext.mod_switch(dev(torch_cuda, b_raw), n)
    assert np.array_equal(host(at).reshape(batch, n_lwe), cref.tfhe_mod_switch(a_raw, n)) and np.array_equal(host(bt), cref.tfhe_mod_switch(b_raw, n))
    oa, ob = key.blind_rotate(at, bt, dev(torch_cuda, v))
    ksa, ksb = r64(n * ks_d, n_lwe), r64(n * ks_d)
    ea, eb = fhe.tglwe_sample_extract(oa, ob, n, 0)
    ka, kb = fhe.tlwe_key_switch(ks_log_b, ks_d, dev(torch_cuda, ksa), dev(torch_cuda, ksb), ea, eb, n, n_lwe)
    # the single-call gate (bootstrapping.rs:78-82, fhe_tfhe_bootstrap) is the same four steps on one stream: bit-identical
    ga1, gb1 = key.bootstrap(ks_log_b, ks_d, dev(torch_cuda, ksa), dev(torch_cuda, ksb), dev(torch_cuda, v), dev(torch_cuda, a_raw), dev(torch_cuda, b_raw))
    assert torch_cuda.equal(ga1, ka) and torch_cuda.equal(gb1, kb)
    if batch <= 16:  # ... and with every operand in host memory
        ga2, gb2 = key.bootstrap(ks_log_b, ks_d, ksa, ksb, v, a_raw, b_raw)
        assert np.array_equal(ga2, host(ka).reshape(batch, n_lwe)) and np.array_equal(gb2, host(kb))
    pick = list(range(batch)) if batch <= 16 else sorted(set([0, 1, 63, 64, 255, 256, 511, 512, 513, batch - 2, batch - 1] + list(range(7, batch, 97))))
    ga, gb = cref.tfhe_bootstrap(log_b, d, ks_log_b, ks_d, bra, brb, ksa, ksb, v, a_raw[pick], b_raw[pick], threads=16)
    ra_, rb_ = cref.tfhe_blind_rotate(log_b, d, bra, brb, v, cref.tfhe_mod_switch(a_raw[pick], n), cref.tfhe_mod_switch(b_raw[pick], n), threads=16)
    hoa, hob, hka, hkb = host(oa).reshape(batch, n), host(ob).reshape(batch, n), host(ka).reshape(batch, n_lwe), host(kb)
    for j, i in enumerate(pick):
        assert np.array_equal(hoa[i], ra_[j]) and np.array_equal(hob[i], rb_[j]), ("blind rotation", i)
        assert np.array_equal(hka[i], ga[j]) and int(hkb[i]) == int(gb[j]), ("gate output", i)


def test_gate_bootstrap_decode_level_n1024(fhe, cref, torch_cuda):
    """scheme/tfhe/src/bootstrapping.rs:139-165 shaped for cfg5: big_n = 1024, k = 1, base 2^7 x 3 (three 30-bit primes), binary
    LWE key of 256 bits, key switch (4, 5), log_p 4, padding 1; LUTs identity / double / parity over all 16 messages with a VALID
    bootstrapping key (encryptions of zero built with the exact GPU product, noise of a few units).  Decode-level equality, and
    the same ciphertexts bit-equal to the exact oracle's gate."""
    from oracle import pyref as P
    rnd = random.Random(12)
    n, n_lwe, log_p, padding, log_b, d = 1024, 256, 4, 1, 7, 3
    p, log_delta = 1 << log_p, 64 - (log_p + padding)
    dec, ksdec = P.TorusDecomposor(log_b, d), P.TorusDecomposor(4, 5)
    z = [rnd.randint(0, 1) for _ in range(n_lwe)]
    s = [rnd.randint(0, 1) for _ in range(n)]
    t = fhe.TorusContext()
    rows = n_lwe * 2 * d
    rng = np.random.Generator(np.random.PCG64(6))
    A = rng.integers(0, 1 << 63, size=(rows, n), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(rows, n), dtype=np.uint64)
    S = np.tile(np.array(s, dtype=np.uint64), (rows, 1))
    dB = dev(torch_cuda, A.copy())
    t.mul_(dB, dev(torch_cuda, S), 1, n)
    Bm = host(dB) + rng.integers(0, 5, size=(rows, n), dtype=np.uint64) - np.uint64(2)  # b = a s + e, e in [-2, 2]
    assert np.array_equal(host(dB)[3], cref.torus_mul_exact(A[3], S[3]))                 # the key material itself is exact
    ra, rb = A.reshape(n_lwe, 2 * d, n).copy(), Bm.reshape(n_lwe, 2 * d, n).copy()
    for i, zi in enumerate(z):
        for j, base in enumerate(dec.bases):  # torus addition wraps
            ra[i, j, 0] = np.uint64((int(ra[i, j, 0]) + zi * base) % P.M64)
            rb[i, d + j, 0] = np.uint64((int(rb[i, d + j, 0]) + zi * base) % P.M64)
    key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, ra), dev(torch_cuda, rb), n)
    ksa, ksb = P.tlwe_ksk_gen(ksdec, z, s, rnd, noise=2)
    ksa, ksb = U(ksa), U(ksb)

    def table(f):
        m_ = n >> log_p
        tt = [f(v) % p for v in range(p)]
        out = [tt[0]] * (m_ // 2)
        for x in tt[1:]:
            out += [x] * m_
        return out + [(-tt[0]) % p] * (m_ // 2)

    for li, f in enumerate((lambda v: v, lambda v: 2 * v, lambda v: v % 2)):
        v = U([(x << log_delta) % P.M64 for x in table(f)])
        cts = [P.tlwe_sk_encrypt(z, (m << log_delta) % P.M64, rnd, noise=2) for m in range(p)]
        a_raw, b_raw = U([c[0] for c in cts]), U([c[1] for c in cts])
        at = fhe.TorusContext.mod_switch(dev(torch_cuda, a_raw), n)
        bt = fhe.TorusContext.mod_switch(dev(torch_cuda, b_raw), n)
        oa, ob = key.blind_rotate(at, bt, dev(torch_cuda, v))
        ea, eb = fhe.tglwe_sample_extract(oa, ob, n, 0)
        ka, kb = fhe.tlwe_key_switch(4, 5, dev(torch_cuda, ksa), dev(torch_cuda, ksb), ea, eb, n, n_lwe)
        hka, hkb = host(ka).reshape(p, n_lwe), host(kb)
        for m in range(p):
            mu = ((P.tlwe_phase(z, L(hka[m]), int(hkb[m])) + (1 << (log_delta - 1))) % P.M64) >> log_delta
            assert mu % p == f(m) % p, (li, m, mu)
        if li == 0:  # 4 of the 16 through the exact oracle's whole gate (n_lwe = 256 CMUXes each)
            ga, gb = cref.tfhe_bootstrap(log_b, d, 4, 5, ra, rb, ksa, ksb, v, a_raw[[0, 5, 10, 15]], b_raw[[0, 5, 10, 15]], threads=16)
            assert np.array_equal(hka[[0, 5, 10, 15]], ga) and np.array_equal(hkb[[0, 5, 10, 15]], gb)


@pytest.mark.parametrize("log_n,log_b,d", [(8, 15, 2), (10, 7, 3), (11, 23, 1)])
def test_cmux_and_rotate_entries(fhe, torch_cuda, log_n, log_b, d):
    """Stand-alone entries of row T: `Tggsw::cmux(b, ct0, ct1)` = ct0 + external_product(b, ct1 - ct0) for ARBITRARY ct1
    (scheme/tfhe/src/tggsw.rs:114-121; the blind rotation only ever needs ct1 = ct0 X^a) and `TglweCiphertext::rotate`
    (tglwe.rs:61-66), bit-equal to the exact oracle on both prime paths; cmux(b, ct, ct.rotate(a)) is one step of the reference's
    blind rotation (bootstrapping.rs:91-94)."""
    from oracle import cref
    n, batch = 1 << log_n, 3
    rng = np.random.Generator(np.random.PCG64(log_n))
    r64 = lambda *s: rng.integers(0, 1 << 63, size=s, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=s, dtype=np.uint64)  # noqa: E731
    ra, rb = r64(2, 2 * d, n), r64(2, 2 * d, n)
    t = fhe.TorusContext()
    key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, ra), dev(torch_cuda, rb), n)
    c0a, c0b, c1a, c1b = r64(batch, n), r64(batch, n), r64(batch, n), r64(batch, n)
    oa, ob = key.cmux(1, dev(torch_cuda, c0a), dev(torch_cuda, c0b), dev(torch_cuda, c1a), dev(torch_cuda, c1b))
    for i in range(batch):
        ea, eb = cref.tggsw_external_product(log_b, d, ra[1], rb[1], c1a[i] - c0a[i], c1b[i] - c0b[i])
        assert np.array_equal(host(oa)[i], c0a[i] + ea) and np.array_equal(host(ob)[i], c0b[i] + eb), i
    for k in (0, 1, n - 1, n, n + 3, 2 * n - 1, -1, -n - 5, 7 * n + 2):
        xa, xb = fhe.tglwe_rotate(dev(torch_cuda, c0a), dev(torch_cuda, c0b), n, k)
        for i in range(batch):
            assert np.array_equal(host(xa)[i], cref.torus_monomial_mul(c0a[i], k)) and np.array_equal(host(xb)[i], cref.torus_monomial_mul(c0b[i], k)), (k, i)
    # one blind-rotation step through the two entries == the fused CMUX of the blind rotation kernel (rotation form)
    a_t = 2 * n - 37
    xa, xb = fhe.tglwe_rotate(dev(torch_cuda, c0a), dev(torch_cuda, c0b), n, a_t)
    sa, sb = key.cmux(0, dev(torch_cuda, c0a), dev(torch_cuda, c0b), xa, xb)
    for i in range(batch):
        ra_, rb_ = cref.torus_monomial_mul(c0a[i], a_t), cref.torus_monomial_mul(c0b[i], a_t)
        ea, eb = cref.tggsw_external_product(log_b, d, ra[0], rb[0], ra_ - c0a[i], rb_ - c0b[i])
        assert np.array_equal(host(sa)[i], c0a[i] + ea) and np.array_equal(host(sb)[i], c0b[i] + eb)


@pytest.mark.parametrize("log_n,log_b,d,batch", [(10, 7, 3, 5), (10, 7, 4, 3), (10, 8, 3, 2), (9, 6, 2, 4), (8, 4, 4, 6), (11, 7, 2, 2)])
def test_packed_digit_blind_rotation_equals_the_unpacked_one(fhe, torch_cuda, log_n, log_b, d, batch):
    """Gadgets of base <= 2^7 (a digit of base 2^8 reaches +128: no byte) and at most 8 limbs run the blind rotation with each CMUX's digits computed once and parked as bytes, and
    the multiply-accumulate unreduced (torus30_kernels.hpp, `_pk` kernels); the lab switch NO_PACKED_DIGITS keeps the kernel that
    decomposes once per prime.  Same integers, same CRT: bit-identical accumulators -- incl. extreme inputs (digits at both ends of
    their range, key words -2^63) -- and (cfg5's shape) equal to the exact oracle."""
    from oracle import cref
    n, n_lwe = 1 << log_n, 7
    rng = np.random.Generator(np.random.PCG64(100 + log_n + d))
    r64 = lambda *s: rng.integers(0, 1 << 63, size=s, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=s, dtype=np.uint64)  # noqa: E731
    ra, rb, v = r64(n_lwe, 2 * d, n), r64(n_lwe, 2 * d, n), r64(n)
    ra[0, :, :] = np.uint64(1 << 63)                       # the most negative key words
    half = 1 << (log_b - 1)
    v[: n // 2] = np.uint64((-sum(half << (64 - log_b * (j + 1)) for j in range(d))) % (1 << 64))  # every digit at its most negative
    a_t = rng.integers(0, 2 * n, size=(batch, n_lwe), dtype=np.uint64)
    a_t[0, 0] = 0                                          # a rotation by zero short-circuits
    b_t = rng.integers(0, 2 * n, size=batch, dtype=np.uint64)
    t = fhe.TorusContext()
    key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, ra), dev(torch_cuda, rb), n)
    outs = []
    for off in (0, 1):  # the packed kernel | the older one
        fhe.set_option("NO_PACKED_DIGITS", off)
        try:
            outs.append(key.blind_rotate(dev(torch_cuda, a_t), dev(torch_cuda, b_t), dev(torch_cuda, v)))
        finally:
            fhe.set_option("NO_PACKED_DIGITS", 0)
    for o in outs[1:]:
        assert torch_cuda.equal(outs[0][0], o[0]) and torch_cuda.equal(outs[0][1], o[1])
    if (log_n, log_b, d) == (10, 7, 3):
        ea, eb = cref.tfhe_blind_rotate(log_b, d, ra, rb, v, a_t, b_t, threads=8)
        assert np.array_equal(host(outs[0][0]).reshape(batch, n), ea) and np.array_equal(host(outs[0][1]).reshape(batch, n), eb)


def test_c_program_tfhe_gate(tmp_path, fhe):
    """examples/tfhe_gate_demo.c: row T through the boundary from plain C (no Python, no torch in that process): TGGSW external product
    against a schoolbook computed in the C program -- exact mode bit for bit, fft64 mode within the reference's bound -- the single-call
    gate on host buffers, status codes."""
    import os
    import subprocess
    from conftest import ROOT
    lib_dir = os.path.dirname(fhe.lib_path())
    exe = tmp_path / "tfhe_gate_demo"
    cmd = ["gcc", "-std=c99", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "tfhe_gate_demo.c"), "-o", str(exe),
           "-L", lib_dir, "-lfhe_ring", "-Wl,--allow-shlib-undefined", "-Wl,-rpath," + lib_dir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "tfhe_gate_demo ok" in r.stdout, r.stdout + r.stderr
