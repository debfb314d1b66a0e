"""GPU tests of the opt-in f64 FFT mode of row T (fhe_tggsw_prepare_fft64, csrc/torusf_kernels.hpp): the product the reference itself
computes with (util/src/ring/fft/c64.rs:11-108).  Floating point: NOT bit-identical to anything (the reference's own low bits depend
on its libm), so the acceptance is the reference's -- (ii) per product |result - exact| <= 2^(64 + log_b + log_n - 53), the bound of
c64.rs:186-208 `precision`, here against the exact oracle with 2d products summed; small operands exact (c64.rs:169-184); (i)
decode-level equality of everything built on it (tggsw.rs / bootstrapping.rs tests).  The tolerance is written in each test."""
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


def r64(rng, *s):
    return rng.integers(0, 1 << 63, size=s, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=s, dtype=np.uint64)


@pytest.mark.parametrize("log_n,log_b,d", [(8, 8, 8), (8, 15, 2), (9, 6, 2), (10, 7, 3), (10, 10, 2), (11, 23, 1), (11, 4, 5)])
def test_fft64_external_product_within_the_reference_bound(fhe, cref, torch_cuda, log_n, log_b, d):
    """TOLERANCE: |fft64 - exact| <= 2d * 2^(64 + log_b + log_n - 53) on uniform 64-bit keys and ciphertexts (2d products of a
    uniform torus polynomial with a digit polynomial |digit| <= 2^(log_b - 1), each inside c64.rs:186-208's bound); cmux and the
    plain external product; every gadget shape of the exact tests incl. the reference's bootstrap set (N = 2048, base 2^23)."""
    n, batch = 1 << log_n, 4
    rng = np.random.Generator(np.random.PCG64(300 + log_n + d))
    ra, rb = r64(rng, 2, 2 * d, n), r64(rng, 2, 2 * d, n)
    ca, cb, c1a, c1b = r64(rng, batch, n), r64(rng, batch, n), r64(rng, batch, n), r64(rng, batch, n)
    t = fhe.TorusContext()
    key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, ra), dev(torch_cuda, rb), n, fft64=True)
    bound = 2 * d * (1 << (64 + log_b + log_n - 53))
    xa, xb = dev(torch_cuda, ca), dev(torch_cuda, cb)
    key.external_product_(1, xa, xb)
    worst = 0
    for i in range(batch):
        ea, eb = cref.tggsw_external_product(log_b, d, ra[1], rb[1], ca[i], cb[i])
        for got, want in ((host(xa)[i], ea), (host(xb)[i], eb)):
            err = int(np.abs((got - want).view(np.int64)).max())
            worst = max(worst, err)
            assert err <= bound, (i, err, bound)
    assert worst > 0 or log_b * d < 20          # it IS floating point: uniform 64-bit key words do not survive 53 bits
    oa, ob = key.cmux(0, dev(torch_cuda, ca), dev(torch_cuda, cb), dev(torch_cuda, c1a), dev(torch_cuda, c1b))
    for i in range(batch):
        ea, eb = cref.tggsw_external_product(log_b, d, ra[0], rb[0], c1a[i] - ca[i], c1b[i] - cb[i])
        assert int(np.abs((host(oa)[i] - (ca[i] + ea)).view(np.int64)).max()) <= bound
        assert int(np.abs((host(ob)[i] - (cb[i] + eb)).view(np.int64)).max()) <= bound


@pytest.mark.parametrize("log_n", [8, 9, 10, 11])
def test_fft64_small_operands_are_exact(fhe, cref, torch_cuda, log_n):
    """c64.rs:169-184: with |key word| < 2^15 every partial sum is an integer below 2^53: the f64 transform loses nothing and the mode
    must equal the exact product bit for bit (a wrong twiddle, layout or rounding rule cannot hide behind a tolerance here)."""
    n, log_b, d, batch = 1 << log_n, 6, 3, 3
    rng = np.random.Generator(np.random.PCG64(400 + log_n))
    ra = rng.integers(-(1 << 15), 1 << 15, size=(2, 2 * d, n), dtype=np.int64).view(np.uint64)
    rb = rng.integers(-(1 << 15), 1 << 15, size=(2, 2 * d, n), dtype=np.int64).view(np.uint64)
    ca, cb = r64(rng, batch, n), r64(rng, batch, n)
    t = fhe.TorusContext()
    key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, ra), dev(torch_cuda, rb), n, fft64=True)
    xa, xb = dev(torch_cuda, ca), dev(torch_cuda, cb)
    key.external_product_(0, xa, xb)
    for i in range(batch):
        ea, eb = cref.tggsw_external_product(log_b, d, ra[0], rb[0], ca[i], cb[i])
        assert np.array_equal(host(xa)[i], ea) and np.array_equal(host(xb)[i], eb), i
    # the whole blind rotation on such a key: rotations, differences, accumulation -- bit-equal to the exact oracle
    n_lwe = 5
    bra = rng.integers(-(1 << 10), 1 << 10, size=(n_lwe, 2 * d, n), dtype=np.int64).view(np.uint64)
    brb = rng.integers(-(1 << 10), 1 << 10, size=(n_lwe, 2 * d, n), dtype=np.int64).view(np.uint64)
    v = r64(rng, n)
    a_t = rng.integers(0, 2 * n, size=(batch, n_lwe), dtype=np.uint64)
    a_t[0, 0] = 0
    b_t = rng.integers(0, 2 * n, size=batch, dtype=np.uint64)
    brk = fhe.TggswKey(t, log_b, d, dev(torch_cuda, bra), dev(torch_cuda, brb), n, fft64=True)
    oa, ob = brk.blind_rotate(dev(torch_cuda, a_t), dev(torch_cuda, b_t), dev(torch_cuda, v))
    ea, eb = cref.tfhe_blind_rotate(log_b, d, bra, brb, v, a_t, b_t, threads=8)
    assert np.array_equal(host(oa).reshape(batch, n), ea) and np.array_equal(host(ob).reshape(batch, n), eb)


@pytest.mark.parametrize("log_n,log_b,d", [(10, 7, 3), (10, 7, 4), (9, 7, 4)])
def test_f64_transforms_keep_a_margin_beyond_the_exact_paths_operands(fhe, cref, torch_cuda, log_n, log_b, d):
    """The exact mode's three-piece path (csrc/torusf_kernels.hpp TorusX3) relies on these transforms reproducing integer products with an
    error below 1/2 for key pieces up to 2^21 in magnitude (bound in DESIGN.md section 9: <= 0.04).  The fft64 mode runs the SAME butterflies
    on whatever key words it is given, so it measures the margin: with every digit at +-2^(log_b - 1) and key words at +-2^21 x 2^s, random
    and alternating signs, its output must still be bit-equal to the exact product for s = 0 (the pieces' size) and for s = 3 and 5 (8 and
    32 times larger: the bound would be 0.3 and 1.3) -- the error in practice sits far inside the worst-case bound."""
    n, batch = 1 << log_n, 4
    rng = np.random.Generator(np.random.PCG64(900 + log_n + d))
    half = 1 << (log_b - 1)
    neg = (-sum(half << (64 - log_b * (j + 1)) for j in range(d))) % (1 << 64)
    pos = (-neg) % (1 << 64)
    ca, cb = np.empty((batch, n), dtype=np.uint64), np.empty((batch, n), dtype=np.uint64)
    ca[0, :], cb[0, :] = np.uint64(neg), np.uint64(pos)
    ca[1, 0::2], ca[1, 1::2], cb[1, :] = np.uint64(neg), np.uint64(pos), np.uint64(neg)
    sign = rng.integers(0, 2, size=(2, 2, n), dtype=np.uint64)
    ca[2:], cb[2:] = np.where(sign[0] == 1, np.uint64(neg), np.uint64(pos)), np.where(sign[1] == 1, np.uint64(neg), np.uint64(pos))
    t = fhe.TorusContext()
    for s_ in (0, 3, 5):
        mag = (1 << (21 + s_)) - 1
        ra = np.where(rng.integers(0, 2, size=(1, 2 * d, n)) == 1, mag, -mag).astype(np.int64).view(np.uint64)
        rb = np.full((1, 2 * d, n), mag, dtype=np.int64)
        rb[:, :, 1::2] = -mag
        rb = rb.view(np.uint64)
        key = fhe.TggswKey(t, log_b, d, dev(torch_cuda, ra), dev(torch_cuda, rb), n, fft64=True)
        xa, xb = dev(torch_cuda, ca), dev(torch_cuda, cb)
        key.external_product_(0, xa, xb)
        for i in range(batch):
            ea, eb = cref.tggsw_external_product(log_b, d, ra[0], rb[0], ca[i], cb[i])
            assert np.array_equal(host(xa)[i], ea) and np.array_equal(host(xb)[i], eb), (s_, i)


@pytest.mark.parametrize("n,n_lwe,log_p,log_b,d,sd_lwe,sd_glwe", [(2048, 1024, 4, 23, 1, 1.339775301998614e-7, 2.845267479601915e-15),
                                                                   (1024, 630, 3, 7, 3, 2.0 ** -20, 2.0 ** -25)])
def test_fft64_gate_bootstrap_decode_level(fhe, torch_cuda, n, n_lwe, log_p, log_b, d, sd_lwe, sd_glwe):
    """(i) decode level: the reference's own `bootstrap` test (scheme/tfhe/src/bootstrapping.rs:139-165) with ITS parameter set (big_n =
    2048, base 2^23 x 1, n = 1024, key switch (4, 5), log_p 4) -- the setting its f64 product was written for -- and BASELINE config 5's
    shape (N = 1024, n = 630, base 2^7 x 3), both through the fft64 mode on device-made keys: LUTs identity / double / parity over every
    message; and the same inputs through the exact mode decode to the same messages."""
    from oracle import pyref as P
    padding, ks_lb, ks_d = 1, 4, 5
    p, log_delta = 1 << log_p, 64 - (log_p + padding)
    like = dev(torch_cuda, U([0]))
    t = fhe.TorusContext()
    z, s = fhe.sample_binary(800, 0, like, n_lwe), fhe.sample_binary(800, 1, like, n)
    zh = L(host(z))
    pt = np.zeros((n_lwe, n), dtype=np.uint64)
    pt[:, 0] = host(z)
    ra, rb = fhe.tggsw_encrypt(t, log_b, d, s, dev(torch_cuda, pt), n, sd_glwe, 801, 0)
    keys = [fhe.TggswKey(t, log_b, d, ra, rb, n, fft64=True), fhe.TggswKey(t, log_b, d, ra, rb, n)]
    ksa, ksb = fhe.tlwe_ksk_gen(ks_lb, ks_d, z, s, sd_lwe, 802, 0)

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
        ca, cb = fhe.tlwe_sk_encrypt(z, msgs, n_lwe, p, sd_lwe, 803, li)
        outs = [k.bootstrap(ks_lb, ks_d, ksa, ksb, v, ca, cb) for k in keys]
        for oa, ob in outs:
            for m in range(p):
                mu = ((P.tlwe_phase(zh, L(host(oa)[m]), int(host(ob)[m])) + (1 << (log_delta - 1))) % P.M64) >> log_delta
                assert mu % p == f(m) % p, (li, m, mu)
        # both modes leave the phase within a quarter of a message step of the encoded result (TOLERANCE 2^(log_delta - 2)); their raw
        # ciphertexts are NOT comparable: the first digit that rounds the other way re-randomises the masks, only the phases stay close
        for oa, ob in outs:
            for m in range(p):
                ph = (P.tlwe_phase(zh, L(host(oa)[m]), int(host(ob)[m])) - ((f(m) % p) << log_delta)) % P.M64
                ph = ph - P.M64 if ph >= 1 << 63 else ph
                assert abs(ph) < 1 << (log_delta - 2), (li, m, ph)
