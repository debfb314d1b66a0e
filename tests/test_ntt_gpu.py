"""GPU parity tests of the transform path, through the C ABI, against the oracle (oracle/cref.py) on
identical seeded u64 inputs: bit-exact.  Full-size cfg2 (N = 2^14, batch 4096) is covered by
size-independent properties."""
import itertools
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu


def rand_u64(seed, q, count):
    """uniform in [0, q): numpy PCG64, seeded"""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, q, size=count, dtype=np.uint64)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def to_dev(torch, a):
    return torch.from_numpy(a.view(np.int64)).cuda()


def to_host(t):
    return t.cpu().numpy().view(np.uint64)


def test_golden_vectors(fhe, torch_cuda):
    g = load_golden("ntt.json")
    for v in g["ntt"]:
        ctx = fhe.NttContext(v["q"])
        a = np.array(v["a"], dtype=np.uint64)
        d = to_dev(torch_cuda, a)
        ctx.ntt_(d, v["n"])
        assert to_host(d).tolist() == v["ntt"]
        ctx.intt_(d, v["n"])
        assert to_host(d).tolist() == v["a"]
        h = a.copy()  # host-memory entry path
        ctx.ntt_(h, v["n"])
        assert h.tolist() == v["ntt"]
    for v in g["mul"]:
        ctx = fhe.NttContext(v["q"])
        a, b = to_dev(torch_cuda, np.array(v["a"], dtype=np.uint64)), to_dev(torch_cuda, np.array(v["b"], dtype=np.uint64))
        ctx.mul_(a, b, v["n"])
        assert to_host(a).tolist() == v["c"]
        assert to_host(b).tolist() == v["b"]  # rhs untouched
    z = np.load(os.path.join(GOLDEN, "ntt_2p14.npz"))
    ctx = fhe.NttContext(int(z["q"]))
    d = to_dev(torch_cuda, z["a"])
    ctx.ntt_(d, 1 << 14)
    assert np.array_equal(to_host(d), z["ntt"])
    ctx.intt_(d, 1 << 14)
    assert np.array_equal(to_host(d), z["a"])


@pytest.mark.parametrize("log_n", range(0, 15))
def test_forward_inverse_vs_oracle(fhe, cref, torch_cuda, log_n):
    """every supported size, three moduli widths, ragged batch (not a multiple of the polynomials per workgroup)"""
    n = 1 << log_n
    # 60-/54-bit primes of two_adic_primes are pseudo-Mersenne eligible (ntt14w.hpp: ArithDS forward, ArithPM inverse), 45-/61-bit ones are not (Shoup)
    cases = [(45, 3), (30, 2), (60, 2)] if log_n <= 12 else [(60, 2), (54, 1), (45, 1), (61, 1)]
    for bits, count in cases:
        if bits <= log_n + 1:
            continue
        for q in cref.two_adic_primes(bits, log_n + 1, count):
            batch = 67 if log_n <= 10 else 5
            a = rand_u64(1000 * log_n + bits, q, n * batch)
            a[:min(4, n)] = [0, q - 1, 1, q >> 1][:min(4, n)]
            ctx = fhe.NttContext(q)
            d = to_dev(torch_cuda, a)
            ctx.ntt_(d, n)
            exp = cref.ntt_fwd(q, a, n, threads=8)
            assert np.array_equal(to_host(d), exp), (log_n, q)
            ctx.intt_(d, n)
            assert np.array_equal(to_host(d), a), (log_n, q)
            # inverse alone on oracle-produced evaluations
            d2 = to_dev(torch_cuda, exp)
            ctx.intt_(d2, n)
            assert np.array_equal(to_host(d2), a)


@pytest.mark.parametrize("log_n", [0, 1, 3, 6, 9, 10])
def test_ring_product_vs_schoolbook(fhe, cref, torch_cuda, log_n):
    """util/src/ring.rs:443-452: a * b == nega_cyclic_schoolbook_mul(a, b)"""
    n = 1 << log_n
    for q in cref.two_adic_primes(45, log_n + 1, 3):
        a, b = rand_u64(7 + log_n, q, n), rand_u64(77 + log_n, q, n)
        ctx = fhe.NttContext(q)
        da, db = to_dev(torch_cuda, a), to_dev(torch_cuda, b)
        ctx.mul_(da, db, n)
        assert np.array_equal(to_host(da), cref.schoolbook_mul(q, a, b))


def test_extreme_values_2p14(fhe, cref, torch_cuda):
    """lazy-reduction bounds: all-(q-1), all-zero and alternating extremes through both arithmetic policies"""
    n = 1 << 14
    for q in (1152921504606748673, cref.two_adic_primes(54, 15, 1)[0], cref.two_adic_primes(45, 15, 1)[0],
              cref.two_adic_primes(61, 15, 1)[0]):
        ctx = fhe.NttContext(q)
        pats = np.zeros((4, n), dtype=np.uint64)
        pats[0, :] = q - 1
        pats[2, 0::2] = q - 1
        pats[3, :] = q >> 1
        d = to_dev(torch_cuda, pats)
        ctx.ntt_(d, n)
        assert np.array_equal(to_host(d).reshape(-1), cref.ntt_fwd(q, pats.reshape(-1), n, threads=4)), q
        ctx.intt_(d, n)
        assert np.array_equal(to_host(d), pats), q
        ev = np.full((2, n), q - 1, dtype=np.uint64)  # inverse on extreme evaluations
        ev[1, 1::2] = 0
        d = to_dev(torch_cuda, ev)
        ctx.intt_(d, n)
        assert np.array_equal(to_host(d).reshape(-1), cref.ntt_inv(q, ev.reshape(-1), n, threads=4)), q


def test_pointwise_mul(fhe, cref, torch_cuda):
    for q in (1152921504606748673, 1073707009, 18014398509404161, 97, 5):
        a, b = rand_u64(1, q, 5000), rand_u64(2, q, 5000)
        a[:3] = [0, q - 1, q - 1]
        b[:3] = [q - 1, q - 1, 1]
        ctx = fhe.NttContext(q)
        da, db = to_dev(torch_cuda, a), to_dev(torch_cuda, b)
        ctx.pointwise_mul_(da, db)
        assert np.array_equal(to_host(da), cref.pointwise_mul(q, a, b))


def test_cfg2_full_size_properties(fhe, cref, torch_cuda):
    """BASELINE config 2: N = 2^14, q = 1152921504606748673, batch 4096 (512 MiB), device resident.
    round trip == identity; forward == oracle on a 64-polynomial sample; linearity; checksum of evaluations
    (sum_k NTT(a)[k] = N * a_0 ... not used: bit-reversal keeps sums) -> sum of all evaluations == N * a[0]."""
    torch = torch_cuda
    q, n, batch = 1152921504606748673, 1 << 14, 4096
    ctx = fhe.NttContext(q)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2)
    a = torch.randint(0, q, (batch, n), dtype=torch.int64, device="cuda", generator=gen)
    ref = a.clone()
    ctx.ntt_(a, n)
    sample = list(range(0, batch, 64))
    exp = cref.ntt_fwd(q, to_host(ref[sample]).reshape(-1), n, threads=8)
    assert np.array_equal(to_host(a[sample]).reshape(-1), exp)
    # sum over all evaluation points of a(x) = N * a_0 (roots of X^N + 1 sum to zero in every power 1..N-1)
    ev = to_host(a[:8])
    for r in range(8):
        s = sum(int(v) for v in ev[r]) % q
        assert s == (n * int(to_host(ref[r])[0])) % q
    ctx.intt_(a, n)
    assert torch.equal(a, ref)
    # linearity on a slice: NTT(x + y) == NTT(x) + NTT(y) mod q
    x, y = ref[:16].clone(), ref[16:32].clone()
    xy = torch.where(x + y >= q, x + y - q, x + y)
    ctx.ntt_(xy, n); ctx.ntt_(x, n); ctx.ntt_(y, n)
    s = torch.where(x + y >= q, x + y - q, x + y)
    assert torch.equal(xy, s)


def test_error_behaviour_on_device(fhe, torch_cuda):
    ctx = fhe.NttContext(1073707009)  # s = 11 -> n <= 1024
    d = torch_cuda.zeros(2048, dtype=torch_cuda.int64, device="cuda")
    with pytest.raises(fhe.FheError) as e:
        ctx.ntt_(d, 2048)
    assert e.value.code == 3
    with pytest.raises(fhe.FheError) as e:
        ctx.ntt_(d[:24].contiguous(), 24)
    assert e.value.code == 1
    empty = torch_cuda.zeros(0, dtype=torch_cuda.int64, device="cuda")
    ctx.ntt_(empty, 8)  # empty batch is a no-op


def test_rq_elementwise_ops(fhe, torch_cuda):
    """util/src/ring.rs:328-366 (+, -, unary -, scalar *) and zq.rs:63-69 `from_i64`, any modulus < 2^62 (prime or not)"""
    for q in (1152921504606748673, 1 << 16, 12289, (1 << 62) - 57, 3):
        a, b = rand_u64(1, q, 5000), rand_u64(2, q, 5000)
        a[:4] = [0, q - 1, q - 1, 0]
        b[:4] = [0, q - 1, 1 % q, q - 1]
        ia, ib = [int(x) for x in a], [int(x) for x in b]
        da, db = to_dev(torch_cuda, a), to_dev(torch_cuda, b)
        assert to_host(fhe.rq_add(q, da, db)).tolist() == [(x + y) % q for x, y in zip(ia, ib)]
        assert to_host(fhe.rq_sub(q, da, db)).tolist() == [(x - y) % q for x, y in zip(ia, ib)]
        assert to_host(fhe.rq_neg(q, da)).tolist() == [(-x) % q for x in ia]
        s = (q - 1) if q > 3 else 2
        assert to_host(fhe.rq_scalar_mul(q, da, s)).tolist() == [x * s % q for x in ia]
        v = np.random.Generator(np.random.PCG64(3)).integers(-(1 << 62), 1 << 62, size=3000, dtype=np.int64)
        v[:5] = [0, -1, 1, -(1 << 63), (1 << 63) - 1]
        out = fhe.rq_from_i64(q, torch_cuda.from_numpy(v).cuda())
        assert to_host(out).tolist() == [int(x) % q for x in v]
    # host-memory entry and the range checks of the boundary
    h = fhe.rq_add(12289, rand_u64(4, 12289, 10), rand_u64(5, 12289, 10))
    assert h.tolist() == [(int(x) + int(y)) % 12289 for x, y in zip(rand_u64(4, 12289, 10), rand_u64(5, 12289, 10))]
    with pytest.raises(fhe.FheError):
        fhe.rq_scalar_mul(12289, rand_u64(4, 12289, 10), 12289)  # scalar not reduced
    with pytest.raises(fhe.FheError):
        fhe.rq_add(1 << 62, rand_u64(4, 7, 10), rand_u64(5, 7, 10))  # modulus out of range


def test_very_large_batches(fhe, cref, torch_cuda):
    """grid-dimension limits: hundreds of thousands of small polynomials in one call (spot-checked against the oracle), and a
    single polynomial of the largest degree"""
    q = cref.two_adic_primes(60, 18, 1)[0]
    ctx = fhe.NttContext(q)
    for n, batch in [(64, 300001), (1024, 40003), (2, 1000003)]:
        a = rand_u64(n, q, n * batch).reshape(batch, n)
        d = to_dev(torch_cuda, a)
        ctx.ntt_(d, n)
        out = to_host(d)
        for i in (0, 1, batch // 2, batch - 2, batch - 1):
            assert np.array_equal(out[i], cref.ntt_fwd(q, a[i], n)), (n, i)
        ctx.intt_(d, n)
        assert np.array_equal(to_host(d), a), n
    n = 1 << 17
    a = rand_u64(17, q, n)
    d = to_dev(torch_cuda, a.reshape(1, n))
    ctx.ntt_(d, n)
    assert np.array_equal(to_host(d)[0], cref.ntt_fwd(q, a, n))
    ctx.intt_(d, n)
    assert np.array_equal(to_host(d)[0], a)


def test_c_program_through_the_abi(tmp_path, fhe):
    """examples/c_abi_demo.c: the boundary driven from plain C (no Python, no torch in that process): BASELINE config 1's ring,
    product against a schoolbook computed in the C program, round trip, status code for a non-prime modulus"""
    import subprocess
    from conftest import ROOT
    lib_dir = os.path.dirname(fhe.lib_path())
    exe = tmp_path / "c_abi_demo"
    cmd = ["gcc", "-std=c99", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.c"), "-o", str(exe),
           "-L", lib_dir, "-lfhe_ring", "-Wl,--allow-shlib-undefined", "-Wl,-rpath," + lib_dir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c_abi_demo ok" in r.stdout, r.stdout + r.stderr


def test_c_program_sharding_over_devices(tmp_path, fhe):
    """examples/multi_gpu_demo.c: SURVEY.md 8(e) at the C boundary -- one process, one context + stream + shard per device (two
    shards on two streams of the device on a one-GPU box), asynchronous entry points, the caller's own gather; sharded result
    bit-equal to the single-device result"""
    import subprocess
    from conftest import ROOT
    lib_dir = os.path.dirname(fhe.lib_path())
    exe = tmp_path / "multi_gpu_demo"
    cmd = ["gcc", "-std=c99", "-O2", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "examples", "multi_gpu_demo.c"), "-o", str(exe), "-L", lib_dir, "-lfhe_ring", "-L", "/opt/rocm/lib", "-lamdhip64",
           "-Wl,--allow-shlib-undefined", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "multi_gpu_demo ok" in r.stdout, r.stdout + r.stderr


def test_c_program_limb_sharded_key_switch(tmp_path, fhe):
    """examples/multi_gpu_ckks_demo.c: SURVEY.md 8(e) row 3 at the C boundary -- cfg4's key switch with its limbs sharded over the
    devices of ONE process (two shards on two streams of the device on a one-GPU box): fhe_ckks_shard_products, the caller's own
    gather by hipMemcpyPeerAsync + events (no host synchronisation in between), fhe_ckks_shard_finish; every output limb bit-equal
    to the single-device fhe_ckks_key_switch"""
    import subprocess
    from conftest import ROOT
    lib_dir = os.path.dirname(fhe.lib_path())
    exe = tmp_path / "multi_gpu_ckks_demo"
    cmd = ["gcc", "-std=c99", "-O2", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "examples", "multi_gpu_ckks_demo.c"), "-o", str(exe), "-L", lib_dir, "-lfhe_ring", "-L", "/opt/rocm/lib", "-lamdhip64",
           "-Wl,--allow-shlib-undefined", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "multi_gpu_ckks_demo ok" in r.stdout, r.stdout + r.stderr


def test_scratch_pool_is_private_and_trimmable(fhe, torch_cuda):
    """device scratch comes from the library's own stream-ordered pool (api_common.hpp), not the device's default pool: a
    host-memory call and a workspace-using device call leave the default pool's release threshold alone; fhe_trim() succeeds"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    pool, thr = C.c_void_p(), C.c_uint64(0)
    assert hip.hipDeviceGetDefaultMemPool(C.byref(pool), 0) == 0
    HIP_MEMPOOL_ATTR_RELEASE_THRESHOLD = 4  # hipMemPoolAttrReleaseThreshold
    assert hip.hipMemPoolGetAttribute(pool, HIP_MEMPOOL_ATTR_RELEASE_THRESHOLD, C.byref(thr)) == 0
    before = thr.value
    q, n = 1073707009, 1024
    ctx = fhe.NttContext(q)
    a, b = rand_u64(5, q, 3 * n), rand_u64(6, q, 3 * n)
    ctx.mul_(a.copy(), b, n)                       # host operands: mirrors + workspace
    ctx.mul_(to_dev(torch_cuda, a), to_dev(torch_cuda, b), n)
    torch_cuda.cuda.synchronize()
    assert hip.hipMemPoolGetAttribute(pool, HIP_MEMPOOL_ATTR_RELEASE_THRESHOLD, C.byref(thr)) == 0 and thr.value == before
    assert fhe.lib().fhe_trim() == 0


@pytest.mark.parametrize("log_n", [12, 13, 14, 15])
def test_alternative_routes_agree(fhe, cref, torch_cuda, log_n):
    """2^12 / 2^13 rings run the wave-local kernels (R0 = 1 / 2) and ring products at 2^13 .. 2^15 the fused forward-multiply-inverse
    kernel; the library keeps the older routes behind lab switches (fhe_set_option): both must give the same bits, and the oracle's."""
    import torch
    n, batch = 1 << log_n, 5
    for q in (cref.two_adic_primes(60, 17, 1)[0], cref.two_adic_primes(54, 17, 1)[0], cref.two_adic_primes(45, 17, 1)[0]):  # two-operand products at 60 bits; Shoup
        rng = np.random.Generator(np.random.PCG64(log_n))
        a = rng.integers(0, q, size=(batch, n), dtype=np.uint64)
        b = rng.integers(0, q, size=(batch, n), dtype=np.uint64)
        a[0, :] = q - 1
        b[0, :] = q - 1
        ctx = fhe.NttContext(q)
        outs = []
        for env in ({}, {"NO_W12": 1, "NO_FUSED_MUL": 1}):
            for k, v in env.items():
                fhe.set_option(k, v)
            f = torch.from_numpy(a.view(np.int64)).cuda()
            ctx.ntt_(f, n)
            m = torch.from_numpy(a.view(np.int64)).cuda()
            ctx.mul_(m, torch.from_numpy(b.view(np.int64)).cuda(), n)
            r = f.clone()
            ctx.intt_(r, n)
            outs.append((f.cpu().numpy().view(np.uint64), m.cpu().numpy().view(np.uint64), r.cpu().numpy().view(np.uint64)))
            for k in env:
                fhe.set_option(k, 0)
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        assert np.array_equal(outs[0][2], a) and np.array_equal(outs[1][2], a)
        assert np.array_equal(outs[0][0][1], cref.ntt_fwd(q, a[1], n))
        assert np.array_equal(outs[0][1][0], cref.ntt_mul(q, a[0], b[0], n))
