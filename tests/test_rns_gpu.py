"""GPU parity tests for SURVEY.md section 8(a) row a14 (RNS base extension, rescale_k, CKKS key switch) and for
rings above 2^14 (radix-2^pb pass + 2^14 sub-transforms), bit-exact against the oracle."""
import math
import os
import random

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

L_ = lambda x: [int(v) for v in np.asarray(x).ravel()]  # noqa: E731


def rand_limbs(seed, mods, n, batch=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    rows = [rng.integers(0, m, size=(n if batch is None else (batch, n)), dtype=np.uint64) for m in mods]
    return np.stack(rows, axis=0 if batch is None else 1)  # [limb][n] or [batch][limb][n]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("log_n", [15, 16])
def test_big_ring_transforms(fhe, cref, torch_cuda, log_n):
    """N = 2^15 (cfg4 ring), 2^16: forward == oracle, inverse(forward) == identity, ring product == oracle"""
    n = 1 << log_n
    for q in cref.two_adic_primes(60, log_n + 1, 2) + cref.two_adic_primes(45, log_n + 1, 1):  # pseudo-Mersenne and Shoup paths
        batch = 3
        rng = np.random.Generator(np.random.PCG64(log_n))
        a = rng.integers(0, q, size=(batch, n), dtype=np.uint64)
        ctx = fhe.NttContext(q)
        d = dev(torch_cuda, a)
        ctx.ntt_(d, n)
        assert np.array_equal(host(d).reshape(-1), cref.ntt_fwd(q, a.reshape(-1), n, threads=8))
        ctx.intt_(d, n)
        assert np.array_equal(host(d), a)
    b = rng.integers(0, q, size=(1, n), dtype=np.uint64)
    da, db = dev(torch_cuda, a[:1]), dev(torch_cuda, b)
    ctx.mul_(da, db, n)
    assert np.array_equal(host(da).reshape(-1), cref.ntt_mul(q, a[0], b[0], n))


def test_rns_golden(fhe, torch_cuda):
    v = load_golden("rns.json")
    n, qs, ps = v["n"], v["qs"], v["ps"]
    U = lambda x: np.array(x, dtype=np.uint64)  # noqa: E731
    rns = fhe.RnsContext(qs, ps)
    assert host(rns.extend_bases(dev(torch_cuda, U(v["limbs"])), n))[0].tolist() == v["extended"]
    assert host(rns.rescale_k(dev(torch_cuda, U(v["full"])), n))[0].tolist() == v["rescale_k2"]
    assert rns.rescale_k(U(v["full"]), n)[0].tolist() == v["rescale_k2"]  # host-memory path
    rns1 = fhe.RnsContext(qs, ps[:1])  # K == 1 shortcut (rns.rs:108-111)
    assert host(rns1.rescale_k(dev(torch_cuda, U(v["full"][:3])), n))[0].tolist() == v["rescale_k1"]
    key = fhe.CkksKey(rns, dev(torch_cuda, U(v["ksk_b"])), dev(torch_cuda, U(v["ksk_a"])), n)
    b, a = dev(torch_cuda, U(v["ct_b"])), dev(torch_cuda, U(v["ct_a"]))
    key.key_switch_(b, a)
    assert host(b).reshape(len(qs), n).tolist() == v["ks_b"] and host(a).reshape(len(qs), n).tolist() == v["ks_a"]


@pytest.mark.parametrize("log_n", [0, 3, 6, 9])
def test_extend_rescale_vs_oracle(fhe, cref, torch_cuda, log_n):
    """util/src/ring/rns.rs:373-386 shape: 8 + 8 primes of 55 bits; batch of 3; CRT value preserved"""
    from oracle import pyref as P
    n, batch = 1 << log_n, 3
    primes = cref.two_adic_primes(55, log_n + 1, 16)
    qs, ps = primes[:8], primes[8:]
    rns = fhe.RnsContext(qs, ps)
    limbs = rand_limbs(log_n, qs, n, batch)
    ext = host(rns.extend_bases(dev(torch_cuda, limbs), n))
    for b in range(batch):
        assert np.array_equal(ext[b], cref.rns_extend_bases(qs, ps, limbs[b]))
    r0, r1 = P.Rns(qs), P.Rns(qs + ps)
    for i in range(min(n, 4)):
        full = [int(limbs[0, l, i]) for l in range(8)] + [int(ext[0, l, i]) for l in range(8)]
        assert r0.reconstruct(full[:8]) == r1.reconstruct(full)
    for k in (1, 3, 8):
        rk = fhe.RnsContext(qs, ps[:k])
        full = rand_limbs(100 + k, qs + ps[:k], n, batch)
        out = host(rk.rescale_k(dev(torch_cuda, full), n))
        for b in range(batch):
            assert np.array_equal(out[b], cref.rns_rescale_k(qs + ps[:k], k, full[b])), (k, b)


@pytest.mark.parametrize("log_n,bits,big_l", [(4, 50, 3), (10, 55, 4), (13, 60, 2)])
def test_ckks_key_switch_vs_oracle(fhe, cref, torch_cuda, log_n, bits, big_l):
    n, batch = 1 << log_n, 2
    primes = cref.two_adic_primes(bits, log_n + 1, 2 * big_l)
    qs, ps = primes[:big_l], primes[big_l:]
    rns = fhe.RnsContext(qs, ps)
    kb, ka = rand_limbs(1, qs + ps, n), rand_limbs(2, qs + ps, n)
    cb, ca = rand_limbs(3, qs, n, batch), rand_limbs(4, qs, n, batch)
    key = fhe.CkksKey(rns, dev(torch_cuda, kb), dev(torch_cuda, ka), n)
    b, a = dev(torch_cuda, cb), dev(torch_cuda, ca)
    key.key_switch_(b, a)
    for i in range(batch):
        eb, ea = cref.ckks_key_switch(qs, ps, kb, ka, cb[i], ca[i])
        assert np.array_equal(host(b)[i], eb) and np.array_equal(host(a)[i], ea), i


def test_ckks_key_switch_cfg4(fhe, cref, torch_cuda):
    """BASELINE config 4: CkksParam::new(15, 60, 8) -- N = 2^15, 8 + 8 sixty-bit primes; all 2 x 8 output limbs bit-exact"""
    g = load_golden("moduli.json")
    qs, ps = g["cfg4_qs"], g["cfg4_ps"]
    n = 1 << 15
    rns = fhe.RnsContext(qs, ps)
    kb, ka = rand_limbs(11, qs + ps, n), rand_limbs(12, qs + ps, n)
    cb, ca = rand_limbs(13, qs, n, 1), rand_limbs(14, qs, n, 1)
    key = fhe.CkksKey(rns, dev(torch_cuda, kb), dev(torch_cuda, ka), n)
    b, a = dev(torch_cuda, cb), dev(torch_cuda, ca)
    key.key_switch_(b, a)
    eb, ea = cref.ckks_key_switch(qs, ps, kb, ka, cb[0], ca[0])
    assert np.array_equal(host(b)[0], eb) and np.array_equal(host(a)[0], ea)


def test_rns_errors(fhe):
    import ctypes as C
    lib = fhe.lib()
    h = C.c_void_p()
    qs = (C.c_uint64 * 2)(1073707009, 1073707009)
    ps = (C.c_uint64 * 1)(1073692673)
    assert lib.fhe_rns_ctx_create(qs, 2, ps, 1, 0, C.byref(h)) == 1  # duplicate modulus (rns.rs:25 all_unique)
    qs = (C.c_uint64 * 2)(1073707009, 15)
    assert lib.fhe_rns_ctx_create(qs, 2, ps, 1, 0, C.byref(h)) == 2  # not prime


@pytest.mark.parametrize("log_n,bits,big_l,big_k", [(10, 55, 3, 2), (14, 60, 2, 2), (15, 60, 3, 1)])
def test_rns_evaluation_residency(fhe, cref, torch_cuda, log_n, bits, big_l, big_k):
    """util/src/ring/rns.rs:40-49, 148-158: limb-wise transforms in one launch and the evaluation-basis product, over qs and
    over qs ++ ps, against the single-modulus oracle limb by limb; the product of two RnsRq via the evaluation basis equals
    the limb-wise negacyclic ring product"""
    n, batch = 1 << log_n, 2
    primes = cref.two_adic_primes(bits, log_n + 1, big_l + big_k)
    qs, ps = primes[:big_l], primes[big_l:]
    rns = fhe.RnsContext(qs, ps)
    for extended, mods in ((False, qs), (True, qs + ps)):
        a, b = rand_limbs(1, mods, n, batch), rand_limbs(2, mods, n, batch)
        da, db = dev(torch_cuda, a), dev(torch_cuda, b)
        rns.ntt_(da, n, extended=extended)
        ev = host(da)
        for i in range(batch):
            for l, m in enumerate(mods):
                assert np.array_equal(ev[i, l], cref.ntt_fwd(m, a[i, l], n)), (extended, i, l)
        rns.ntt_(db, n, extended=extended)
        rns.pointwise_mul_(da, db, n, extended=extended)
        rns.ntt_(da, n, extended=extended, inverse=True)
        prod = host(da)
        for l, m in enumerate(mods):
            assert np.array_equal(prod[1, l], cref.ntt_mul(m, a[1, l], b[1, l], n)), (extended, l)
    with pytest.raises(fhe.FheError):
        rns.ntt_(dev(torch_cuda, rand_limbs(3, qs, 1 << 18, 1)), 1 << 18)  # no 2^19-th root of unity in these primes


SHARDED_WORKER = """
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import learn_fhe_amd as F
from learn_fhe_amd.shard import GpuLimbOps, ckks_key_switch_limb_sharded, dist_all_gather
from oracle import cref
dist.init_process_group(backend="gloo")                    # one GPU on this box: every rank drives cuda:0, gloo carries the gather
rank, world = dist.get_rank(), dist.get_world_size()
log_n = %d
n = 1 << log_n
primes = cref.two_adic_primes(60, log_n + 1, 2 * world)
qs, ps = primes[:world], primes[world:]
rng = np.random.Generator(np.random.PCG64(17))             # same seed on every rank: replicated inputs
limbs = lambda mods: np.stack([rng.integers(0, m, size=n, dtype=np.uint64) for m in mods])
ksk_b, ksk_a, ct_b, ct_a = limbs(qs + ps), limbs(qs + ps), limbs(qs), limbs(qs)
D = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
H = lambda t: t.cpu().numpy().view(np.uint64)
ops = GpuLimbOps(F, qs, ps, rank)
all_gather = dist_all_gather
kq_b, kq_a = ops.key_to_eval("q", D(ksk_b[rank]), n), ops.key_to_eval("q", D(ksk_a[rank]), n)
kp_b, kp_a = ops.key_to_eval("p", D(ksk_b[world + rank]), n), ops.key_to_eval("p", D(ksk_a[world + rank]), n)
b, a = ckks_key_switch_limb_sharded(ops, rank, world, n, D(ct_b[rank]), D(ct_a), kq_b, kq_a, kp_b, kp_a, all_gather)
eb, ea = cref.ckks_key_switch(qs, ps, ksk_b, ksk_a, ct_b, ct_a)
assert np.array_equal(H(b), eb[rank]) and np.array_equal(H(a), ea[rank]), "limb-sharded key switch != oracle"
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world,log_n", [(2, 12), (4, 15)])
def test_ckks_key_switch_limb_sharded(tmp_path, world, log_n):
    """SURVEY.md section 8(e) cfg4 partition on the device path: rank r owns q-limb r and p-limb r, ONE all-gather of the p-limb
    products; `world` processes share this box's GPU (gloo carries the gather), each rank's output limb bit-exact vs the oracle"""
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "sharded_worker.py"
    script.write_text(SHARDED_WORKER % (ROOT, log_n))
    port = str(29550 + world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", port, str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == world
