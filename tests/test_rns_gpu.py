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


@pytest.mark.parametrize("bits,two_adicity,big_l,big_k", [(55, 10, 8, 8), (60, 16, 3, 5), (45, 17, 4, 2), (55, 12, 9, 3)])
def test_switch_bases_vs_oracle(fhe, cref, torch_cuda, bits, two_adicity, big_l, big_k):
    """util/src/ring/rns.rs:93-97 `switch_bases` in both directions through ONE context (the reference builds an `Rns` per call): the
    new limbs of extend_bases(qs -> ps) and of extend_bases(ps -> qs), bit-equal to the oracle -- pseudo-Mersenne bases of 55 and 60
    bits (the unreduced route, incl. more than eight source limbs) and primes that are not (the Shoup route)."""
    n, batch = 64, 3
    primes = cref.two_adic_primes(bits, two_adicity, big_l + big_k)
    qs, ps = primes[:big_l], primes[big_l:]
    rns = fhe.RnsContext(qs, ps)
    xq, xp = rand_limbs(bits, qs, n, batch), rand_limbs(bits + 1, ps, n, batch)
    xq[0, :, 0] = [q - 1 for q in qs]  # extremes: every residue at its maximum / zero
    xp[0, :, 1] = 0
    to_p, to_q = host(rns.switch_bases(dev(torch_cuda, xq), n)), host(rns.switch_bases(dev(torch_cuda, xp), n, to_qs=True))
    assert np.array_equal(to_p, host(rns.extend_bases(dev(torch_cuda, xq), n)))
    for b in range(batch):
        assert np.array_equal(to_p[b], cref.rns_extend_bases(qs, ps, xq[b])), b
        assert np.array_equal(to_q[b], cref.rns_extend_bases(ps, qs, xp[b])), b


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


@pytest.mark.parametrize("big_l,big_k", [(3, 3), (2, 1), (5, 2), (9, 3)])
def test_ckks_key_switch_2p15_edge_route(fhe, cref, torch_cuda, big_l, big_k):
    """N = 2^15 on 60-bit pseudo-Mersenne primes: the key switch runs layer 0 of its transforms inside the extend / rescale kernels
    (rns_kernels.hpp, edge kernels) and 2^14 sub-transforms in between.  Ragged limb counts (predicated instantiations, the K = 1
    shortcut of rescale_k, L above the register bound): bit-equal to the oracle AND to the route with whole 2^15 transforms."""
    n, batch = 1 << 15, 2
    primes = cref.two_adic_primes(60, 16, big_l + big_k)
    qs, ps = primes[:big_l], primes[big_l:]
    rns = fhe.RnsContext(qs, ps)
    kb, ka = rand_limbs(41, qs + ps, n), rand_limbs(42, qs + ps, n)
    cb, ca = rand_limbs(43, qs, n, batch), rand_limbs(44, qs, n, batch)
    key = fhe.CkksKey(rns, dev(torch_cuda, kb), dev(torch_cuda, ka), n)
    b, a = dev(torch_cuda, cb), dev(torch_cuda, ca)
    key.key_switch_(b, a)
    fhe.set_option("NO_EDGE", 1)
    try:
        b2, a2 = dev(torch_cuda, cb), dev(torch_cuda, ca)
        key.key_switch_(b2, a2)
    finally:
        fhe.set_option("NO_EDGE", 0)
    assert np.array_equal(host(b), host(b2)) and np.array_equal(host(a), host(a2))
    eb, ea = cref.ckks_key_switch(qs, ps, kb, ka, cb[0], ca[0])
    assert np.array_equal(host(b)[0], eb) and np.array_equal(host(a)[0], ea)


def test_modulus_major_dispatch_large_batches(fhe, cref, torch_cuda):
    """Launches over several moduli that span several generations of workgroups are dispatched modulus by modulus (a 2-D grid,
    ntt14w.hpp `sub_of_block`): N = 2^14 transforms of 512 x 4 limbs against the oracle on a spread sample + round trip on all of
    them, and a batch-32 key switch at N = 2^15 bit-equal to the linear dispatch order and to the oracle."""
    n, batch = 1 << 14, 512
    primes = cref.two_adic_primes(60, 16, 4)
    qs, ps = primes[:2], primes[2:]
    rns = fhe.RnsContext(qs, ps)
    a = rand_limbs(51, qs + ps, n, batch)
    d = dev(torch_cuda, a)
    rns.ntt_(d, n, extended=True)
    got = host(d)
    for b in (0, 1, 255, 510, 511):
        for l, q in enumerate(qs + ps):
            assert np.array_equal(got[b, l], cref.ntt_fwd(q, a[b, l], n)), (b, l)
    rns.ntt_(d, n, extended=True, inverse=True)
    assert np.array_equal(host(d), a)
    # the fused inverse of the key switch (2 x 32 x 4 x 2 = 512 ... x 16 limbs at cfg4's shape = 2048 sub-transforms)
    g = load_golden("moduli.json")
    qs, ps = g["cfg4_qs"], g["cfg4_ps"]
    n, batch = 1 << 15, 32
    rns = fhe.RnsContext(qs, ps)
    kb, ka = rand_limbs(52, qs + ps, n), rand_limbs(53, qs + ps, n)
    cb, ca = rand_limbs(54, qs, n, batch), rand_limbs(55, qs, n, batch)
    key = fhe.CkksKey(rns, dev(torch_cuda, kb), dev(torch_cuda, ka), n)
    b, a2 = dev(torch_cuda, cb), dev(torch_cuda, ca)
    key.key_switch_(b, a2)
    fhe.set_option("NO_LIMB_MAJOR", 1)
    try:
        b2, a3 = dev(torch_cuda, cb), dev(torch_cuda, ca)
        key.key_switch_(b2, a3)
    finally:
        fhe.set_option("NO_LIMB_MAJOR", 0)
    assert np.array_equal(host(b), host(b2)) and np.array_equal(host(a2), host(a3))
    eb, ea = cref.ckks_key_switch(qs, ps, kb, ka, cb[31], ca[31])
    assert np.array_equal(host(b)[31], eb) and np.array_equal(host(a2)[31], ea)


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


@pytest.mark.parametrize("log_n,bits,big_l,big_k,shards,batch", [(15, 60, 8, 8, 8, 3), (15, 60, 4, 4, 2, 8), (15, 60, 9, 3, 3, 2), (14, 60, 4, 2, 2, 3),
                                                                   (13, 60, 4, 4, 4, 2), (10, 55, 4, 4, 4, 3), (4, 50, 3, 3, 3, 2), (0, 50, 2, 2, 2, 2), (0, 45, 2, 2, 2, 2)])
def test_ckks_shard_entries_equal_the_key_switch(fhe, cref, torch_cuda, log_n, bits, big_l, big_k, shards, batch):
    """fhe_ckks_shard_products / fhe_ckks_shard_finish (SURVEY.md section 8(e) row 3 behind the C ABI): `shards` owners of contiguous
    q- and p-limb slices, each running its two stages on a BATCH with the fused kernels, the p-limb products gathered in between --
    every owned output limb bit-equal to fhe_ckks_key_switch on the whole ciphertext and to the oracle.  Covers the 2^15 edge route
    (layer 0 of the transforms inside extend / rescale), whole-transform 2^15 (L > 8), the split store of the wave-local inverse
    (2^13, 2^14), the pack fallback (2^10, 2^4, n = 1), ragged q-slices (9 limbs over 3 shards) and a base that is not pseudo-Mersenne."""
    from learn_fhe_amd.shard import limb_slices
    torch = torch_cuda
    n = 1 << log_n
    primes = cref.two_adic_primes(bits, max(log_n + 1, 17 if bits == 45 else 1), big_l + big_k)
    qs, ps = primes[:big_l], primes[big_l:]
    rns = fhe.RnsContext(qs, ps)
    kb, ka = rand_limbs(61, qs + ps, n), rand_limbs(62, qs + ps, n)
    cb, ca = rand_limbs(63, qs, n, batch), rand_limbs(64, qs, n, batch)
    key = fhe.CkksKey(rns, dev(torch, kb), dev(torch, ka), n)
    if bits == 45:  # two_adic_primes(45, 17): 2^45 - q is too large for the two-operand products -> no limb subsets on the Shoup route
        with pytest.raises(fhe.FheError):
            fhe.CkksShard(key, 0, 1, 0, 1)
        return
    b, a = dev(torch, cb), dev(torch, ca)
    key.key_switch_(b, a)
    owners = [fhe.CkksShard(key, *limb_slices(big_l, big_k, r, shards)) for r in range(shards)]
    d_a = dev(torch, ca)
    stage1 = [o.products(d_a) for o in owners]
    gathered = torch.stack([pp for _, pp in stage1], dim=0).contiguous()  # what the all-gather leaves on every device
    for r, o in enumerate(owners):
        q_lo, q_hi, _, _ = limb_slices(big_l, big_k, r, shards)
        ob, oa = o.finish(stage1[r][0], gathered, dev(torch, np.ascontiguousarray(cb[:, q_lo:q_hi])))
        assert np.array_equal(host(ob), host(b)[:, q_lo:q_hi]) and np.array_equal(host(oa), host(a)[:, q_lo:q_hi]), r
    eb, ea = cref.ckks_key_switch(qs, ps, kb, ka, cb[batch - 1], ca[batch - 1])
    assert np.array_equal(host(b)[batch - 1], eb) and np.array_equal(host(a)[batch - 1], ea)


SHARDED_BATCH_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch
import torch.distributed as dist
import learn_fhe_amd as F
from learn_fhe_amd.shard import limb_slices, ckks_key_switch_sharded_batch
from oracle import cref
dist.init_process_group(backend="gloo")                    # one GPU on this box: every rank drives cuda:0, gloo carries the gather
rank, world = dist.get_rank(), dist.get_world_size()
log_n, batch = %d, 8
n = 1 << log_n
primes = cref.two_adic_primes(60, log_n + 1, 2 * world)
qs, ps = primes[:world], primes[world:]
rng = np.random.Generator(np.random.PCG64(18))             # same seed on every rank: replicated inputs
limbs = lambda mods, *lead: np.stack([rng.integers(0, m, size=(*lead, n), dtype=np.uint64) for m in mods], axis=len(lead))
ksk_b, ksk_a, ct_b, ct_a = limbs(qs + ps), limbs(qs + ps), limbs(qs, batch), limbs(qs, batch)
D = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
H = lambda t: t.cpu().numpy().view(np.uint64)
rns = F.RnsContext(qs, ps)
key = F.CkksKey(rns, D(ksk_b), D(ksk_a), n)
q_lo, q_hi, p_lo, p_hi = limb_slices(world, world, rank, world)
shard = F.CkksShard(key, q_lo, q_hi, p_lo, p_hi)
b, a = ckks_key_switch_sharded_batch(shard, D(ct_b[:, q_lo:q_hi]), D(ct_a))
for c in (0, batch - 1):
    eb, ea = cref.ckks_key_switch(qs, ps, ksk_b, ksk_a, ct_b[c], ct_a[c])
    assert np.array_equal(H(b)[c], eb[q_lo:q_hi]) and np.array_equal(H(a)[c], ea[q_lo:q_hi]), "limb-sharded batch key switch != oracle"
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world,log_n", [(2, 13), (4, 15)])
def test_ckks_key_switch_limb_sharded_batch(tmp_path, world, log_n):
    """The batched, fused limb-sharded key switch (fhe_ckks_shard_*) across PROCESSES: `world` ranks share this box's GPU, gloo carries
    the one all-gather of the p-limb products (under nccl the same call is one RCCL all-gather on device memory); batch 8, every rank's
    owned limbs of the first and last ciphertext bit-exact against the oracle."""
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "sharded_batch_worker.py"
    script.write_text(SHARDED_BATCH_WORKER % (ROOT, log_n))
    port = str(29650 + world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", port, str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == world


# ---- homomorphic CKKS operations on top of the key switch ----------------------------------

@pytest.mark.parametrize("log_n,bits,big_l,big_k", [(4, 50, 3, 3), (10, 55, 4, 2), (13, 60, 2, 2), (14, 60, 3, 1)])
def test_ckks_mul_vs_oracle(fhe, cref, torch_cuda, log_n, bits, big_l, big_k):
    """`Ckks::mul` (scheme/ckks/src/ckks.rs:250-272) on random limbs, all 2 x (L-1) output limbs bit-exact"""
    n, batch = 1 << log_n, 3
    primes = cref.two_adic_primes(bits, log_n + 1, big_l + big_k)
    qs, ps = primes[:big_l], primes[big_l:]
    rns = fhe.RnsContext(qs, ps)
    kb, ka = rand_limbs(21, qs + ps, n), rand_limbs(22, qs + ps, n)
    cts = [rand_limbs(23 + i, qs, n, batch) for i in range(4)]
    key = fhe.CkksKey(rns, dev(torch_cuda, kb), dev(torch_cuda, ka), n)
    ob, oa = key.mul(*[dev(torch_cuda, c) for c in cts])
    assert tuple(ob.shape) == (batch, big_l - 1, n)
    for i in range(batch):
        eb, ea = cref.ckks_mul(qs, ps, kb, ka, *[c[i] for c in cts])
        assert np.array_equal(host(ob)[i], eb) and np.array_equal(host(oa)[i], ea), i
    # host-memory call: same result
    hb, ha = key.mul(*cts)
    assert np.array_equal(hb, host(ob)) and np.array_equal(ha, host(oa))


def test_ckks_mul_cfg4_shape(fhe, cref, torch_cuda):
    """the BASELINE cfg4 parameter set (N = 2^15, 8 + 8 sixty-bit primes): one `Ckks::mul`, all 2 x 7 output limbs"""
    g = load_golden("moduli.json")
    qs, ps = g["cfg4_qs"], g["cfg4_ps"]
    n = 1 << 15
    rns = fhe.RnsContext(qs, ps)
    kb, ka = rand_limbs(31, qs + ps, n), rand_limbs(32, qs + ps, n)
    cts = [rand_limbs(33 + i, qs, n, 1) for i in range(4)]
    key = fhe.CkksKey(rns, dev(torch_cuda, kb), dev(torch_cuda, ka), n)
    ob, oa = key.mul(*[dev(torch_cuda, c) for c in cts])
    eb, ea = cref.ckks_mul(qs, ps, kb, ka, *[c[0] for c in cts])
    assert np.array_equal(host(ob)[0], eb) and np.array_equal(host(oa)[0], ea)


@pytest.mark.parametrize("log_n,bits,big_l,big_k", [(4, 50, 3, 3), (11, 55, 4, 4), (15, 60, 2, 2)])
def test_ckks_rotate_conjugate_rescale_vs_oracle(fhe, cref, torch_cuda, log_n, bits, big_l, big_k):
    """`Ckks::rotate` / `conjugate` (ckks.rs:274-282), `RnsRq::automorphism` and `rescale()` (rns.rs:99-111)"""
    n, batch = 1 << log_n, 2
    primes = cref.two_adic_primes(bits, log_n + 1, big_l + big_k)
    qs, ps = primes[:big_l], primes[big_l:]
    rns = fhe.RnsContext(qs, ps)
    kb, ka = rand_limbs(41, qs + ps, n), rand_limbs(42, qs + ps, n)
    cb, ca = rand_limbs(43, qs, n, batch), rand_limbs(44, qs, n, batch)
    key = fhe.CkksKey(rns, dev(torch_cuda, kb), dev(torch_cuda, ka), n)
    for t in (pow(5, 3, 2 * n), -1, 2 * n + 5):
        b, a = dev(torch_cuda, cb), dev(torch_cuda, ca)
        key.rotate_(t, b, a)
        for i in range(batch):
            eb, ea = cref.ckks_rotate(qs, ps, kb, ka, t, cb[i], ca[i])
            assert np.array_equal(host(b)[i], eb) and np.array_equal(host(a)[i], ea), (t, i)
        au = rns.automorphism(dev(torch_cuda, cb), t, n)
        for i in range(batch):
            assert np.array_equal(host(au)[i], cref.rns_automorphism(qs, cb[i], t)), (t, i)
    out = rns.rescale(dev(torch_cuda, cb), n)
    for i in range(batch):
        assert np.array_equal(host(out)[i], cref.rns_rescale_k(qs, 1, cb[i])), i
    with pytest.raises(fhe.FheError):
        rns.automorphism(dev(torch_cuda, cb), 4, n)  # even t: not a permutation, refused


def test_ckks_mul_rotate_decrypt(fhe, torch_cuda):
    """Decrypt level, as the reference's own `mul` / `rotate` / `conjugate` tests (scheme/ckks/src/ckks.rs:378-420) minus the
    encoder: keys and ciphertexts made with the Python oracle (ckks.rs:139-161, 215-225), the products on the device;
    b' + a' s must be round(pt0 pt1 / q_last) resp. pt(X^t) up to the scheme's noise."""
    from oracle import pyref as P
    log_n, bits, big_l = 5, 50, 3
    n = 1 << log_n
    primes = list(P.ckks_primes(log_n, bits, big_l)) if False else None
    allp = []
    it = P.two_adic_primes(bits, log_n + 1)
    while len(allp) < 2 * big_l:
        allp.append(next(it))
    qs, ps = allp[:big_l], allp[big_l:]
    qps = qs + ps
    rnd = random.Random(77)
    big_p = math.prod(ps)
    sk = [rnd.choice([-1, 0, 0, 1]) for _ in range(n)]  # zo(0.5), ckks.rs:139-141
    lift = lambda mods, v: [[x % m for x in v] for m in mods]  # noqa: E731
    small = lambda: [rnd.randint(-6, 6) for _ in range(n)]  # noqa: E731

    def negacyclic(a, b):
        c = [0] * n
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                if i + j < n:
                    c[i + j] += x * y
                else:
                    c[i + j - n] -= x * y
        return c

    def encrypt(mods, pt_big):  # ckks.rs:215-225: b = -(a s) + e + pt
        a = [[rnd.randrange(m) for _ in range(n)] for m in mods]
        e = small()
        a_s = P.rns_mul(mods, a, lift(mods, sk))
        b = [[(-x + ee + p) % m for x, ee, p in zip(row, e, pt_big)] for m, row in zip(mods, a_s)]
        return b, a

    def decrypt(mods, b, a):  # ckks.rs:241-248, then the centred integer by CRT
        a_s = P.rns_mul(mods, a, lift(mods, sk))
        pt = [[(x + y) % m for x, y in zip(rb, ra)] for m, rb, ra in zip(mods, b, a_s)]
        big_q = math.prod(mods)
        out = []
        for i in range(n):
            v = 0
            for m, row in zip(mods, pt):
                qh = big_q // m
                v += row[i] * qh * pow(qh, -1, m)
            v %= big_q
            out.append(v - big_q if v > big_q // 2 else v)
        return out

    def ksk_gen(sk_prime):  # ckks.rs:154-161: pt = sk' * P over qs ++ ps
        return encrypt(qps, [x * big_p for x in sk_prime])

    U = lambda rows: np.array(rows, dtype=np.uint64)  # noqa: E731
    rns = fhe.RnsContext(qs, ps)
    delta = 1 << 45
    m0, m1 = small(), small()
    ct0, ct1 = encrypt(qs, [delta * x for x in m0]), encrypt(qs, [delta * x for x in m1])
    assert max(abs(x - delta * y) for x, y in zip(decrypt(qs, *ct0), m0)) < 64
    # mul with rlk = ksk_gen(sk, sk^2) (ckks.rs:163-166)
    rlk = ksk_gen(negacyclic(sk, sk))
    key = fhe.CkksKey(rns, dev(torch_cuda, U(rlk[0])), dev(torch_cuda, U(rlk[1])), n)
    ob, oa = key.mul(*[dev(torch_cuda, U(x)[None]) for x in (ct0[0], ct0[1], ct1[0], ct1[1])])
    got = decrypt(qs[:-1], [L_(r) for r in host(ob)[0]], [L_(r) for r in host(oa)[0]])
    want = negacyclic(m0, m1)
    scale2 = delta * delta / qs[-1]
    err = max(abs(g - scale2 * w) for g, w in zip(got, want))
    assert err < 2 ** 16, err  # noise: e m Delta / q_last + key-switch noise + rounding, far below Delta^2 / q_last = 2^40
    # rotate / conjugate with rtk = ksk_gen(sk, sk(X^t)) (ckks.rs:168-183)
    for t in (pow(5, 2, 2 * n), -1):
        sk_t = [x - qs[0] if x > qs[0] // 2 else x for x in P.automorphism(qs[0], [x % qs[0] for x in sk], t)]
        rtk = ksk_gen(sk_t)
        kt = fhe.CkksKey(rns, dev(torch_cuda, U(rtk[0])), dev(torch_cuda, U(rtk[1])), n)
        b, a = dev(torch_cuda, U(ct0[0])[None]), dev(torch_cuda, U(ct0[1])[None])
        kt.rotate_(t, b, a)
        got = decrypt(qs, [L_(r) for r in host(b)[0]], [L_(r) for r in host(a)[0]])
        big = 1 << 62
        want = [x - big if x > big // 2 else x for x in P.automorphism(big, [(delta * x) % big for x in m0], t)]
        assert max(abs(g - w) for g, w in zip(got, want)) < 2 ** 16, t


def test_ckks_device_keys_decrypt_level(fhe, torch_cuda):
    """`Ckks::sk_gen` / `sk_encrypt` / `rlk_gen` / `rtk_gen` / `cjk_gen` (scheme/ckks/src/ckks.rs:139-183, 215-225) on the device, then
    `mul`, `rotate`, `conjugate` on those keys; checked as the reference checks them (ckks.rs:378-420), at decrypt level on the host."""
    from oracle import pyref as P
    log_n, bits, big_l = 7, 50, 3
    n = 1 << log_n
    allp, it = [], P.two_adic_primes(bits, log_n + 1)
    while len(allp) < 2 * big_l:
        allp.append(next(it))
    qs, ps = allp[:big_l], allp[big_l:]
    rns = fhe.RnsContext(qs, ps)
    like = dev(torch_cuda, np.zeros(1, dtype=np.uint64))
    sk_d = fhe.sample_zo(0.5, 50, 0, like, n)
    sk = [int(x) for x in host(sk_d).view(np.int64)]
    assert set(sk) <= {-1, 0, 1} and 0.3 < sum(1 for x in sk if x) / n < 0.7
    lift = lambda mods, v: [[x % m for x in v] for m in mods]  # noqa: E731

    def decrypt(mods, b, a):  # ckks.rs:241-248 + centred CRT
        a_s = P.rns_mul(mods, a, lift(mods, sk))
        pt = [[(x + y) % m for x, y in zip(rb, ra)] for m, rb, ra in zip(mods, b, a_s)]
        big_q = math.prod(mods)
        out = []
        for i in range(n):
            v = sum(row[i] * (big_q // m) * pow(big_q // m, -1, m) for m, row in zip(mods, pt)) % big_q
            out.append(v - big_q if v > big_q // 2 else v)
        return out

    rows = lambda t: [L_(r) for r in host(t)]  # noqa: E731
    rnd = random.Random(3)
    delta = 1 << 45
    ms = [[rnd.randint(-5, 5) for _ in range(n)] for _ in range(2)]
    pts = np.array([[[(delta * x) % m for x in mm] for m in qs] for mm in ms], dtype=np.uint64)
    b, a = rns.sk_encrypt(sk_d, dev(torch_cuda, pts), n, 2, 51, 0)
    for c in range(2):
        got = decrypt(qs, rows(b[c]), rows(a[c]))
        assert max(abs(g - delta * w) for g, w in zip(got, ms[c])) < 64        # e <- dg(3.2, 6): |e| <= 19
    # a fresh encryption of zero is not zero, and two draws differ
    zb, za = rns.sk_encrypt(sk_d, None, n, 1, 51, 1)
    assert max(abs(x) for x in decrypt(qs, rows(zb[0]), rows(za[0]))) <= 19 and not np.array_equal(host(za[0]), host(a[0]))
    # mul on the device-made relinearisation key
    rlk = fhe.CkksKey(rns, *rns.ksk_gen(sk_d, None, n, 52, 0), n)
    ob, oa = rlk.mul(b[0:1].contiguous(), a[0:1].contiguous(), b[1:2].contiguous(), a[1:2].contiguous())
    got = decrypt(qs[:-1], rows(ob[0]), rows(oa[0]))
    want = [0] * n
    for i, x in enumerate(ms[0]):
        for j, y in enumerate(ms[1]):
            if i + j < n:
                want[i + j] += x * y
            else:
                want[i + j - n] -= x * y
    scale2 = delta * delta / qs[-1]
    assert max(abs(g - scale2 * w) for g, w in zip(got, want)) < 2 ** 18
    # rotate / conjugate on device-made keys: sk' = sk(X^t)
    for t in (pow(5, 3, 2 * n), -1):
        big = 1 << 62
        cen = lambda v: [x - big if x > big // 2 else x for x in v]  # noqa: E731
        sk_t = cen(P.automorphism(big, [x % big for x in sk], t))
        key = fhe.CkksKey(rns, *rns.ksk_gen(sk_d, dev(torch_cuda, np.array(sk_t, dtype=np.int64).view(np.uint64)), n, 53, t % 1000), n)
        rb, ra = b[0:1].clone(), a[0:1].clone()
        key.rotate_(t, rb, ra)
        got = decrypt(qs, rows(rb[0]), rows(ra[0]))
        want = cen(P.automorphism(big, [(delta * x) % big for x in ms[0]], t))
        assert max(abs(g - w) for g, w in zip(got, want)) < 2 ** 18, t


@pytest.mark.parametrize("log_n,bits,big_l,big_k,batch", [(7, 50, 3, 3, 3), (12, 55, 4, 2, 2), (14, 60, 3, 1, 2), (15, 59, 2, 1, 1)])
def test_ckks_decrypt_mul_plain_pk_encrypt(fhe, cref, torch_cuda, log_n, bits, big_l, big_k, batch):
    """`Ckks::decrypt` (scheme/ckks/src/ckks.rs:240-248: b + a sk) and `Ckks::mul_constant` after its `encode` (250-253: (pt b, pt a)
    .rescale()) bit-equal to the oracle's compositions on random operands, shared and per-ciphertext plaintexts, host and device memory,
    pt aliasing ct_b; `Ckks::pk_encrypt` (227-238) at decrypt level as the reference's `encrypt_decrypt` test checks it (ckks.rs:316-333):
    pk = `pk_gen` (143-146) on the device, decrypt(pk_encrypt(pk, pt)) - pt = e u + e1 + e0 sk, small and not zero."""
    n = 1 << log_n
    primes = cref.two_adic_primes(bits, log_n + 1, big_l + big_k)
    qs, ps = [int(x) for x in primes[:big_l]], [int(x) for x in primes[big_l:]]
    rns = fhe.RnsContext(qs, ps)
    rng = np.random.Generator(np.random.PCG64(log_n * 10 + big_l))
    sk = rng.integers(-1, 2, size=n, dtype=np.int64)
    ct_b, ct_a, pt = rand_limbs(1, qs, n, batch), rand_limbs(2, qs, n, batch), rand_limbs(3, qs, n, batch)
    sk_d = dev(torch_cuda, sk.view(np.uint64))
    got = host(rns.decrypt(sk_d, dev(torch_cuda, ct_b), dev(torch_cuda, ct_a), n))
    for c in range(batch):
        assert np.array_equal(got[c], cref.ckks_decrypt(qs, sk, ct_b[c], ct_a[c])), c
    assert np.array_equal(rns.decrypt(sk.view(np.uint64).copy(), ct_b, ct_a, n), got)       # FHE_MEM_HOST
    import ctypes as C
    from learn_fhe_amd import _lib as LL
    db, da = dev(torch_cuda, ct_b), dev(torch_cuda, ct_a)                                   # pt aliasing ct_b
    st = C.c_void_p(torch_cuda.cuda.current_stream().cuda_stream)
    LL.check(LL.lib().fhe_ckks_decrypt(rns._h, C.c_void_p(sk_d.data_ptr()), C.c_void_p(db.data_ptr()), C.c_void_p(da.data_ptr()), n, batch,
                                       C.c_void_p(db.data_ptr()), LL.MEM_DEVICE, st), "fhe_ckks_decrypt")
    assert np.array_equal(host(db), got)
    for shared in (True, False):
        p_in = np.ascontiguousarray(pt[:1]) if shared else pt
        ob, oa = rns.mul_plain(dev(torch_cuda, p_in), dev(torch_cuda, ct_b), dev(torch_cuda, ct_a), n)
        for c in range(batch):
            wb, wa = cref.ckks_mul_plain(qs, p_in[0 if shared else c], ct_b[c], ct_a[c])
            assert np.array_equal(host(ob)[c], wb) and np.array_equal(host(oa)[c], wa), (shared, c)
    # pk_gen on the device (an encryption of zero under sk), pk_encrypt, decrypt: noise = e u + e1 + e0 sk
    like = dev(torch_cuda, np.zeros(1, dtype=np.uint64))
    sk_z = fhe.sample_zo(0.5, 60, 0, like, n)
    pkb, pka = rns.sk_encrypt(sk_z, None, n, 1, 61, 0)
    eb, ea = rns.pk_encrypt(pkb[0].contiguous(), pka[0].contiguous(), dev(torch_cuda, pt), n, batch, 62, 0)
    eb2, ea2 = rns.pk_encrypt(pkb[0].contiguous(), pka[0].contiguous(), dev(torch_cuda, pt), n, batch, 62, 1)
    assert not np.array_equal(host(ea), host(ea2)) and not np.array_equal(host(ea)[0], host(ea)[-1] if batch > 1 else host(ea2)[0])
    dec = host(rns.decrypt(sk_z, eb, ea, n))
    bound = 19 * n + 19 + 19 * n                                                            # |e| <= 19, |u|, |sk| <= 1
    for c in range(batch):
        for l, q in enumerate(qs):
            diff = (dec[c, l].astype(object) - pt[c, l].astype(object)) % q
            cen = np.array([int(x) - q if int(x) > q // 2 else int(x) for x in diff], dtype=np.int64)
            assert np.abs(cen).max() <= bound and np.abs(cen).max() > 0, (c, l)
            if l:
                assert np.array_equal(cen, first), (c, l)                                   # the same integer noise on every limb
            first = cen
        assert 2.0 * np.sqrt(n) < first.std() < 8.0 * np.sqrt(n), first.std()              # ~ 3.2 sqrt(n/2 + 1 + n/2) in expectation


@pytest.mark.parametrize("log_n,bits,big_l,big_k", [(6, 50, 3, 2), (13, 60, 4, 1), (15, 59, 2, 2)])
def test_rns_add_sub_neg(fhe, cref, torch_cuda, log_n, bits, big_l, big_k):
    """`RnsRq` +, -, unary - (util/src/ring/rns.rs:254-270; `CkksCiphertext` + / - of scheme/ckks/src/ckks.rs:322-339 `add_sub`) over qs and
    over qs ++ ps, device and host memory: exact integers mod every limb, incl. the edge values 0 and q - 1; a - a = 0, -(-a) = a."""
    n, batch = 1 << log_n, 3
    primes = [int(x) for x in cref.two_adic_primes(bits, log_n + 1, big_l + big_k)]
    qs, ps = primes[:big_l], primes[big_l:]
    rns = fhe.RnsContext(qs, ps)
    for ext, mods in ((False, qs), (True, qs + ps)):
        a, b = rand_limbs(11, mods, n, batch), rand_limbs(12, mods, n, batch)
        for l, m in enumerate(mods):
            a[0, l, :4] = [0, m - 1, 0, m - 1]
            b[0, l, :4] = [0, m - 1, m - 1, 0]
        mv = np.array(mods, dtype=object)[None, :, None]
        want_add = ((a.astype(object) + b.astype(object)) % mv).astype(np.uint64)
        want_sub = ((a.astype(object) - b.astype(object)) % mv).astype(np.uint64)
        want_neg = ((-a.astype(object)) % mv).astype(np.uint64)
        assert np.array_equal(host(rns.add_(dev(torch_cuda, a), dev(torch_cuda, b), n, ext)), want_add)
        assert np.array_equal(host(rns.sub_(dev(torch_cuda, a), dev(torch_cuda, b), n, ext)), want_sub)
        assert np.array_equal(host(rns.neg_(dev(torch_cuda, a), n, ext)), want_neg)
        assert np.array_equal(rns.add_(a.copy(), b, n, ext), want_add)                       # FHE_MEM_HOST
        x = dev(torch_cuda, a)
        assert not host(rns.sub_(x, dev(torch_cuda, a), n, ext)).any()
        assert np.array_equal(host(rns.neg_(rns.neg_(dev(torch_cuda, a), n, ext), n, ext)), a)


def test_ckks_ops_edge_cases(fhe, cref, torch_cuda):
    """status codes where the reference would panic or has nothing to do: empty batches, a foreign key, a one-limb base, even t"""
    import ctypes as C
    lib = fhe.lib()
    n = 64
    primes = cref.two_adic_primes(50, 7, 5)
    rns, other = fhe.RnsContext(primes[:2], primes[2:4]), fhe.RnsContext(primes[:2], primes[3:5])
    one = fhe.RnsContext(primes[:1], primes[1:2])
    kb, ka = rand_limbs(1, primes[:4], n), rand_limbs(2, primes[:4], n)
    key = fhe.CkksKey(rns, dev(torch_cuda, kb), dev(torch_cuda, ka), n)
    z = dev(torch_cuda, np.zeros((1, 2, n), dtype=np.uint64))
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    # batch 0: nothing to do
    assert lib.fhe_ckks_mul(rns.handle, key._h, None, None, None, None, None, None, 0, 1, None) == 0
    assert lib.fhe_ckks_rotate(rns.handle, key._h, 5, None, None, 0, 1, None) == 0
    # a key prepared for another base (ckks.rs would hit the limb intersection / zip_eq panic)
    assert lib.fhe_ckks_mul(other.handle, key._h, p(z), p(z), p(z), p(z), p(z), p(z), 1, 1, None) == 1
    assert lib.fhe_ckks_rotate(other.handle, key._h, 5, p(z), p(z), 1, 1, None) == 1
    # rescale() of a one-limb polynomial would leave nothing
    assert lib.fhe_rns_rescale(one.handle, p(z), p(z), n, 1, 1, None) == 1
    # even automorphism exponents are not permutations
    with pytest.raises(fhe.FheError):
        key.rotate_(6, z.clone(), z.clone())
    # t is taken mod 2n: X -> X^(2n + 5) is X -> X^5
    cb = dev(torch_cuda, rand_limbs(5, primes[:2], n, 1))
    assert np.array_equal(host(rns.automorphism(cb, 2 * n + 5, n)), host(rns.automorphism(cb, 5, n)))
    # n = 2 (the smallest ring the reference's tests use, ckks.rs:307 log_n = 1) through the whole multiplication
    n2 = 2
    small = fhe.RnsContext(primes[:2], primes[2:4])
    k2b, k2a = rand_limbs(7, primes[:4], n2), rand_limbs(8, primes[:4], n2)
    key2 = fhe.CkksKey(small, dev(torch_cuda, k2b), dev(torch_cuda, k2a), n2)
    cts = [rand_limbs(9 + i, primes[:2], n2, 1) for i in range(4)]
    ob, oa = key2.mul(*[dev(torch_cuda, c) for c in cts])
    eb, ea = cref.ckks_mul(primes[:2], primes[2:4], k2b, k2a, *[c[0] for c in cts])
    assert np.array_equal(host(ob)[0], eb) and np.array_equal(host(oa)[0], ea)
