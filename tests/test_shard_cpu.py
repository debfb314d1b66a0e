"""Multi-GPU path, rehearsed on CPU: world_size-2 gloo ranks shard a batch of independent polynomials with the same
helper bench.py uses (learn-fhe_amd/shard.py), transform their shard (oracle stands in for the device here), and the
final gather reproduces the single-process result.  No data-path collective exists; only the final gather."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT


def test_shard_range_partition():
    from learn_fhe_amd.shard import shard_range
    for total in (0, 1, 7, 8, 4096, 4097):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from learn_fhe_amd.shard import shard_range, gather_results
    from oracle import cref
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    q, n, total = 1073707009, 64, 10
    rng = np.random.Generator(np.random.PCG64(5))
    full = rng.integers(0, q, size=(total, n), dtype=np.uint64)
    lo, hi = shard_range(total, rank, world)
    mine = cref.ntt_fwd(q, full[lo:hi].reshape(-1), n).reshape(hi - lo, n)
    # equal-shape gather: pad ragged shards to the largest shard
    width = -(-total // world)
    pad = np.zeros((width, n), dtype=np.uint64); pad[: hi - lo] = mine
    out = gather_results(torch.from_numpy(pad.view(np.int64))).numpy().view(np.uint64)
    rows = [out[r * width: r * width + (shard_range(total, r, world)[1] - shard_range(total, r, world)[0])] for r in range(world)]
    got = np.concatenate(rows, axis=0)
    exp = cref.ntt_fwd(q, full.reshape(-1), n).reshape(total, n)
    assert np.array_equal(got, exp), "gathered shards != single-process result"
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)   # the max-over-ranks timing reduction bench.py uses
    assert t.item() == world
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


CKKS_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from oracle import cref
    from learn_fhe_amd.shard import ckks_key_switch_limb_sharded
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    log_n, n = 6, 64
    primes = cref.two_adic_primes(50, log_n + 1, 2 * world)
    qs, ps = primes[:world], primes[world:]
    rng = np.random.Generator(np.random.PCG64(7))          # same seed on every rank: replicated inputs
    limbs = lambda mods: np.stack([rng.integers(0, m, size=n, dtype=np.uint64) for m in mods])
    ksk_b, ksk_a, ct_b, ct_a = limbs(qs + ps), limbs(qs + ps), limbs(qs), limbs(qs)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64))
    U = lambda t: t.numpy().view(np.uint64)

    class OracleOps:  # the arithmetic of one rank, from the CPU oracle: this test checks the ORCHESTRATION (who computes what,
        def extend_to_my_p(self, a_all, n):  # what is exchanged), the GPU arithmetic has its own parity tests
            return T(cref.rns_extend_bases(qs, [ps[rank]], U(a_all))[0])
        def key_to_eval(self, which, k, n):
            return T(cref.ntt_fwd(qs[rank] if which == "q" else ps[rank], U(k), n))
        def limb_product(self, which, x, key_eval, n):
            m = qs[rank] if which == "q" else ps[rank]
            return T(cref.ntt_inv(m, cref.pointwise_mul(m, cref.ntt_fwd(m, U(x), n), U(key_eval)), n))
        def rescale_my_q(self, x_q, x_p_all, n):
            return T(cref.rns_rescale_k([qs[rank]] + ps, len(ps), np.concatenate([U(x_q).reshape(1, n), U(x_p_all).reshape(len(ps), n)]))[0])
        def add_my_q(self, x, y):
            return T(np.array([(int(p) + int(q)) %% qs[rank] for p, q in zip(U(x), U(y))], dtype=np.uint64))

    ops = OracleOps()
    def all_gather(x):
        out = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(out, x.contiguous())
        return torch.stack(out, dim=0)
    kq_b, kq_a = ops.key_to_eval("q", T(ksk_b[rank]), n), ops.key_to_eval("q", T(ksk_a[rank]), n)
    kp_b, kp_a = ops.key_to_eval("p", T(ksk_b[world + rank]), n), ops.key_to_eval("p", T(ksk_a[world + rank]), n)
    b, a = ckks_key_switch_limb_sharded(ops, rank, world, n, T(ct_b[rank]), T(ct_a), kq_b, kq_a, kp_b, kp_a, all_gather)
    eb, ea = cref.ckks_key_switch(qs, ps, ksk_b, ksk_a, ct_b, ct_a)
    assert np.array_equal(U(b), eb[rank]) and np.array_equal(U(a), ea[rank]), "limb-sharded key switch != single-process result"
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def test_two_rank_gloo_limb_sharded_ckks_key_switch(tmp_path):
    """SURVEY.md section 8(e), cfg4 partition: rank r owns q-limb r and p-limb r; one all-gather of the p-limb products"""
    script = tmp_path / "ckks_worker.py"
    script.write_text(CKKS_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


BOOTSTRAP_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from oracle import cref
    from learn_fhe_amd.shard import shard_range, gather_results
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.Generator(np.random.PCG64(11))          # same seed on every rank: keys are REPLICATED, the batch is split
    r64 = lambda *s: rng.integers(0, 1 << 63, size=s, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=s, dtype=np.uint64)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64))
    # cfg5 shape at toy size: TFHE gate bootstraps (mod switch, CMUX chain, sample extract, key switch), 7 ciphertexts over 2 ranks
    n, n_lwe, log_b, d, total = 32, 5, 10, 2, 7
    bra, brb, v = r64(n_lwe, 2 * d, n), r64(n_lwe, 2 * d, n), r64(n)
    ksa, ksb = r64(n * 5, n_lwe), r64(n * 5)
    a_raw, b_raw = r64(total, n_lwe), r64(total)
    lo, hi = shard_range(total, rank, world)
    ga, gb = cref.tfhe_bootstrap(log_b, d, 4, 5, bra, brb, ksa, ksb, v, a_raw[lo:hi], b_raw[lo:hi])
    width = -(-total // world)
    pad = np.zeros((width, n_lwe + 1), dtype=np.uint64); pad[: hi - lo, :n_lwe] = ga; pad[: hi - lo, n_lwe] = gb
    out = gather_results(T(pad)).numpy().view(np.uint64)   # the ONLY collective: the final gather of the output LWE ciphertexts
    rows = [out[r * width: r * width + (shard_range(total, r, world)[1] - shard_range(total, r, world)[0])] for r in range(world)]
    got = np.concatenate(rows, axis=0)
    ea, eb = cref.tfhe_bootstrap(log_b, d, 4, 5, bra, brb, ksa, ksb, v, a_raw, b_raw)
    assert np.array_equal(got[:, :n_lwe], ea) and np.array_equal(got[:, n_lwe], eb), "sharded TFHE gates != single-process result"
    # cfg3 shape at toy size: LMKCDEY blind rotations, 5 ciphertexts over 2 ranks
    q, n3, w, lb3, d3, nl3 = 1073707009, 16, 2, 6, 5, 4
    assert cref.is_prime(q)
    brk = rng.integers(0, q, size=(nl3, 2, 2 * d3, n3), dtype=np.uint64)
    ak = rng.integers(0, q, size=(w + 1, 2, d3, n3), dtype=np.uint64)
    f = rng.integers(0, q, size=n3, dtype=np.uint64)
    ts, x = [], 1
    for _ in range(w):
        x = x * 5 %% (2 * n3); ts.append(x if x < n3 else x - 2 * n3)
    ts = [-5] + ts
    tot3 = 5
    lwe_a = rng.integers(0, n3, size=(tot3, nl3), dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    lwe_b = rng.integers(0, 2 * n3, size=tot3, dtype=np.uint64)
    rot = lambda i: np.concatenate(cref.blind_rotate(q, n3, w, lb3, d3, lb3, d3, brk, ak, ts, f, lwe_a[i], int(lwe_b[i])))
    lo, hi = shard_range(tot3, rank, world)
    width = -(-tot3 // world)
    pad = np.zeros((width, 2 * n3), dtype=np.uint64)
    for j, i in enumerate(range(lo, hi)):
        pad[j] = rot(i)
    out = gather_results(T(pad)).numpy().view(np.uint64)
    rows = [out[r * width: r * width + (shard_range(tot3, r, world)[1] - shard_range(tot3, r, world)[0])] for r in range(world)]
    assert np.array_equal(np.concatenate(rows, axis=0), np.stack([rot(i) for i in range(tot3)])), "sharded blind rotations != single-process result"
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def test_two_rank_gloo_sharded_bootstraps(tmp_path):
    """SURVEY.md section 8(e) row 2 (cfg3 / cfg5): a batch of bootstraps split across ranks with the helper bench.py uses, keys
    replicated, no data-path collective, final gather of the outputs; ragged batches (7 over 2, 5 over 2)"""
    script = tmp_path / "bootstrap_worker.py"
    script.write_text(BOOTSTRAP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29545", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29545", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2
