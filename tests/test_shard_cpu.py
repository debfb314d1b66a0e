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
