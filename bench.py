#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ring hot path on MI355X.

Metric (BASELINE.json): NTTs/sec at N = 2^14, q ~ 60-bit.  Workload (`configs[1]`): batched forward +
inverse negacyclic NTT, N = 2^14, q = 1152921504606748673, batch = 4096 polynomials per GPU resident in HBM.
One "step" = forward over the whole batch, then inverse over the whole batch (2 * 4096 transforms).
Multi-GPU: one process per GPU (torch.distributed / RCCL), polynomials sharded, no data-path collective,
weak scaling (4096 polynomials per GPU).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event
timed on the launch stream) and, at N = 1, `cpu_baseline` (the oracle's C restatement timed on host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

Q = 1152921504606748673  # two_adic_primes(60, 15).next()
LOG_N = 14
BATCH = 4096
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec peak
ALGO_BYTES_PER_NTT = 16 * (1 << LOG_N)  # 8N read + 8N write


def cpu_baseline(sample_per_thread=48):
    """Oracle C restatement (kind = "port": same u128 % q arithmetic and loop nest as the Rust reference,
    which cannot be built here) on the host cores of this box: forward+inverse over a bounded sample."""
    import numpy as np
    from oracle import cref
    n = 1 << LOG_N
    threads = max(1, min(os.cpu_count() or 1, cref.num_threads(), 16))  # a 1-GPU box's CPU share is 16 cores
    rng = np.random.Generator(np.random.PCG64(2))
    # single thread (the reference's execution model)
    a1 = rng.integers(0, Q, size=n * 16, dtype=np.uint64)
    cref.ntt_fwd_inplace(Q, a1[:n].copy(), n, 1)  # twiddle set-up outside the timed region
    t0 = time.perf_counter()
    cref.ntt_fwd_inplace(Q, a1, n, 1)
    cref.ntt_inv_inplace(Q, a1, n, 1)
    t1 = time.perf_counter() - t0
    single = 2 * 16 / t1
    polys = sample_per_thread * threads
    a = rng.integers(0, Q, size=n * polys, dtype=np.uint64)
    t0 = time.perf_counter()
    cref.ntt_fwd_inplace(Q, a, n, threads)
    cref.ntt_inv_inplace(Q, a, n, threads)
    tm = time.perf_counter() - t0
    return {"value": 2 * polys / tm, "unit": "NTTs/sec", "cores": threads, "kind": "port",
            "sample": "%d polynomials (fwd+inv) of the bench workload, OpenMP over the batch; "
                      "single-thread: %.1f NTTs/sec on 16 polynomials" % (polys, single),
            "single_thread_value": single}


def fhew_bench(torch, F, dev, local_rank, batches=(64, 1024), reps=3):
    """Secondary metric of BASELINE.json ("+ FHEW gate-bootstraps/sec"): BASELINE config 3 -- the full LMKCDEY blind
    rotation (bootstrapping.rs:158-209: ~100 external products + ~150 automorphism key switches per ciphertext) at
    N = 2^10, q = 18014398509404161, base 2^6, d = 9, LWE n = 100, w = 10, uniform-random keys, device resident."""
    from oracle import pyref as P  # only ak_t(): the exponent list [-5, 5, 25, ...] mod 2N
    q, n, log_b, d, w, n_lwe = 18014398509404161, 1024, 6, 9, 10, 100
    ctx = F.NttContext(q, device=local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    rnd = lambda *shape: torch.randint(0, q, shape, dtype=torch.int64, device=dev, generator=gen)  # noqa: E731
    brk = F.GadgetKey(ctx, log_b, d, rnd(n_lwe, 2 * d, n), rnd(n_lwe, 2 * d, n), n, rgsw=True)
    ak = F.GadgetKey(ctx, log_b, d, rnd(w + 1, d, n), rnd(w + 1, d, n), n, rgsw=False)
    bk = F.BootstrapKey(ctx, brk, ak, P.ak_t(n, w), w)
    f = rnd(n)
    out = {"workload": "cfg3: LMKCDEY blind rotation N=2^10 q=%d log_b=6 d=9 n_lwe=100 w=10" % q}
    for batch in batches:
        lwe_a = torch.randint(0, n, (batch, n_lwe), dtype=torch.int64, device=dev, generator=gen) * 2 + 1
        lwe_b = torch.randint(0, 2 * n, (batch,), dtype=torch.int64, device=dev, generator=gen)
        bk.blind_rotate(lwe_a, lwe_b, f)  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            bk.blind_rotate(lwe_a, lwe_b, f)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out["blind_rotations_per_sec_batch%d" % batch] = batch / dt
    ca, cb = rnd(4096, n), rnd(4096, n)
    brk.external_product_(0, ca, cb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        brk.external_product_(i % n_lwe, ca, cb)
    torch.cuda.synchronize()
    out["external_products_per_sec_batch4096"] = 10 * 4096 / (time.perf_counter() - t0)
    return out


def load_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (profiles/), if any."""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        with open(p) as f:
            return json.load(f).get("ntt_fwd_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="polynomials per GPU (default: BASELINE cfg2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true", help="also time the final all_gather of results (not in value)")
    ap.add_argument("--no-fhew", action="store_true", help="skip the secondary FHEW blind-rotation figures")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the control path)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (only meaningful with --dist-backend gloo)")
    args = ap.parse_args()

    import torch
    import learn_fhe_amd as F
    from learn_fhe_amd.shard import shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.single_device:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using %d" % (args.gpus, world, n_gpus), file=sys.stderr)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    n = 1 << LOG_N
    total = args.batch * n_gpus  # weak scaling: fixed work per GPU
    lo, hi = shard_range(total, rank, n_gpus)
    batch = hi - lo
    ctx = F.NttContext(Q, device=local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(2 + rank)
    a = torch.randint(0, Q, (batch, n), dtype=torch.int64, device=dev, generator=gen)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.ntt_(a, n)
        ctx.intt_(a, n)
    # per-kernel HIP-event timing on the launch stream (torch's current stream is the one handed to the library)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        ev[s][0].record()
        ctx.ntt_(a, n)
        ev[s][1].record()
        ctx.intt_(a, n)
        ev[s][2].record()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fwd_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    inv_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps

    gather_ms = None
    if args.gather and dist is not None:
        from learn_fhe_amd.shard import gather_results
        barrier()
        t0 = time.perf_counter()
        gather_results(a if args.dist_backend == "nccl" else a.cpu())
        barrier()
        gather_ms = (time.perf_counter() - t0) * 1e3

    if rank == 0:
        transforms = 2.0 * total * args.steps
        achieved = ALGO_BYTES_PER_NTT * batch / (fwd_ms * 1e-3) / 1e9
        out = {
            "metric": "NTTs/sec at N=2^14 q~60-bit", "value": transforms / elapsed, "unit": "NTTs/sec",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "cfg2: batched forward+inverse negacyclic NTT, N=2^14, q=%d, batch=%d per GPU, "
                                   "HBM-resident" % (Q, args.batch), "n": n, "q": Q, "batch_per_gpu": args.batch,
                       "parallelism": "batch-sharded x%d, no data-path collective" % n_gpus},
            "roofline": {"bound": "hbm", "kernel": "ntt_fwd_kernel<ArithPM<60>,14,4,1> (forward transform)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": load_traffic() if args.batch == BATCH else None, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_NTT * batch,
                         "avg_launch_ms": fwd_ms, "inv_avg_launch_ms": inv_ms,
                         "inv_achieved": ALGO_BYTES_PER_NTT * batch / (inv_ms * 1e-3) / 1e9},
        }
        if gather_ms is not None:
            out["final_gather_ms"] = gather_ms
        if n_gpus == 1 and not args.no_fhew:
            out["fhew"] = fhew_bench(torch, F, dev, local_rank)
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
