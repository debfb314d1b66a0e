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


def cpu_baseline(sample_per_thread=512, reps=4):
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
    for _ in range(reps):
        cref.ntt_fwd_inplace(Q, a, n, threads)
        cref.ntt_inv_inplace(Q, a, n, threads)
    tm = (time.perf_counter() - t0) / reps
    return {"value": 2 * polys / tm, "unit": "NTTs/sec", "cores": threads, "kind": "port",
            "sample": "%d polynomials (fwd+inv) x %d repetitions of the bench workload, OpenMP over the batch; "
                      "single-thread: %.1f NTTs/sec on 16 polynomials" % (polys, reps, single),
            "single_thread_value": single}


def _timeit(torch, fn, reps):
    fn()  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def fhew_bench(torch, F, dev, local_rank, batches=(1, 64, 1024, 4096), reps=3):
    """Secondary metric of BASELINE.json ("+ FHEW gate-bootstraps/sec"): BASELINE config 3 -- the full LMKCDEY blind
    rotation (bootstrapping.rs:158-209: ~100 external products + ~150 automorphism key switches per ciphertext) at
    N = 2^10, q = 18014398509404161, base 2^6, d = 9, LWE n = 100, w = 10, uniform-random keys, device resident; and the
    whole gate bootstrap around it (bootstrapping.rs:149-155: mod switch, LWE key switch over 2^16 with base 2^4 d = 4
    as in the reference's parameter sets, odd mod switch, blind rotation, sample extract)."""
    q, n, log_b, d, w, n_lwe = 18014398509404161, 1024, 6, 9, 10, 100
    q_ks, kb, kd = 1 << 16, 4, 4
    ctx = F.NttContext(q, device=local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    rnd = lambda *shape, m=q: torch.randint(0, m, shape, dtype=torch.int64, device=dev, generator=gen)  # noqa: E731
    brk = F.GadgetKey(ctx, log_b, d, rnd(n_lwe, 2 * d, n), rnd(n_lwe, 2 * d, n), n, rgsw=True)
    ak = F.GadgetKey(ctx, log_b, d, rnd(w + 1, d, n), rnd(w + 1, d, n), n, rgsw=False)
    bk = F.BootstrapKey(ctx, brk, ak, F.ak_t(n, w), w)
    ksk_a, ksk_b = rnd(kd * n, n_lwe, m=q_ks), rnd(kd * n, m=q_ks)
    f = rnd(n)
    out = {"workload": "cfg3: LMKCDEY blind rotation N=2^10 q=%d log_b=6 d=9 n_lwe=100 w=10; gate = + LWE key switch "
                       "q_ks=2^16 (4,4), mod switches, sample extract" % q}
    for batch in batches:
        lwe_a = torch.randint(0, n, (batch, n_lwe), dtype=torch.int64, device=dev, generator=gen) * 2 + 1
        lwe_b = torch.randint(0, 2 * n, (batch,), dtype=torch.int64, device=dev, generator=gen)
        dt = _timeit(torch, lambda: bk.blind_rotate(lwe_a, lwe_b, f), reps)
        out["blind_rotations_per_sec_batch%d" % batch] = batch / dt
    batch = 1024
    ct_a, ct_b = rnd(batch, n), rnd(batch)
    dt = _timeit(torch, lambda: bk.bootstrap(q_ks, kb, kd, ksk_a, ksk_b, f, ct_a, ct_b, addend=q // 8), reps)
    out["gate_bootstraps_per_sec_batch%d" % batch] = batch / dt
    ca, cb = rnd(4096, n), rnd(4096, n)
    dt = _timeit(torch, lambda: brk.external_product_(0, ca, cb), 10)
    out["external_products_per_sec_batch4096"] = 4096 / dt
    return out


def ckks_bench(torch, F, dev, local_rank, batch=8, reps=3):
    """BASELINE config 4 on one GPU: CKKS key switch (scheme/ckks/src/ckks.rs:284-293) with CkksParam::new(15, 60, 8):
    N = 2^15, 8 + 8 sixty-bit primes (two_adic_primes(60, 16)), `batch` ciphertexts resident in HBM."""
    import ctypes as C
    n, big_l = 1 << 15, 8
    primes = (C.c_uint64 * (2 * big_l))()
    assert F.lib().fhe_two_adic_primes(60, 16, 2 * big_l, primes) == 2 * big_l
    qs, ps = list(primes)[:big_l], list(primes)[big_l:]
    rns = F.RnsContext(qs, ps, device=local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(4)
    limbs = lambda ms, *lead: torch.stack([torch.randint(0, m, (*lead, n), dtype=torch.int64, device=dev, generator=gen)  # noqa: E731
                                           for m in ms], dim=len(lead)).contiguous()
    key = F.CkksKey(rns, limbs(qs + ps), limbs(qs + ps), n)
    out = {"workload": "cfg4: CKKS key switch N=2^15, 8+8 60-bit primes"}
    for b in (batch, 8 * batch):
        cb, ca = limbs(qs, b), limbs(qs, b)
        dt = _timeit(torch, lambda: key.key_switch_(cb, ca), reps)
        out["key_switches_per_sec_batch%d" % b] = b / dt
    return out


def tfhe_bench(torch, F, dev, local_rank, batch=1024, reps=1):
    """BASELINE config 5, one GPU's share (8192 / 8 = 1024 ciphertexts): TFHE gate bootstrap (scheme/tfhe/src/
    bootstrapping.rs:139-165) at N = 2^10, k = 1: mod switch, n_lwe = 630 CMUXes (base 2^7, d = 3 -- the reference ships
    no N = 2^10 parameter set; these are the usual ones for that ring), sample extract, TLWE key switch (base 2^4, d = 5
    as in the reference's test).  Exact torus arithmetic (CRT over three 30-bit primes at this shape), uniform-random keys."""
    n, n_lwe, log_b, d, ks_lb, ks_d = 1024, 630, 7, 3, 4, 5
    t = F.TorusContext(device=local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    rnd = lambda *shape: torch.randint(-(1 << 63), (1 << 63) - 1, shape, dtype=torch.int64, device=dev, generator=gen)  # noqa: E731
    key = F.TggswKey(t, log_b, d, rnd(n_lwe, 2 * d, n), rnd(n_lwe, 2 * d, n), n)
    ksa, ksb = rnd(n * ks_d, n_lwe), rnd(n * ks_d)
    v = rnd(n)
    a_raw, b_raw = rnd(batch, n_lwe), rnd(batch)

    def gate():
        at, bt = F.TorusContext.mod_switch(a_raw, n), F.TorusContext.mod_switch(b_raw, n)
        oa, ob = key.blind_rotate(at, bt, v)
        ea, eb = F.tglwe_sample_extract(oa, ob, n, 0)
        return F.tlwe_key_switch(ks_lb, ks_d, ksa, ksb, ea, eb, n, n_lwe)

    dt = _timeit(torch, gate, reps)
    return {"workload": "cfg5 (one GPU's share): TFHE gate bootstrap N=2^10 k=1 n_lwe=630 (7,3) ks (4,5), batch=%d" % batch,
            "gate_bootstraps_per_sec": batch / dt}


def cpu_fhew_baseline():
    """cfg3 blind rotation of ONE ciphertext on one host core with the oracle's C restatement (the reference is single-threaded)."""
    import numpy as np
    from oracle import cref
    q, n, log_b, d, w, n_lwe = 18014398509404161, 1024, 6, 9, 10, 100
    rng = np.random.Generator(np.random.PCG64(6))
    brk = rng.integers(0, q, size=(n_lwe, 2, 2 * d, n), dtype=np.uint64)
    ak = rng.integers(0, q, size=(w + 1, 2, d, n), dtype=np.uint64)
    f = rng.integers(0, q, size=n, dtype=np.uint64)
    lwe_a = rng.integers(0, n, size=n_lwe, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    ts = []
    x, q2 = 1, 2 * n
    for _ in range(w):
        x = x * 5 % q2
        ts.append(x if x < q2 // 2 else x - q2)
    ts = [-5] + ts
    cref.ntt_fwd(q, f, n)  # twiddle set-up outside the timed region
    t0 = time.perf_counter()
    cref.blind_rotate(q, n, w, log_b, d, log_b, d, brk, ak, ts, f, lwe_a, 7)
    dt = time.perf_counter() - t0
    return {"blind_rotations_per_sec": 1.0 / dt, "cores": 1, "kind": "port", "sample": "1 cfg3 blind rotation"}


def load_pmc(key):
    """A figure of the dominant kernel from the committed PMC summary (profiles/pmc_summary.json), if any."""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        with open(p) as f:
            return json.load(f).get(key)
    except Exception:
        return None


def load_traffic():
    return load_pmc("ntt_fwd_bytes_per_launch")


def issue_roofline(torch, dev, batch, fwd_ms):
    """Secondary, for interpretation only (hard part 1): how close the forward kernel runs to the integer
    ISSUE ceiling of its own instruction stream -- VALU instructions per wave (SQ_INSTS_VALU / SQ_WAVES from the committed
    counter run) x 8 waves per polynomial, one wave-instruction per SIMD every ~4.3 cycles (4 for plain integer ops, ~4.7 for
    v_mad_u64_u32: tools/microbench_intmul.hip), 4 SIMDs per CU at the device's maximum clock."""
    insts = load_pmc("ntt_fwd_valu_insts_per_wave")
    if not insts:
        return None
    prop = torch.cuda.get_device_properties(dev)
    clock_hz = getattr(prop, "clock_rate", 2400000) * 1e3
    simds = prop.multi_processor_count * 4
    ceiling_ms = insts * 8 * batch * 4.3 / (simds * clock_hz) * 1e3
    return {"valu_insts_per_wave": insts, "ceiling_ms_at_max_clock": ceiling_ms, "frac_of_issue_ceiling": ceiling_ms / fwd_ms,
            "max_clock_mhz": clock_hz / 1e6, "simds": simds}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # ~0.15 s timed: the GPU needs ~10 ms of load to reach steady clocks
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=BATCH, help="polynomials per GPU (default: BASELINE cfg2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true", help="also time the final all_gather of results (not in value)")
    ap.add_argument("--no-fhew", action="store_true", help="skip the secondary figures (cfg3 FHEW, cfg4 CKKS, cfg5 TFHE)")
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
            # RCCL carries only the barriers and the max-over-ranks reduction of one scalar (no data-path collective exists)
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
            "roofline": {"bound": "hbm", "kernel": "ntt14_fwd_kernel<ArithPM<60>> (forward transform)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": load_traffic() if args.batch == BATCH else None, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_NTT * batch,
                         "avg_launch_ms": fwd_ms, "inv_avg_launch_ms": inv_ms,
                         "inv_achieved": ALGO_BYTES_PER_NTT * batch / (inv_ms * 1e-3) / 1e9,
                         "issue": issue_roofline(torch, dev, batch, fwd_ms)},
        }
        prop = torch.cuda.get_device_properties(dev)
        out["device"] = {"name": prop.name, "compute_units": prop.multi_processor_count, "max_clock_mhz": getattr(prop, "clock_rate", 2400000) / 1e3,  # torch builds without the field: the 2.4 GHz specification
                         "hbm_gib": round(prop.total_memory / 2 ** 30, 1), "hbm_peak_gbs_used": HBM_PEAK_GBS}
        if gather_ms is not None:
            out["final_gather_ms"] = gather_ms
        if n_gpus == 1 and not args.no_fhew:
            out["fhew"] = fhew_bench(torch, F, dev, local_rank)
            out["ckks"] = ckks_bench(torch, F, dev, local_rank)
            out["tfhe"] = tfhe_bench(torch, F, dev, local_rank)
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            if not args.no_fhew:
                out["cpu_baseline"]["fhew"] = cpu_fhew_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
