#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ring hot path on MI355X.

Metric (BASELINE.json): NTTs/sec at N = 2^14, q ~ 60-bit.  Workload (`configs[1]`): batched forward +
inverse negacyclic NTT, N = 2^14, q = 1152921504606748673, batch = 4096 polynomials per GPU resident in HBM.
One "step" = forward over the whole batch, then inverse over the whole batch (2 * 4096 transforms).
Multi-GPU: one process per GPU (torch.distributed / RCCL), polynomials sharded, no data-path collective,
weak scaling (4096 polynomials per GPU).  `python bench.py --gpus N` without a launcher starts the N ranks itself.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event
timed on the launch stream), `verified` (the timed buffers are checked: round trip == identity over the whole run,
forward == the CPU oracle on 64 polynomials) and, at N = 1, `cpu_baseline` (the oracle's C restatement timed on host
cores).  Secondary blocks (`ntt_mul`, `fhew`, `ckks`, `tfhe`) carry their own roofline (SURVEY.md 8(d) bytes) and CPU figure.
"""
import argparse
import json
import os
import socket
import subprocess
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
FWD_KERNEL = "ntt14w_fwd_kernel<ArithDS<60>, false, 3> (forward transform)"
INV_KERNEL = "ntt14w_inv_kernel<ArithDS<60>, false, false, 3> (inverse transform)"


def roof(units_per_sec, bytes_per_unit, kernel, note=None):
    """achieved = algorithmic bytes per unit x units per second, against the 8 TB/s HBM peak."""
    gbs = units_per_sec * bytes_per_unit / 1e9
    r = {"bound": "hbm", "kernel": kernel, "algorithmic_bytes_per_unit": bytes_per_unit, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": gbs / HBM_PEAK_GBS}
    if note:
        r["note"] = note
    return r


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


def ntt_mul_bench(torch, F, dev, local_rank, batch, reps=20, verify=True):
    """`Rq * Rq` (util/src/ring/fft/zq.rs:14-19) on cfg2's ring: 3 launches, 24 N algorithmic bytes (read a, b; write c)."""
    n = 1 << LOG_N
    ctx = F.NttContext(Q, device=local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(12)
    a = torch.randint(0, Q, (batch, n), dtype=torch.int64, device=dev, generator=gen)
    b = torch.randint(0, Q, (batch, n), dtype=torch.int64, device=dev, generator=gen)
    a0 = a.clone()
    dt = _timeit(torch, lambda: ctx.mul_(a, b, n), reps)
    out = {"workload": "ring product a *= b, N=2^14, q=%d, batch=%d" % (Q, batch), "products_per_sec": batch / dt,
           "hbm_traffic_bytes_per_product_by_construction": 40 * n}
    if verify:  # one launch of the timed shape on the untouched operands, first and last product against the CPU oracle
        import numpy as np
        from oracle import cref
        u = lambda t, i: t[i].cpu().numpy().view(np.uint64)  # noqa: E731
        left = {i: u(a0, i).copy() for i in (0, batch - 1)}
        ctx.mul_(a0, b, n)
        out["verified"] = bool(all(np.array_equal(u(a0, i), cref.ntt_mul(Q, left[i], u(b, i), n)) for i in (0, batch - 1)))
        out["verification"] = "ring product at batch %d: products 0 and %d bit-equal to the CPU oracle" % (batch, batch - 1)
    out["roofline"] = roof(batch / dt, 24 * n, "ntt14w_fwd_kernel (right operand, out of place) + ntt14w_mul_kernel (forward of the left operand, pointwise product, inverse: one workgroup, one load and one store)")
    return out


def fhew_setup(torch, F, dev, local_rank):
    q, n, log_b, d, w, n_lwe = 18014398509404161, 1024, 6, 9, 10, 100
    q_ks, kb, kd = 1 << 16, 4, 4
    ctx = F.NttContext(q, device=local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    rnd = lambda *shape, m=q: torch.randint(0, m, shape, dtype=torch.int64, device=dev, generator=gen)  # noqa: E731
    raw = [rnd(n_lwe, 2 * d, n), rnd(n_lwe, 2 * d, n), rnd(w + 1, d, n), rnd(w + 1, d, n)]  # brk a | b rows, ak a | b rows
    brk = F.GadgetKey(ctx, log_b, d, raw[0], raw[1], n, rgsw=True)
    ak = F.GadgetKey(ctx, log_b, d, raw[2], raw[3], n, rgsw=False)
    bk = F.BootstrapKey(ctx, brk, ak, F.ak_t(n, w), w)
    return dict(q=q, n=n, log_b=log_b, d=d, w=w, n_lwe=n_lwe, q_ks=q_ks, kb=kb, kd=kd, ctx=ctx, gen=gen, rnd=rnd, brk=brk, ak=ak, bk=bk, raw=raw,
                ksk_a=rnd(kd * n, n_lwe, m=q_ks), ksk_b=rnd(kd * n, m=q_ks), f=rnd(n))


def fhew_bench(torch, F, dev, local_rank, batches=(1, 64, 1024, 4096), reps=3, verify=True):
    """Secondary metric of BASELINE.json ("+ FHEW gate-bootstraps/sec"): BASELINE config 3 -- the full LMKCDEY blind
    rotation (bootstrapping.rs:158-209: ~100 external products + ~150 automorphism key switches per ciphertext) at
    N = 2^10, q = 18014398509404161, base 2^6, d = 9, LWE n = 100, w = 10, uniform-random keys, device resident; and the
    whole gate bootstrap around it (bootstrapping.rs:149-155: mod switch, LWE key switch over 2^16 with base 2^4 d = 4
    as in the reference's parameter sets, odd mod switch, blind rotation, sample extract)."""
    S = fhew_setup(torch, F, dev, local_rank)
    q, n, d, n_lwe, gen, rnd, bk, brk = S["q"], S["n"], S["d"], S["n_lwe"], S["gen"], S["rnd"], S["bk"], S["brk"]
    out = {"workload": "cfg3: LMKCDEY blind rotation N=2^10 q=%d log_b=6 d=9 n_lwe=100 w=10; gate = + LWE key switch "
                       "q_ks=2^16 (4,4), mod switches, sample extract" % q}
    n_auto = None
    for batch in batches:
        lwe_a = torch.randint(0, n, (batch, n_lwe), dtype=torch.int64, device=dev, generator=gen) * 2 + 1
        lwe_b = torch.randint(0, 2 * n, (batch,), dtype=torch.int64, device=dev, generator=gen)
        dt = _timeit(torch, lambda: bk.blind_rotate(lwe_a, lwe_b, S["f"]), reps)
        out["blind_rotations_per_sec_batch%d" % batch] = batch / dt
        if verify and batch == max(batches):  # the largest timed launch: first and last ciphertext against the CPU oracle
            import numpy as np
            from oracle import cref
            u = lambda t: t.cpu().numpy().view(np.uint64)  # noqa: E731
            oa, ob = bk.blind_rotate(lwe_a, lwe_b, S["f"])[:2]
            brk_h = np.stack([u(S["raw"][0]), u(S["raw"][1])], axis=1)  # [key][a|b][row][n] as the oracle takes it
            ak_h = np.stack([u(S["raw"][2]), u(S["raw"][3])], axis=1)
            ts = [int(t) for t in F.ak_t(n, S["w"])]
            ok = True
            for i in (0, batch - 1):
                ea, eb = cref.blind_rotate(q, n, S["w"], S["log_b"], d, S["log_b"], d, brk_h, ak_h, ts, u(S["f"]), u(lwe_a[i]), int(lwe_b[i]))
                ok = ok and np.array_equal(u(oa[i]), ea) and np.array_equal(u(ob[i]), eb)
            out["verified"] = bool(ok)
            out["verification"] = "blind rotation at batch %d: ciphertexts 0 and %d bit-equal to the CPU oracle" % (batch, batch - 1)
        if batch == 1024:
            # automorphism key switches per blind rotation: data dependent (bootstrapping.rs:172-231); counted on 64 ciphertexts of this batch
            _, _, sched = bk.blind_rotate(lwe_a[:64].contiguous(), lwe_b[:64].contiguous(), S["f"], want_schedule=True)
            n_auto = sum(sum(1 for kind, _ in ops if kind == "ak") for ops in sched) / float(len(sched))
    for batch in (1024, 4096):
        if batch not in batches:
            continue
        ct_a, ct_b = rnd(batch, n), rnd(batch)
        dt = _timeit(torch, lambda: bk.bootstrap(S["q_ks"], S["kb"], S["kd"], S["ksk_a"], S["ksk_b"], S["f"], ct_a, ct_b, addend=q // 8), reps)
        out["gate_bootstraps_per_sec_batch%d" % batch] = batch / dt
    ca, cb = rnd(4096, n), rnd(4096, n)
    dt = _timeit(torch, lambda: brk.external_product_(0, ca, cb), 10)
    out["external_products_per_sec_batch4096"] = 4096 / dt
    ep_bytes = (4 * d + 4) * 8 * n                      # ct in + 2d rows x (a, b) + ct out
    ks_bytes = (2 * d + 4) * 8 * n
    out["roofline_external_product"] = roof(4096 / dt, ep_bytes, "gadget_product_kernel<ArithDS<54>, WaveRing<10, 2>> (fused decompose -> NTT -> accumulate; 4 coefficients per lane)",
                                            "key rows (288 KiB per RGSW ciphertext) are L2 / Infinity-Cache hits: the kernel is bound by VALU issue, not HBM")
    if n_auto is not None:
        br_bytes = n_lwe * ep_bytes + n_auto * ks_bytes
        out["automorphism_key_switches_per_blind_rotation"] = n_auto
        top = max(batches)  # the largest resident batch (4096: the 4-coefficients-per-lane form, four waves per SIMD); batch 1024 beside it
        out["roofline"] = roof(out["blind_rotations_per_sec_batch%d" % top], br_bytes, "blind_rotate_kernel<ArithDS<54>, WaveRing<10, 2>> (batch %d)" % top,
                               "%d external products x %d B + %.1f automorphism key switches x %d B per blind rotation; keys cache resident"
                               % (n_lwe, ep_bytes, n_auto, ks_bytes))
        if 1024 in batches:
            out["roofline_batch1024"] = roof(out["blind_rotations_per_sec_batch1024"], br_bytes, "blind_rotate_kernel<ArithDS<54>, WaveRing<10, 3>> (batch 1024: one generation of 256 CUs)")
    return out


def ckks_bench(torch, F, dev, local_rank, batch=8, reps=3, verify=True):
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
    kb_raw, ka_raw = limbs(qs + ps), limbs(qs + ps)
    key = F.CkksKey(rns, kb_raw, ka_raw, n)
    out = {"workload": "cfg4: CKKS key switch N=2^15, 8+8 60-bit primes"}
    big = 32 * batch  # 256 ciphertexts: the resident batch the roofline figure is quoted on (launch ramps and tails weigh ~6 % at 64)
    for b in (batch, 8 * batch, big):
        cb, ca = limbs(qs, b), limbs(qs, b)
        cb0, ca0 = cb.clone(), ca.clone()
        dt = _timeit(torch, lambda: key.key_switch_(cb, ca), max(reps, 10 if b >= 64 else reps))
        out["key_switches_per_sec_batch%d" % b] = b / dt
        if verify and b == big:  # one launch sequence of the timed shape on the untouched inputs, first and last ciphertext against the CPU oracle
            import numpy as np
            from oracle import cref
            u = lambda t: t.cpu().numpy().view(np.uint64)  # noqa: E731
            wb, wa = cb0.clone(), ca0.clone()
            key.key_switch_(wb, wa)
            ok = True
            for i in (0, b - 1):
                eb, ea = cref.ckks_key_switch(qs, ps, u(kb_raw), u(ka_raw), u(cb0[i]), u(ca0[i]))
                ok = ok and np.array_equal(u(wb[i]), eb) and np.array_equal(u(wa[i]), ea)
            out["verified"] = bool(ok)
            out["verification"] = "key switch at batch %d: ciphertexts 0 and %d, all 2 x 8 output limbs bit-equal to the CPU oracle" % (b, b - 1)
        del cb0, ca0, cb, ca
    # `Ckks::mul` (ckks.rs:250-263: tensor + relinearisation + rescale) on the same parameter set, 7 L transforms + one key switch
    b = 2 * batch
    c4 = [limbs(qs, b) for _ in range(4)]
    dt = _timeit(torch, lambda: key.mul(*c4), reps)
    out["muls_per_sec_batch%d" % b] = b / dt
    del c4
    # ct in 2L 8N + ksk 2(L+K) 8N + ct out 2L 8N = 16 MiB at cfg4
    out["roofline"] = roof(out["key_switches_per_sec_batch%d" % big], (2 * big_l + 2 * 2 * big_l + 2 * big_l) * 8 * n,
                           "rns_extend_edge (base extension + layer 0 of the forward transforms) + ntt14w_fwd<PFX> + ntt14w_inv<PFX, MUL> (2^14 sub-transforms, two workgroups per CU; ksk products fused into the inverse's load) + rns_rescale_edge x2 (layer 0 of the inverse transforms, n^-1, rescale_k)",
                           "batch %d resident; the base conversions mix limbs, so extend / transforms / rescales stay separate passes over the limbs: ~2.7x the algorithmic bytes move" % big)
    return out


def tfhe_setup(torch, F, dev, local_rank, batch):
    n, n_lwe, log_b, d, ks_lb, ks_d = 1024, 630, 7, 3, 4, 5
    t = F.TorusContext(device=local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    rnd = lambda *shape: torch.randint(-(1 << 63), (1 << 63) - 1, shape, dtype=torch.int64, device=dev, generator=gen)  # noqa: E731
    raw = [rnd(n_lwe, 2 * d, n), rnd(n_lwe, 2 * d, n)]
    key = F.TggswKey(t, log_b, d, raw[0], raw[1], n)
    return dict(n=n, n_lwe=n_lwe, log_b=log_b, d=d, ks_lb=ks_lb, ks_d=ks_d, t=t, key=key, ksa=rnd(n * ks_d, n_lwe), ksb=rnd(n * ks_d), v=rnd(n),
                a_raw=rnd(batch, n_lwe), b_raw=rnd(batch), raw=raw)


def tfhe_bench(torch, F, dev, local_rank, batch=1024, reps=3, verify=True):
    """BASELINE config 5, one GPU's share (8192 / 8 = 1024 ciphertexts): TFHE gate bootstrap (scheme/tfhe/src/
    bootstrapping.rs:78-104) at N = 2^10, k = 1, through the single-call gate fhe_tfhe_bootstrap: mod switch, n_lwe = 630 CMUXes
    (base 2^7, d = 3 -- the reference ships no N = 2^10 parameter set; these are the usual ones for that ring), sample extract,
    TLWE key switch (base 2^4, d = 5 as in the reference's test).  Exact torus arithmetic (at this shape: key words cut into three
    pieces whose products with the digits are exact in f64 transforms -- DESIGN.md section 9), uniform-random keys."""
    S = tfhe_setup(torch, F, dev, local_rank, batch)
    dt = _timeit(torch, lambda: S["key"].bootstrap(S["ks_lb"], S["ks_d"], S["ksa"], S["ksb"], S["v"], S["a_raw"], S["b_raw"]), reps)
    out = {"workload": "cfg5 (one GPU's share): TFHE gate bootstrap N=2^10 k=1 n_lwe=630 (7,3) ks (4,5), batch=%d" % batch,
           "gate_bootstraps_per_sec": batch / dt}
    if verify:  # the timed call's outputs: first and last ciphertext against the EXACT CPU oracle (wrapping schoolbook products)
        import numpy as np
        from oracle import cref
        u = lambda t: t.cpu().numpy().view(np.uint64)  # noqa: E731
        ga, gb = S["key"].bootstrap(S["ks_lb"], S["ks_d"], S["ksa"], S["ksb"], S["v"], S["a_raw"], S["b_raw"])
        pick = [0, batch - 1]
        ea, eb = cref.tfhe_bootstrap(S["log_b"], S["d"], S["ks_lb"], S["ks_d"], u(S["raw"][0]), u(S["raw"][1]), u(S["ksa"]), u(S["ksb"]), u(S["v"]),
                                     u(S["a_raw"])[pick], u(S["b_raw"])[pick], threads=max(1, min(os.cpu_count() or 1, 16)))
        ha, hb = u(ga).reshape(batch, S["n_lwe"]), u(gb).reshape(batch)
        out["verified"] = bool(all(np.array_equal(ha[i], ea[j]) and int(hb[i]) == int(eb[j]) for j, i in enumerate(pick)))
        out["verification"] = "gate bootstrap at batch %d: ciphertexts 0 and %d bit-equal to the exact CPU oracle" % (batch, batch - 1)
    cmux_bytes = (4 * S["d"] + 4) * 8 * S["n"]  # as an RGSW external product: ct in + 2d rows x 2 + ct out
    out["roofline"] = roof(batch / dt, S["n_lwe"] * cmux_bytes, "torusx3_blind_rotate_kernel<WaveRing<9,2>> (630 CMUXes in one launch; exact: key words in three pieces through f64 transforms)",
                           "%d CMUXes x %d B per gate (TLWE key switch not counted); TGGSW rows are cache hits: bound by VALU issue" % (S["n_lwe"], cmux_bytes))
    # the same gate through the f64 FFT product the reference itself computes with (util/src/ring/fft/c64.rs; fhe_tggsw_prepare_fft64):
    # floating point, so NOT bit-exact -- checked against the exact mode's accumulators within the reference's own error bound
    fkey = F.TggswKey(S["t"], S["log_b"], S["d"], S["raw"][0], S["raw"][1], S["n"], fft64=True)
    fdt = _timeit(torch, lambda: fkey.bootstrap(S["ks_lb"], S["ks_d"], S["ksa"], S["ksb"], S["v"], S["a_raw"], S["b_raw"]), reps)
    f = {"mode": "fft64: 2d + 2 N/2-point complex f64 transforms per CMUX on the key words as they are (the reference's algorithm) where the exact mode runs 4d + 6 on three key pieces",
         "gate_bootstraps_per_sec": batch / fdt, "vs_exact_mode": dt / fdt}
    if verify:
        # ONE CMUX over the whole batch in both modes (the exact mode's gate is checked against the oracle above): every coefficient within
        # the reference's bound.  (Whole accumulators are not comparable: the first digit that rounds the other way re-randomises the masks
        # of everything after it -- decode-level equality of whole gates is what tests/test_torus_fft64_gpu.py checks, on valid keys.)
        c0a, c0b, c1a, c1b = (torch.randint(-(1 << 63), (1 << 63) - 1, (batch, S["n"]), dtype=torch.int64, device=dev) for _ in range(4))
        xa, xb = S["key"].cmux(7, c0a, c0b, c1a, c1b)
        ya, yb = fkey.cmux(7, c0a, c0b, c1a, c1b)
        err = max(int((xa - ya).abs().max().item()), int((xb - yb).abs().max().item()))  # wrapping int64 differences = signed torus distance
        bound = 2 * S["d"] * (1 << (64 + S["log_b"] + 10 - 53))
        f["max_abs_error_of_one_cmux_vs_exact_mode"] = err
        f["error_bound"] = bound
        f["verified"] = bool(0 < err <= bound)
        f["verification"] = ("one CMUX of all %d ciphertexts (2 x %d coefficients each) against the exact mode: max |difference| %d = 2^%.1f of the 2^64 torus, "
                             "bound 2d x 2^(64 + log_b + log_n - 53) (util/src/ring/fft/c64.rs:186-208 per product)" % (batch, S["n"], err, __import__("math").log2(max(err, 1))))
    f["roofline"] = roof(batch / fdt, S["n_lwe"] * cmux_bytes, "torusf_blind_rotate_kernel<WaveRing<9,2>> (630 CMUXes in one launch)", "as above")
    out["fft64"] = f
    return out


def cpu_secondary_baselines():
    """cfg3 / cfg4 / cfg5 units on host cores with the oracle's C restatement (kind "port"): one core (the reference is
    single-threaded: `single_thread_value`) and all cores of the box's share (one unit per thread: `value`), bounded samples."""
    import numpy as np
    from oracle import cref
    out = {}
    # cfg3: one blind rotation
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
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(os.cpu_count() or 1, cref.num_threads(), 16))  # a 1-GPU box's CPU share is 16 cores

    def all_cores(fn):  # one unit per thread, all threads at once (ctypes releases the GIL inside the C oracle)
        with ThreadPoolExecutor(threads) as ex:
            t0 = time.perf_counter()
            list(ex.map(lambda _: fn(), range(threads)))
            return threads / (time.perf_counter() - t0)

    cref.ntt_fwd(q, f, n)  # twiddle set-up outside the timed region
    t0 = time.perf_counter()
    cref.blind_rotate(q, n, w, log_b, d, log_b, d, brk, ak, ts, f, lwe_a, 7)
    dt = time.perf_counter() - t0
    out["fhew"] = {"blind_rotations_per_sec": all_cores(lambda: cref.blind_rotate(q, n, w, log_b, d, log_b, d, brk, ak, ts, f, lwe_a, 7)), "cores": threads,
                   "kind": "port", "single_thread_value": 1.0 / dt, "sample": "%d cfg3 blind rotations, one per thread (the reference is single-threaded: single_thread_value)" % threads}
    # cfg4: one CKKS key switch
    n4, big_l = 1 << 15, 8
    primes = cref.two_adic_primes(60, 16, 2 * big_l)
    qs, ps = primes[:big_l], primes[big_l:]
    limbs = lambda ms: np.stack([rng.integers(0, m, size=n4, dtype=np.uint64) for m in ms])  # noqa: E731
    kb, ka, cb, ca = limbs(qs + ps), limbs(qs + ps), limbs(qs), limbs(qs)
    for m in qs + ps:
        cref.ntt_fwd(m, np.zeros(n4, dtype=np.uint64), n4)  # twiddle set-up
    t0 = time.perf_counter()
    cref.ckks_key_switch(qs, ps, kb, ka, cb, ca)
    dt = time.perf_counter() - t0
    out["ckks"] = {"key_switches_per_sec": all_cores(lambda: cref.ckks_key_switch(qs, ps, kb, ka, cb, ca)), "cores": threads, "kind": "port",
                   "single_thread_value": 1.0 / dt, "sample": "%d cfg4 key switches (N=2^15, 8+8 limbs), one per thread" % threads}
    # cfg5: gate bootstraps with the reference's own floating-point product (util/src/ring/fft/c64.rs restated)
    n5, n_lwe5, lb5, d5 = 1024, 630, 7, 3
    r64 = lambda *s: rng.integers(0, 1 << 63, size=s, dtype=np.uint64) * np.uint64(2)  # noqa: E731
    bra, brb, v = r64(n_lwe5, 2 * d5, n5), r64(n_lwe5, 2 * d5, n5), r64(n5)
    ksa, ksb = r64(n5 * 5, n_lwe5), r64(n5 * 5)
    a_raw, b_raw = r64(threads, n_lwe5), r64(threads)
    cref.tfhe_bootstrap(lb5, d5, 4, 5, bra[:2], brb[:2], ksa[:, :2].copy(), ksb, v, a_raw[:1, :2].copy(), b_raw[:1], fft=True)  # twiddle set-up
    t0 = time.perf_counter()
    cref.tfhe_bootstrap(lb5, d5, 4, 5, bra, brb, ksa, ksb, v, a_raw[:1], b_raw[:1], threads=1, fft=True)
    dt1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    cref.tfhe_bootstrap(lb5, d5, 4, 5, bra, brb, ksa, ksb, v, a_raw, b_raw, threads=threads, fft=True)
    dtm = time.perf_counter() - t0
    out["tfhe"] = {"gate_bootstraps_per_sec": threads / dtm, "cores": threads, "kind": "port", "single_thread_value": 1.0 / dt1,
                   "sample": "%d cfg5 gate bootstraps, one per thread, f64 FFT products as in the reference" % threads}
    return out


def load_pmc(name="pmc_summary.json"):
    """Counter figures of the headline kernels from the committed PMC summary (profiles/pmc_summary.json), if any."""
    p = os.path.join(ROOT, "profiles", name)
    try:
        with open(p) as f:
            return json.load(f)
    except Exception:
        return {}


def issue_block(kind, measured_ms, batch, cus, pmc, isa):
    """roofline.issue: how close the kernel runs to the ceiling of its OWN vector-instruction stream.  The stream (static listing
    = dynamic: the kernels have no loops) is priced with the per-instruction issue costs of DESIGN.md section 4; a launch gives every
    SIMD batch * 8 / (4 * CUs) waves; the clock is the effective shader clock of the committed counter run (GRBM_GUI_ACTIVE / 8 /
    duration), 2.4 GHz if that run has none.  frac_of_valu_ceiling = ceiling time / measured launch time."""
    entry = next((v for k, v in isa.items() if ("ntt14w_%s_kernel" % kind) in k), None)
    if not entry:
        return None
    cycles = entry["priced_simd_cycles_per_wave"]
    clock = pmc.get("ntt_%s_effective_clock_mhz" % kind) or 2400.0
    waves_per_simd = batch * 8.0 / (4.0 * cus)
    ceiling_ms = cycles * waves_per_simd / (clock * 1e3)
    return {"valu_insts_per_wave_listing": entry["valu_insts"], "valu_insts_per_wave_counters": pmc.get("ntt_%s_valu_insts_per_wave" % kind),
            "valu_cycles_per_wave_transform": cycles, "waves_per_simd_per_launch": waves_per_simd, "clock_mhz": clock,
            "clock_source": "profiles/pmc_summary.json (GRBM_GUI_ACTIVE / 8 / duration of the profiled launches)" if pmc.get("ntt_%s_effective_clock_mhz" % kind) else "2.4 GHz specification (no counter run)",
            "valu_ceiling_ms": ceiling_ms, "measured_ms": measured_ms, "frac_of_valu_ceiling": ceiling_ms / measured_ms,
            "wave_cycle_split": pmc.get("ntt_%s_wave_cycle_split" % kind),
            "source": "profiles/ntt14w_isa.json (tools/isa_hist.py on the shipped listing) priced with DESIGN.md section 4's issue costs"}


def sharded_secondary(torch, F, dist, dev, local_rank, rank, world, backend):
    """SURVEY.md 8(e) rows 2 and 3 under the multi-GPU launch: cfg3 / cfg5 ciphertext batches split across the ranks (keys
    replicated, no collective), cfg4 with the RNS limbs sharded (L = K = world: one all-gather of the p-limb products per key
    switch, timed apart).  Rates are whole-job: units of all ranks / max time over ranks."""
    from learn_fhe_amd.shard import shard_range

    def job_rate(units_total, fn, reps):
        fn()
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(); dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return units_total * reps / float(t.item())

    out = {}
    # cfg3: 1024 ciphertexts per GPU
    S = fhew_setup(torch, F, dev, local_rank)
    total = 1024 * world
    lo, hi = shard_range(total, rank, world)
    lwe_a = torch.randint(0, S["n"], (hi - lo, S["n_lwe"]), dtype=torch.int64, device=dev, generator=S["gen"]) * 2 + 1
    lwe_b = torch.randint(0, 2 * S["n"], (hi - lo,), dtype=torch.int64, device=dev, generator=S["gen"])
    out["fhew_blind_rotations_per_sec"] = job_rate(total, lambda: S["bk"].blind_rotate(lwe_a, lwe_b, S["f"]), 3)
    # cfg5: BASELINE's 8192 ciphertexts over 8 GPUs = 1024 per GPU
    T = tfhe_setup(torch, F, dev, local_rank, 1024)
    out["tfhe_gate_bootstraps_per_sec"] = job_rate(1024 * world, lambda: T["key"].bootstrap(T["ks_lb"], T["ks_d"], T["ksa"], T["ksb"], T["v"], T["a_raw"], T["b_raw"]), 1)
    fkey = F.TggswKey(T["t"], T["log_b"], T["d"], T["raw"][0], T["raw"][1], T["n"], fft64=True)  # the reference's f64 FFT product (opt-in mode)
    out["tfhe_fft64_gate_bootstraps_per_sec"] = job_rate(1024 * world, lambda: fkey.bootstrap(T["ks_lb"], T["ks_d"], T["ksa"], T["ksb"], T["v"], T["a_raw"], T["b_raw"]), 2)
    # cfg4: limbs sharded, L = K = world, a BATCH of 64 ciphertexts per call on the library's sharded entry points (fhe_ckks_shard_*):
    # every rank holds the context and the key and owns q-limb `rank` and p-limb `rank`; ONE all-gather of the p-limb products
    # per batch (RCCL on device memory under nccl, no host synchronisation around it)
    import ctypes as C
    from learn_fhe_amd.shard import limb_slices, ckks_key_switch_sharded_batch, all_gather_into
    n, cbatch = 1 << 15, 64
    primes = (C.c_uint64 * (2 * world))()
    if F.lib().fhe_two_adic_primes(60, 16, 2 * world, primes) == 2 * world:
        qs, ps = list(primes)[:world], list(primes)[world:]
        rns = F.RnsContext(qs, ps, device=local_rank)
        gen = torch.Generator(device=dev)
        gen.manual_seed(40)  # the same key and ciphertexts on every rank (ct.a is replicated at staging)
        limbs = lambda ms, *lead: torch.stack([torch.randint(0, m, (*lead, n), dtype=torch.int64, device=dev, generator=gen) for m in ms], dim=len(lead)).contiguous()  # noqa: E731
        key = F.CkksKey(rns, limbs(qs + ps), limbs(qs + ps), n)
        q_lo, q_hi, p_lo, p_hi = limb_slices(world, world, rank, world)
        shard = F.CkksShard(key, q_lo, q_hi, p_lo, p_hi)
        ct_a_all, ct_b_all = limbs(qs, cbatch), limbs(qs, cbatch)
        ct_b = ct_b_all[:, q_lo:q_hi].contiguous()
        out["ckks_limb_sharded_key_switches_per_sec"] = job_rate(cbatch, lambda: ckks_key_switch_sharded_batch(shard, ct_b, ct_a_all), 5)
        probe = torch.zeros((2, cbatch, p_hi - p_lo, n), dtype=torch.int64, device=dev)
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            all_gather_into(probe)
        torch.cuda.synchronize()
        out["ckks_all_gather_ms_per_batch"] = (time.perf_counter() - t0) / 5 * 1e3
        out["ckks_limbs"] = "L = K = %d (one q-limb and one p-limb per rank), N = 2^15, batch %d per call" % (world, cbatch)
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` from a bare interpreter: start the N ranks ourselves (children are fresh processes; this
    parent has not touched the GPU), relay their output and exit code."""
    import torch  # device_count() does not initialise the GPU
    have = torch.cuda.device_count()
    if not args.single_device and have < args.gpus:
        print("bench.py: --gpus %d but only %d device(s) visible; refusing to report a %d-GPU number from fewer GPUs "
              "(rehearse the control path with --dist-backend gloo --single-device)" % (args.gpus, have, args.gpus), file=sys.stderr)
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=BATCH, help="polynomials per GPU (default: BASELINE cfg2)")
    ap.add_argument("--preheat-ms", type=float, default=100.0,
                    help="untimed run of the same kernels before the warm-up steps: the GPU needs ~10 ms of load to settle its clocks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true", help="also time the final all_gather of results (not in value)")
    ap.add_argument("--no-fhew", action="store_true", help="skip the secondary figures (ring product, cfg3 FHEW, cfg4 CKKS, cfg5 TFHE)")
    ap.add_argument("--no-verify-secondary", action="store_true", help="skip the oracle comparison of the secondary blocks (profiling passes)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the control path)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (only meaningful with --dist-backend gloo)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)  # does not return
    if args.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or run `python bench.py --gpus N` without a launcher)"
              % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    import torch
    import learn_fhe_amd as F
    from learn_fhe_amd.shard import shard_range

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.single_device:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        print("bench.py: rank %d has no device cuda:%d" % (rank, local_rank), file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            # RCCL carries only the barriers and the max-over-ranks reduction of one scalar (no data-path collective exists)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    n_gpus = world
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
    a_init = a.clone()  # the run is forward+inverse pairs: at the end the buffer must equal this, bit for bit

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # time-based pre-heat (outside `warmup`, outside the timed region)
    t0 = time.perf_counter()
    preheat_steps = 0
    while (time.perf_counter() - t0) * 1e3 < args.preheat_ms:
        for _ in range(8):
            ctx.ntt_(a, n)
            ctx.intt_(a, n)
        preheat_steps += 8
        torch.cuda.synchronize()
    preheat_ms = (time.perf_counter() - t0) * 1e3
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

    # ---- verification of what was timed (outside the timed region) ----
    import numpy as np
    from oracle import cref  # the checker, never the thing measured
    round_trip_ok = bool(torch.equal(a, a_init))
    chk = a_init[:64].clone()
    ctx.ntt_(chk, n)
    fwd_ok = bool(np.array_equal(chk.cpu().numpy().view(np.uint64).reshape(-1),
                                 cref.ntt_fwd(Q, a_init[:64].cpu().numpy().view(np.uint64).reshape(-1), n, threads=8)))
    verified = round_trip_ok and fwd_ok
    if dist is not None:
        v = torch.tensor([1.0 if verified else 0.0], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        verified = bool(v.item() == 1.0)

    gather_ms = None
    if args.gather and dist is not None:
        from learn_fhe_amd.shard import gather_results
        barrier()
        t0 = time.perf_counter()
        gather_results(a if args.dist_backend == "nccl" else a.cpu())
        barrier()
        gather_ms = (time.perf_counter() - t0) * 1e3

    sharded = None
    if dist is not None and not args.no_fhew:
        try:  # extras: a failure here (the same code runs on every rank, so it fails on every rank) must not cost the headline line
            sharded = sharded_secondary(torch, F, dist, dev, local_rank, rank, world, args.dist_backend)
        except Exception as e:  # noqa: BLE001
            sharded = {"error": "%s: %s" % (type(e).__name__, e)}

    secondary_ok = True
    if rank == 0:
        transforms = 2.0 * total * args.steps
        achieved = ALGO_BYTES_PER_NTT * batch / (fwd_ms * 1e-3) / 1e9
        inv_achieved = ALGO_BYTES_PER_NTT * batch / (inv_ms * 1e-3) / 1e9
        pmc = load_pmc() if args.batch == BATCH else {}
        isa = load_pmc("ntt14w_isa.json")
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        out = {
            "metric": "NTTs/sec at N=2^14 q~60-bit", "value": transforms / elapsed, "unit": "NTTs/sec",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "preheat_ms": preheat_ms,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "verified": verified,
            "verification": {"round_trip_after_%d_fwd_inv_pairs_is_identity" % (preheat_steps + args.warmup + args.steps): round_trip_ok,
                             "forward_of_64_polynomials_equals_cpu_oracle": fwd_ok},
            "config": {"workload": "cfg2: batched forward+inverse negacyclic NTT, N=2^14, q=%d, batch=%d per GPU, "
                                   "HBM-resident" % (Q, args.batch), "n": n, "q": Q, "batch_per_gpu": args.batch,
                       "parallelism": "batch-sharded x%d, no data-path collective" % n_gpus},
            # the dominant kernel of a step is the INVERSE transform (the slower of two launches of equal bytes): `kernel`, `achieved`,
            # `frac`, `traffic` are its figures; the forward's and the whole step's stand next to them
            "roofline": {"bound": "hbm", "kernel": INV_KERNEL, "achieved": inv_achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": inv_achieved / HBM_PEAK_GBS,
                         "traffic": pmc.get("ntt_inv_bytes_per_launch"), "traffic_source": "profiles/pmc_summary.json (committed counter run of this "
                         "kernel's 4096-polynomial launches, FETCH_SIZE x2 + WRITE_SIZE; not re-measured by this process)" if pmc.get("ntt_inv_bytes_per_launch") else None,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_NTT * batch,
                         "avg_launch_ms": inv_ms, "share_of_step": inv_ms / (fwd_ms + inv_ms),
                         "fwd_kernel": FWD_KERNEL, "fwd_avg_launch_ms": fwd_ms, "fwd_achieved": achieved, "fwd_frac": achieved / HBM_PEAK_GBS,
                         "fwd_traffic": pmc.get("ntt_fwd_bytes_per_launch"),
                         "step_achieved": 2 * ALGO_BYTES_PER_NTT * batch / (elapsed / args.steps) / 1e9,
                         "step_frac": 2 * ALGO_BYTES_PER_NTT * batch / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "issue": issue_block("inv", inv_ms, batch, cus, pmc, isa), "fwd_issue": issue_block("fwd", fwd_ms, batch, cus, pmc, isa),
                         "secondary_bound": "VALU issue (64-bit modular butterflies on 32-bit multipliers): see DESIGN.md 4.3"},
        }
        prop = torch.cuda.get_device_properties(dev)
        out["device"] = {"name": prop.name, "compute_units": prop.multi_processor_count, "max_clock_mhz": getattr(prop, "clock_rate", 2400000) / 1e3,  # torch builds without the field: the 2.4 GHz specification
                         "hbm_gib": round(prop.total_memory / 2 ** 30, 1), "hbm_peak_gbs_used": HBM_PEAK_GBS}
        if gather_ms is not None:
            out["final_gather_ms"] = gather_ms
        if sharded is not None:
            out["sharded"] = sharded
        if n_gpus == 1 and not args.no_fhew:
            vs = not args.no_verify_secondary
            out["ntt_mul"] = ntt_mul_bench(torch, F, dev, local_rank, min(args.batch, 2048), verify=vs)
            out["fhew"] = fhew_bench(torch, F, dev, local_rank, verify=vs)
            out["ckks"] = ckks_bench(torch, F, dev, local_rank, verify=vs)
            out["tfhe"] = tfhe_bench(torch, F, dev, local_rank, verify=vs)
            if vs:  # what the secondary blocks time is checked too: one oracle comparison per block, outside its timed region
                out["verified_secondary"] = {k: out[k].get("verified") for k in ("ntt_mul", "fhew", "ckks", "tfhe")}
                secondary_ok = all(v is True for v in out["verified_secondary"].values())
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            if not args.no_fhew:
                sec = cpu_secondary_baselines()
                out["cpu_baseline"]["fhew"] = sec["fhew"]
                out["cpu_baseline"]["ckks"] = sec["ckks"]
                out["cpu_baseline"]["tfhe"] = sec["tfhe"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not verified or not secondary_ok:
        print("bench.py: VERIFICATION FAILED (see `verification` / `verified_secondary` in the JSON line)", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
