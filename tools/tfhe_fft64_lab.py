#!/usr/bin/env python3
"""Lab: cfg5's gate bootstrap (N = 2^10, n = 630, base 2^7 x 3) through the exact mode and the fft64 mode of one key, several batches.
usage: python tools/tfhe_fft64_lab.py [--ref] [batch ...]   (--ref: the reference's own parameter set instead of cfg5's)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import learn_fhe_amd as F  # noqa: E402

dev = torch.device("cuda:0")
args = [a for a in sys.argv[1:] if a != "--ref"]
batches = [int(x) for x in args] or [1024, 4096]
for batch in batches:
    if "--ref" in sys.argv:  # the reference's own bootstrap set (scheme/tfhe/src/bootstrapping.rs:139-152): N = 2048, n = 1024, base 2^23 x 1
        n, n_lwe, log_b, d, ks_lb, ks_d = 2048, 1024, 23, 1, 4, 5
        t = F.TorusContext(device=0)
        rnd = lambda *shape: torch.randint(-(1 << 63), (1 << 63) - 1, shape, dtype=torch.int64, device=dev)  # noqa: E731
        raw = [rnd(n_lwe, 2 * d, n), rnd(n_lwe, 2 * d, n)]
        S = dict(n=n, n_lwe=n_lwe, log_b=log_b, d=d, ks_lb=ks_lb, ks_d=ks_d, t=t, key=F.TggswKey(t, log_b, d, raw[0], raw[1], n), ksa=rnd(n * ks_d, n_lwe),
                 ksb=rnd(n * ks_d), v=rnd(n), a_raw=rnd(batch, n_lwe), b_raw=rnd(batch), raw=raw)
    else:
        S = bench.tfhe_setup(torch, F, dev, 0, batch)
    keys = {"exact": S["key"], "fft64": F.TggswKey(S["t"], S["log_b"], S["d"], S["raw"][0], S["raw"][1], S["n"], fft64=True)}
    for name, key in keys.items():
        fn = lambda: key.bootstrap(S["ks_lb"], S["ks_d"], S["ksa"], S["ksb"], S["v"], S["a_raw"], S["b_raw"])  # noqa: E731
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("batch %5d %-6s %8.1f gates/s  (%.2f ms)" % (batch, name, batch / dt, dt * 1e3), flush=True)
