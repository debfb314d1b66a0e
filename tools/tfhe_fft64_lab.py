#!/usr/bin/env python3
"""Lab: cfg5's gate bootstrap (N = 2^10, n = 630, base 2^7 x 3) through the exact mode and the fft64 mode of one key, several batches.
usage: python tools/tfhe_fft64_lab.py [batch ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import learn_fhe_amd as F  # noqa: E402

dev = torch.device("cuda:0")
batches = [int(x) for x in sys.argv[1:]] or [1024, 4096]
for batch in batches:
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
