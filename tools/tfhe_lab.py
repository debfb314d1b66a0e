#!/usr/bin/env python3
"""Developer lab: cfg5 gate bootstraps/s (bench.py's tfhe block alone), packed-digit kernel on / off."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import learn_fhe_amd as F
dev = torch.device("cuda", 0)
for batch in [int(a) for a in sys.argv[1:]] or [1024, 1280, 4096]:
    S = bench.tfhe_setup(torch, F, dev, 0, batch)
    fn = lambda: S["key"].bootstrap(S["ks_lb"], S["ks_d"], S["ksa"], S["ksb"], S["v"], S["a_raw"], S["b_raw"])  # noqa: E731
    out = {"batch": batch}
    for name, off in (("packed", 0), ("unpacked", 1), ("packed_again", 0)):
        F.set_option("NO_PACKED_DIGITS", off)
        out[name] = round(batch / bench._timeit(torch, fn, 3))
    F.set_option("NO_PACKED_DIGITS", 0)
    print(json.dumps(out))
