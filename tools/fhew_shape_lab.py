#!/usr/bin/env python3
"""Developer lab: cfg3 blind rotations/s at several batch sizes; run once per shape with FHE_RING_SMALL_BATCH=0 (throughput shape
everywhere) and FHE_RING_SMALL_BATCH=1000000 (latency shape everywhere)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import learn_fhe_amd as F
dev = torch.device("cuda:0")
S = bench.fhew_setup(torch, F, dev, 0)
out = {}
for batch in [int(a) for a in sys.argv[1:]] or [768, 1024, 1536, 2048, 3072, 4096]:
    lwe_a = torch.randint(0, S["n"], (batch, S["n_lwe"]), dtype=torch.int64, device=dev, generator=S["gen"]) * 2 + 1
    lwe_b = torch.randint(0, 2 * S["n"], (batch,), dtype=torch.int64, device=dev, generator=S["gen"])
    dt = bench._timeit(torch, lambda: S["bk"].blind_rotate(lwe_a, lwe_b, S["f"]), 3)
    out[batch] = round(batch / dt)
print(os.environ.get("FHE_RING_SMALL_BATCH", "default"), json.dumps(out))
