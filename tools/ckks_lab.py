#!/usr/bin/env python3
"""Developer lab: the cfg4 key switch alone (N = 2^15, 8 + 8 sixty-bit primes), `--batch` ciphertexts, `--reps` repetitions.
Prints key switches/s; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (tools/scripts/prof_ckks.sh)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import bench  # noqa: E402
import learn_fhe_amd as F  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--bits", type=int, default=60)
args = ap.parse_args()
dev = torch.device("cuda", 0)
import ctypes as C  # noqa: E402
n, big_l = 1 << 15, 8
primes = (C.c_uint64 * (2 * big_l))()
assert F.lib().fhe_two_adic_primes(args.bits, 16, 2 * big_l, primes) == 2 * big_l
qs, ps = list(primes)[:big_l], list(primes)[big_l:]
rns = F.RnsContext(qs, ps, device=0)
gen = torch.Generator(device=dev)
gen.manual_seed(4)
limbs = lambda ms, *lead: torch.stack([torch.randint(0, m, (*lead, n), dtype=torch.int64, device=dev, generator=gen) for m in ms], dim=len(lead)).contiguous()  # noqa: E731
key = F.CkksKey(rns, limbs(qs + ps), limbs(qs + ps), n)
cb, ca = limbs(qs, args.batch), limbs(qs, args.batch)
for _ in range(3):
    key.key_switch_(cb, ca)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.reps):
    key.key_switch_(cb, ca)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.reps
print(json.dumps({"batch": args.batch, "bits": args.bits, "key_switches_per_sec": args.batch / dt, "us_per_batch": dt * 1e6,
                  "roofline_frac": args.batch / dt * 16 * 2 ** 20 / 8e12}))
