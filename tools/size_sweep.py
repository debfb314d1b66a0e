#!/usr/bin/env python3
"""Forward/inverse transform throughput per ring size on one GPU (HIP events around each launch), 512 MiB of data per size.
usage: python tools/size_sweep.py [bits [two_adicity]]   (bits = 60 (default) | 55 | 54 | 45; two_adicity = 18 (default): the modulus is
the first of two_adic_primes(bits, two_adicity) and the sweep stops at N = 2^(two_adicity - 1).  Whether it is pseudo-Mersenne
eligible (2^bits - q <= 2^(bits - 33): the two-operand products) or runs on Shoup products is printed: at 54 / 55 bits only moduli of
two_adicity <= 16 or so are eligible -- the reference's own parameter sets use 11 .. 12)"""
import os
import sys
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import learn_fhe_amd as F  # noqa: E402

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 60
adic = int(sys.argv[2]) if len(sys.argv) > 2 else 18
pr = (C.c_uint64 * 1)()
assert F.lib().fhe_two_adic_primes(bits, adic, 1, pr) == 1
q = pr[0]
ctx = F.NttContext(q)
dev = torch.device("cuda:0")
print("q = %d (%d bits, 2^%d | q - 1, %s)" % (q, bits, adic, "pseudo-Mersenne eligible" if (1 << bits) - q <= 1 << (bits - 33) else "Shoup products"))
for log_n in range(6, min(18, adic)):
    n = 1 << log_n
    batch = (1 << 26) // n
    a = torch.randint(0, q, (batch, n), dtype=torch.int64, device=dev)
    for _ in range(2):
        ctx.ntt_(a, n); ctx.intt_(a, n)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = ti = 0.0
    reps = 5
    for _ in range(reps):
        ev[0].record(); ctx.ntt_(a, n); ev[1].record(); ctx.intt_(a, n); ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]); ti += ev[1].elapsed_time(ev[2])
    gb = 16.0 * n * batch / 1e9
    print("N=2^%-2d batch %7d  fwd %.3f ms %5.0f GB/s | inv %.3f ms %5.0f GB/s" % (log_n, batch, tf / reps, gb / (tf / reps * 1e-3),
                                                                                ti / reps, gb / (ti / reps * 1e-3)))
