#!/usr/bin/env python3
"""Developer lab: ring products/s at N = 2^13 .. 2^15 for a two-operand (60-bit) and a Shoup (45-bit) modulus, fused kernel on / off."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import learn_fhe_amd as F
for bits in (60, 45):
    pr = (C.c_uint64 * 1)()
    F.lib().fhe_two_adic_primes(bits, 17, 1, pr)
    ctx = F.NttContext(pr[0])
    for log_n in (13, 14, 15):
        n = 1 << log_n; batch = (1 << 25) // n
        a = torch.randint(0, pr[0], (batch, n), dtype=torch.int64, device="cuda")
        b = torch.randint(0, pr[0], (batch, n), dtype=torch.int64, device="cuda")
        res = []
        for off in (0, 1):
            F.set_option("NO_FUSED_MUL", off)
            for _ in range(3): ctx.mul_(a, b, n)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ctx.mul_(a, b, n)
            e1.record(); torch.cuda.synchronize()
            res.append(batch * 10 / (e0.elapsed_time(e1) * 1e-3))
        print("bits %d N=2^%d: fused %.3f M/s, unfused %.3f M/s" % (bits, log_n, res[0] / 1e6, res[1] / 1e6))
