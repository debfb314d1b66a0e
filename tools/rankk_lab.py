"""Lab: whole gate bootstraps through the rank-k entries (torusk_api.hip) at a few shapes.  usage: python tools/rankk_lab.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import learn_fhe_amd as F
dev = torch.device("cuda:0")
rnd = lambda *shape: torch.randint(-(1 << 63), (1 << 63) - 1, shape, dtype=torch.int64, device=dev)
t = F.TorusContext()
for k, n, n_lwe, log_b, d in [(2, 512, 630, 7, 3), (1, 1024, 630, 7, 3), (2, 256, 64, 8, 8)]:
    for batch in (256, 1024):
        k1 = k + 1
        key = F.TggswKeyK(t, k, log_b, d, rnd(n_lwe, k1 * d, k1, n), n)
        ksa, ksb, v, a, b = rnd(k * n * 5, n_lwe), rnd(k * n * 5), rnd(n), rnd(batch, n_lwe), rnd(batch)
        fn = lambda: key.bootstrap(4, 5, ksa, ksb, v, a, b)
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("k=%d N=%d n_lwe=%d (%d,%d) batch %d: %.1f gates/s (%.1f ms)" % (k, n, n_lwe, log_b, d, batch, batch / dt, dt * 1e3), flush=True)
