import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, learn_fhe_amd as F
Q = 1152921504606748673; n = 1 << 14; batch = 4096
ctx = F.NttContext(Q)
a = torch.randint(0, Q, (batch, n), dtype=torch.int64, device="cuda")
for _ in range(3): ctx.ntt_(a, n); ctx.intt_(a, n)
for mode in ("events3", "noevents", "events_every4", "events3", "noevents"):
    steps = 40
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(steps):
        rec = mode == "events3" or (mode == "events_every4" and s % 4 == 0)
        if rec: ev[s][0].record()
        ctx.ntt_(a, n)
        if rec: ev[s][1].record()
        ctx.intt_(a, n)
        if rec: ev[s][2].record()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "ms/step %.4f" % (dt / steps * 1e3), "NTT/s %.3e" % (2 * batch * steps / dt))
