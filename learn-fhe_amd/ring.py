"""Harness-side mirror of the reference's ring API (util/src/ring.rs `Rq`, util/src/ring/fft/zq.rs) on top
of the C ABI.  Accepts numpy uint64 arrays (host path) or torch int64/uint64 CUDA tensors (device path,
zero-copy: data_ptr() and the current torch stream are handed to the library)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _buf(x):
    """-> (pointer, element count, mem kind, stream handle)"""
    if _is_torch(x):
        import torch
        assert x.is_cuda and x.is_contiguous() and x.element_size() == 8
        return C.c_void_p(x.data_ptr()), x.numel(), L.MEM_DEVICE, C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    assert isinstance(x, np.ndarray) and x.dtype == np.uint64 and x.flags["C_CONTIGUOUS"]
    return C.c_void_p(x.ctypes.data), x.size, L.MEM_HOST, C.c_void_p(0)


class NttContext:
    """One prime modulus (the reference's per-q twiddle cache entry, util/src/ring/fft/zq.rs:49-56)."""

    def __init__(self, q: int, device: int = 0):
        self._h = C.c_void_p()
        self.q, self.device = q, device
        L.check(L.lib().fhe_ctx_create(C.c_uint64(q), device, C.byref(self._h)), "fhe_ctx_create(q=%d)" % q)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            L.lib().fhe_ctx_destroy(h)

    @property
    def handle(self):
        return self._h

    def info(self):
        q, s, g, w = C.c_uint64(), C.c_int(), C.c_uint64(), C.c_uint64()
        L.check(L.lib().fhe_ctx_info(self._h, C.byref(q), C.byref(s), C.byref(g), C.byref(w)), "fhe_ctx_info")
        return {"q": q.value, "s": s.value, "g": g.value, "omega": w.value}

    def twiddles(self, count, inverse=False):
        out = np.zeros(count, dtype=np.uint64)
        L.check(L.lib().fhe_ctx_twiddles(self._h, int(inverse), out.ctypes.data_as(L.u64p), count), "fhe_ctx_twiddles")
        return out

    # in-place transforms over [batch][n]
    def ntt_(self, a, n):
        p, cnt, mem, st = _buf(a)
        L.check(L.lib().fhe_ntt_fwd(self._h, p, n, cnt // n, mem, st), "fhe_ntt_fwd")
        return a

    def intt_(self, a, n):
        p, cnt, mem, st = _buf(a)
        L.check(L.lib().fhe_ntt_inv(self._h, p, n, cnt // n, mem, st), "fhe_ntt_inv")
        return a

    def mul_(self, a, b, n):
        """a <- a * b in Z_q[X]/(X^n+1) (ring.rs:256-264)."""
        pa, cnt, mem, st = _buf(a)
        pb, cntb, memb, _ = _buf(b)
        assert cnt == cntb and mem == memb
        L.check(L.lib().fhe_ntt_mul(self._h, pa, pb, n, cnt // n, mem, st), "fhe_ntt_mul")
        return a

    def pointwise_mul_(self, a, b):
        pa, cnt, mem, st = _buf(a)
        pb, cntb, memb, _ = _buf(b)
        assert cnt == cntb and mem == memb
        L.check(L.lib().fhe_pointwise_mul(self._h, pa, pb, cnt, mem, st), "fhe_pointwise_mul")
        return a
