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
        if h and L is not None and getattr(L, "lib", None):  # (module globals are gone at interpreter shutdown)
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


# ---- rows a8-a13: decomposition, automorphism, gadget keys, blind rotation -------------------------------


def _like(x, shape):
    if _is_torch(x):
        import torch
        return torch.empty(shape, dtype=x.dtype, device=x.device)
    return np.empty(shape, dtype=np.uint64)


def decompose(q, log_b, d, a, n):
    """util/src/misc/decompose.rs:42-46: [polys][n] -> [polys][d][n]."""
    p, cnt, mem, st = _buf(a)
    polys = cnt // n
    out = _like(a, (polys, d, n))
    po, _, _, _ = _buf(out)
    L.check(L.lib().fhe_decompose(q, log_b, d, p, n, polys, po, mem, st), "fhe_decompose")
    return out


def automorphism(q, t, a, n):
    """util/src/avec.rs:34-50."""
    p, cnt, mem, st = _buf(a)
    out = _like(a, tuple(a.shape))
    po, _, _, _ = _buf(out)
    L.check(L.lib().fhe_automorphism(q, t, p, po, n, cnt // n, mem, st), "fhe_automorphism")
    return out


def monomial_mul(q, k, a, n):
    """util/src/ring.rs:299-313."""
    p, cnt, mem, st = _buf(a)
    out = _like(a, tuple(a.shape))
    po, _, _, _ = _buf(out)
    L.check(L.lib().fhe_monomial_mul(q, k, p, po, n, cnt // n, mem, st), "fhe_monomial_mul")
    return out


class GadgetKey:
    """Prepared (evaluation-domain, device-resident) RGSW ciphertexts or RLWE key-switching keys."""

    def __init__(self, ctx: NttContext, log_b, d, rows_a, rows_b, n, rgsw: bool):
        self.ctx, self.log_b, self.d, self.n, self.rgsw = ctx, log_b, d, n, rgsw
        pa, cnt, mem, _ = _buf(rows_a)
        pb, cntb, memb, _ = _buf(rows_b)
        rows = (2 * d if rgsw else d)
        assert cnt == cntb and mem == memb and cnt % (rows * n) == 0
        self.count = cnt // (rows * n)
        self._h = C.c_void_p()
        fn = L.lib().fhe_rgsw_prepare if rgsw else L.lib().fhe_ksk_prepare
        L.check(fn(ctx.handle, log_b, d, pa, pb, n, self.count, mem, C.byref(self._h)), "fhe_%s_prepare" % ("rgsw" if rgsw else "ksk"))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and L is not None and getattr(L, "lib", None):  # (module globals are gone at interpreter shutdown)
            L.lib().fhe_key_destroy(h)

    @property
    def handle(self):
        return self._h

    def external_product_(self, index, ct_a, ct_b):
        """scheme/fhew/src/rgsw.rs:116-128, in place on [batch][n] a/b."""
        pa, cnt, mem, st = _buf(ct_a)
        pb, _, _, _ = _buf(ct_b)
        L.check(L.lib().fhe_external_product(self.ctx.handle, self._h, index, pa, pb, cnt // self.n, mem, st), "fhe_external_product")

    def internal_product_(self, index, ct1_a, ct1_b):
        """scheme/fhew/src/rgsw.rs:130-150 with ct0 = this key's entry `index`; ct1_a / ct1_b: [count][2d][n], in place."""
        pa, cnt, mem, st = _buf(ct1_a)
        pb, _, _, _ = _buf(ct1_b)
        L.check(L.lib().fhe_rgsw_internal_product(self.ctx.handle, self._h, index, pa, pb, cnt // (2 * self.d * self.n), mem, st),
                "fhe_rgsw_internal_product")

    def key_switch_(self, index, ct_a, ct_b):
        """scheme/fhew/src/rlwe.rs:177-186."""
        pa, cnt, mem, st = _buf(ct_a)
        pb, _, _, _ = _buf(ct_b)
        L.check(L.lib().fhe_rlwe_key_switch(self.ctx.handle, self._h, index, pa, pb, cnt // self.n, mem, st), "fhe_rlwe_key_switch")

    def automorphism_(self, index, t, ct_a, ct_b):
        """scheme/fhew/src/rlwe.rs:188-191."""
        pa, cnt, mem, st = _buf(ct_a)
        pb, _, _, _ = _buf(ct_b)
        L.check(L.lib().fhe_rlwe_automorphism(self.ctx.handle, self._h, index, t, pa, pb, cnt // self.n, mem, st),
                "fhe_rlwe_automorphism")


def _rq_binary(name, q, a, b):
    pa, cnt, mem, st = _buf(a)
    pb, _, _, _ = _buf(b)
    out = _like(a, tuple(a.shape))
    po, _, _, _ = _buf(out)
    L.check(getattr(L.lib(), name)(q, pa, pb, po, cnt, mem, st), name)
    return out


def rq_add(q, a, b):
    """util/src/ring.rs:343-350 `Rq + Rq` (either basis), any q < 2^62."""
    return _rq_binary("fhe_rq_add", q, a, b)


def rq_sub(q, a, b):
    """util/src/ring.rs:351-358."""
    return _rq_binary("fhe_rq_sub", q, a, b)


def rq_neg(q, a):
    pa, cnt, mem, st = _buf(a)
    out = _like(a, tuple(a.shape))
    po, _, _, _ = _buf(out)
    L.check(L.lib().fhe_rq_neg(q, pa, po, cnt, mem, st), "fhe_rq_neg")
    return out


def rq_scalar_mul(q, a, scalar):
    """util/src/ring.rs:359-366 `Rq *= Zq`."""
    pa, cnt, mem, st = _buf(a)
    out = _like(a, tuple(a.shape))
    po, _, _, _ = _buf(out)
    L.check(L.lib().fhe_rq_scalar_mul(q, pa, scalar, po, cnt, mem, st), "fhe_rq_scalar_mul")
    return out


def rq_from_i64(q, a):
    """util/src/zq.rs:63-69 over an array of i64 (torch int64 tensor, or numpy int64 viewed as uint64)."""
    pa, cnt, mem, st = _buf(a)
    out = _like(a, tuple(a.shape))
    po, _, _, _ = _buf(out)
    L.check(L.lib().fhe_rq_from_i64(q, pa, po, cnt, mem, st), "fhe_rq_from_i64")
    return out


def lwe_mod_switch(q, q_prime, v, odd=False):
    """util/src/zq.rs:128-140 (scheme/fhew/src/lwe.rs:90-99)."""
    p, cnt, mem, st = _buf(v)
    out = _like(v, tuple(v.shape))
    po, _, _, _ = _buf(out)
    L.check(L.lib().fhe_lwe_mod_switch(q, q_prime, p, po, cnt, int(odd), mem, st), "fhe_lwe_mod_switch")
    return out


def lwe_key_switch(q, log_b, d, ksk_a, ksk_b, ct_a, ct_b, n_in, n_out):
    """scheme/fhew/src/lwe.rs:151-160."""
    pka, _, mem, st = _buf(ksk_a)
    pkb, _, _, _ = _buf(ksk_b)
    pa, cnt, _, _ = _buf(ct_a)
    pb, _, _, _ = _buf(ct_b)
    batch = cnt // n_in
    out_a, out_b = _like(ct_a, (batch, n_out)), _like(ct_a, (batch,))
    poa, _, _, _ = _buf(out_a)
    pob, _, _, _ = _buf(out_b)
    L.check(L.lib().fhe_lwe_key_switch(q, log_b, d, pka, pkb, pa, pb, n_in, n_out, poa, pob, batch, mem, st), "fhe_lwe_key_switch")
    return out_a, out_b


def lwe_lincomb(q, coefs, xs, addend=0):
    """scheme/fhew/src/lwe.rs:22-75: sum_k coefs[k] * xs[k] + addend over q (element-wise)."""
    k = len(xs)
    bufs = [_buf(x) for x in xs]
    ptrs = (C.c_void_p * k)(*[b[0] for b in bufs])
    cf = (C.c_int64 * k)(*coefs)
    out = _like(xs[0], tuple(xs[0].shape))
    po, _, _, _ = _buf(out)
    L.check(L.lib().fhe_lwe_lincomb(q, k, cf, ptrs, addend, po, bufs[0][1], bufs[0][2], bufs[0][3]), "fhe_lwe_lincomb")
    return out


def rlwe_sample_extract(q, ct_a, ct_b, n, index, addend=0):
    """scheme/fhew/src/rlwe.rs:193-202."""
    pa, cnt, mem, st = _buf(ct_a)
    pb, _, _, _ = _buf(ct_b)
    batch = cnt // n
    out_a, out_b = _like(ct_a, (batch, n)), _like(ct_a, (batch,))
    poa, _, _, _ = _buf(out_a)
    pob, _, _, _ = _buf(out_b)
    L.check(L.lib().fhe_rlwe_sample_extract(q, pa, pb, n, index, addend, poa, pob, batch, mem, st), "fhe_rlwe_sample_extract")
    return out_a, out_b


def ak_t(n, w, g=5):
    """scheme/fhew/src/bootstrapping.rs:86-89: the automorphism exponents [-g, g, g^2 .. g^w] mod 2N as signed values."""
    q2 = 2 * n
    c = lambda v: v - q2 if v >= q2 // 2 else v  # noqa: E731  Zq -> i64 (centered, util/src/zq.rs:240-249)
    out, x = [c((q2 - g) % q2)], 1
    for _ in range(w):
        x = x * g % q2
        out.append(c(x))
    return out


class BootstrapKey:
    """scheme/fhew/src/bootstrapping.rs:93-113 (brk + ak part)."""

    def __init__(self, ctx: NttContext, brk: GadgetKey, ak: GadgetKey, ak_t, w):
        self.ctx, self.brk, self.ak, self.w = ctx, brk, ak, w
        t = (C.c_int64 * len(ak_t))(*ak_t)
        self._h = C.c_void_p()
        L.check(L.lib().fhe_bootstrap_key_create(ctx.handle, brk.handle, ak.handle, t, w, C.byref(self._h)), "fhe_bootstrap_key_create")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and L is not None and getattr(L, "lib", None):  # (module globals are gone at interpreter shutdown)
            L.lib().fhe_bootstrap_key_destroy(h)

    def bootstrap(self, q_ks, ks_log_b, ks_d, lwe_ksk_a, lwe_ksk_b, f, ct_a, ct_b, addend=0):
        """scheme/fhew/src/bootstrapping.rs:149-155 for a batch: ct_a [batch][N], ct_b [batch] over Q -> LWE (a [batch][N], b)."""
        n = self.brk.n
        pka, _, mem, st = _buf(lwe_ksk_a)
        pkb, _, _, _ = _buf(lwe_ksk_b)
        pf, fcnt, _, _ = _buf(f)
        pa, cnt, _, _ = _buf(ct_a)
        pb, batch, _, _ = _buf(ct_b)
        assert cnt == batch * n
        out_a, out_b = _like(ct_a, (batch, n)), _like(ct_a, (batch,))
        poa, _, _, _ = _buf(out_a)
        pob, _, _, _ = _buf(out_b)
        L.check(L.lib().fhe_fhew_bootstrap(self._h, q_ks, ks_log_b, ks_d, pka, pkb, pf, 0 if fcnt == n else n, addend, pa, pb, poa, pob,
                                           batch, mem, st), "fhe_fhew_bootstrap")
        return out_a, out_b

    def check(self, like=None, clear=True):
        """fhe_bootstrap_key_status: device-memory blind rotations / bootstraps are asynchronous; this waits for the stream `like`
        lives on (torch's current stream for a CUDA tensor) and raises if one of them met an even LWE coefficient."""
        st = _buf(like)[3] if like is not None else None
        L.check(L.lib().fhe_bootstrap_key_status(self._h, st, int(clear)), "fhe_bootstrap_key_status")

    def blind_rotate(self, lwe_a, lwe_b, f, want_schedule=False):
        """scheme/fhew/src/bootstrapping.rs:158-209 for a batch: lwe_a [batch][n_lwe], lwe_b [batch], f [n] or [batch][n].
        Device tensors: asynchronous, the data-dependent input check is reported by check()."""
        n, n_lwe = self.brk.n, self.brk.count
        pa, cnt, mem, st = _buf(lwe_a)
        pb, batch, _, _ = _buf(lwe_b)
        pf, fcnt, _, _ = _buf(f)
        assert cnt == batch * n_lwe and fcnt in (n, n * batch)
        out_a, out_b = _like(lwe_a, (batch, n)), _like(lwe_a, (batch, n))
        poa, _, _, _ = _buf(out_a)
        pob, _, _, _ = _buf(out_b)
        ops = nops = None
        po = pn = None
        if want_schedule:
            max_ops = n_lwe + n + 2
            ops = np.zeros((batch, max_ops), dtype=np.uint32)
            nops = np.zeros(batch, dtype=np.uint32)
            po, pn = ops.ctypes.data_as(C.POINTER(C.c_uint32)), nops.ctypes.data_as(C.POINTER(C.c_uint32))
        L.check(L.lib().fhe_blind_rotate(self._h, pa, pb, pf, 0 if fcnt == n else n, poa, pob, batch, mem, st, po, pn),
                "fhe_blind_rotate")
        if want_schedule:
            sched = [[("ak" if int(o) >> 31 else "ep", int(o) & 0x7FFFFFFF) for o in ops[i, :nops[i]]] for i in range(batch)]
            return out_a, out_b, sched
        return out_a, out_b


# ---- row a14: RNS rings / CKKS key switch ----------------------------------------------------------------


class RnsContext:
    """util/src/ring/rns.rs `Rns` for bases qs (L) and ps (K)."""

    def __init__(self, qs, ps, device=0):
        self.qs, self.ps, self.L, self.K = list(qs), list(ps), len(qs), len(ps)
        a, b = (C.c_uint64 * self.L)(*qs), (C.c_uint64 * self.K)(*ps)
        self._h = C.c_void_p()
        L.check(L.lib().fhe_rns_ctx_create(a, self.L, b, self.K, device, C.byref(self._h)), "fhe_rns_ctx_create")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and L is not None and getattr(L, "lib", None):  # (module globals are gone at interpreter shutdown)
            L.lib().fhe_rns_ctx_destroy(h)

    @property
    def handle(self):
        return self._h

    def ntt_(self, a, n, extended=False, inverse=False):
        """rns.rs:40-49 limb by limb, in place on [batch][L (+K)][n]."""
        p, cnt, mem, st = _buf(a)
        limbs = self.L + self.K if extended else self.L
        f = L.lib().fhe_rns_ntt_inv if inverse else L.lib().fhe_rns_ntt_fwd
        L.check(f(self._h, int(extended), p, n, cnt // (limbs * n), mem, st), "fhe_rns_ntt")
        return a

    def pointwise_mul_(self, a, b, n, extended=False):
        """rns.rs:148-158 (evaluation basis), in place on a."""
        pa, cnt, mem, st = _buf(a)
        pb, _, _, _ = _buf(b)
        limbs = self.L + self.K if extended else self.L
        L.check(L.lib().fhe_rns_pointwise_mul(self._h, int(extended), pa, pb, n, cnt // (limbs * n), mem, st), "fhe_rns_pointwise_mul")
        return a

    def extend_bases(self, limbs, n):
        """rns.rs:83-91: [batch][L][n] -> the K new limbs [batch][K][n]."""
        p, cnt, mem, st = _buf(limbs)
        batch = cnt // (self.L * n)
        out = _like(limbs, (batch, self.K, n))
        po, _, _, _ = _buf(out)
        L.check(L.lib().fhe_rns_extend_bases(self._h, p, po, n, batch, mem, st), "fhe_rns_extend_bases")
        return out

    def switch_bases(self, limbs, n, to_qs=False):
        """rns.rs:93-97: [batch][L][n] over qs -> [batch][K][n] over ps, or (to_qs) [batch][K][n] over ps -> [batch][L][n] over qs."""
        p, cnt, mem, st = _buf(limbs)
        la, lb = (self.K, self.L) if to_qs else (self.L, self.K)
        batch = cnt // (la * n)
        out = _like(limbs, (batch, lb, n))
        L.check(L.lib().fhe_rns_switch_bases(self._h, int(bool(to_qs)), p, _buf(out)[0], n, batch, mem, st), "fhe_rns_switch_bases")
        return out

    def rescale_k(self, limbs, n):
        """rns.rs:103-118: [batch][L+K][n] -> [batch][L][n]."""
        p, cnt, mem, st = _buf(limbs)
        batch = cnt // ((self.L + self.K) * n)
        out = _like(limbs, (batch, self.L, n))
        po, _, _, _ = _buf(out)
        L.check(L.lib().fhe_rns_rescale_k(self._h, p, po, n, batch, mem, st), "fhe_rns_rescale_k")
        return out


    def rescale(self, limbs, n):
        """rns.rs:99-101 `rescale()`: [batch][L][n] -> [batch][L-1][n]."""
        p, cnt, mem, st = _buf(limbs)
        batch = cnt // (self.L * n)
        out = _like(limbs, (batch, self.L - 1, n))
        po, _, _, _ = _buf(out)
        L.check(L.lib().fhe_rns_rescale(self._h, p, po, n, batch, mem, st), "fhe_rns_rescale")
        return out

    def sk_encrypt(self, sk, pt, n, batch, seed, stream_id, extended=False):
        """scheme/ckks/src/ckks.rs:215-225 -> (b, a) [batch][limbs][n]; sk [n] i64; pt [batch][limbs][n] or None."""
        ps, _, mem, st = _buf(sk)
        pp = _buf(pt)[0] if pt is not None else None
        limbs = self.L + self.K if extended else self.L
        b, a = _like(sk, (batch, limbs, n)), _like(sk, (batch, limbs, n))
        L.check(L.lib().fhe_ckks_sk_encrypt(self._h, int(extended), ps, pp, n, batch, _rng(seed), stream_id, _buf(b)[0], _buf(a)[0], mem, st),
                "fhe_ckks_sk_encrypt")
        return b, a

    def ksk_gen(self, sk, sk_prime, n, seed, stream_id):
        """scheme/ckks/src/ckks.rs:154-183 -> (ksk_b, ksk_a) [L+K][n]; sk_prime None = sk^2 (the relinearisation key)."""
        ps, _, mem, st = _buf(sk)
        pp = _buf(sk_prime)[0] if sk_prime is not None else None
        kb, ka = _like(sk, (self.L + self.K, n)), _like(sk, (self.L + self.K, n))
        L.check(L.lib().fhe_ckks_ksk_gen(self._h, ps, pp, n, _rng(seed), stream_id, _buf(kb)[0], _buf(ka)[0], mem, st), "fhe_ckks_ksk_gen")
        return kb, ka

    def add_(self, a, b, n, extended=False):
        """util/src/ring/rns.rs:254-270: a += b over [batch][limbs][n]."""
        pa, cnt, mem, st = _buf(a)
        limbs = self.L + self.K if extended else self.L
        L.check(L.lib().fhe_rns_add(self._h, int(extended), pa, _buf(b)[0], n, cnt // (limbs * n), mem, st), "fhe_rns_add")
        return a

    def sub_(self, a, b, n, extended=False):
        pa, cnt, mem, st = _buf(a)
        limbs = self.L + self.K if extended else self.L
        L.check(L.lib().fhe_rns_sub(self._h, int(extended), pa, _buf(b)[0], n, cnt // (limbs * n), mem, st), "fhe_rns_sub")
        return a

    def neg_(self, a, n, extended=False):
        pa, cnt, mem, st = _buf(a)
        limbs = self.L + self.K if extended else self.L
        L.check(L.lib().fhe_rns_neg(self._h, int(extended), pa, n, cnt // (limbs * n), mem, st), "fhe_rns_neg")
        return a

    def pk_encrypt(self, pk_b, pk_a, pt, n, batch, seed, stream_id):
        """scheme/ckks/src/ckks.rs:227-238 -> (b, a) [batch][L][n]; pk [L][n]; pt [batch][L][n] or None."""
        pb, _, mem, st = _buf(pk_b)
        pp = _buf(pt)[0] if pt is not None else None
        b, a = _like(pk_b, (batch, self.L, n)), _like(pk_b, (batch, self.L, n))
        L.check(L.lib().fhe_ckks_pk_encrypt(self._h, pb, _buf(pk_a)[0], pp, n, batch, _rng(seed), stream_id, _buf(b)[0], _buf(a)[0], mem, st),
                "fhe_ckks_pk_encrypt")
        return b, a

    def decrypt(self, sk, ct_b, ct_a, n):
        """scheme/ckks/src/ckks.rs:240-248: b + a sk -> [batch][L][n]."""
        pb, cnt, mem, st = _buf(ct_b)
        batch = cnt // (self.L * n)
        out = _like(ct_b, (batch, self.L, n))
        L.check(L.lib().fhe_ckks_decrypt(self._h, _buf(sk)[0], pb, _buf(ct_a)[0], n, batch, _buf(out)[0], mem, st), "fhe_ckks_decrypt")
        return out

    def mul_plain(self, pt, ct_b, ct_a, n):
        """scheme/ckks/src/ckks.rs:250-253 after `encode`: (pt b, pt a).rescale() -> (b, a) [batch][L-1][n]; pt [1 or batch][L][n]."""
        pb, cnt, mem, st = _buf(ct_b)
        batch = cnt // (self.L * n)
        pp, pcnt, _, _ = _buf(pt)
        ob, oa = _like(ct_b, (batch, self.L - 1, n)), _like(ct_b, (batch, self.L - 1, n))
        L.check(L.lib().fhe_ckks_mul_plain(self._h, pp, pcnt // (self.L * n), pb, _buf(ct_a)[0], _buf(ob)[0], _buf(oa)[0], n, batch, mem, st),
                "fhe_ckks_mul_plain")
        return ob, oa

    def automorphism(self, limbs, t, n):
        """ckks.rs:127-129 on [batch][L][n]."""
        p, cnt, mem, st = _buf(limbs)
        batch = cnt // (self.L * n)
        out = _like(limbs, (batch, self.L, n))
        po, _, _, _ = _buf(out)
        L.check(L.lib().fhe_rns_automorphism(self._h, t, p, po, n, batch, mem, st), "fhe_rns_automorphism")
        return out


class CkksKey:
    """A CKKS key-switching key (scheme/ckks/src/ckks.rs:86-88) prepared in the evaluation domain."""

    def __init__(self, rns: RnsContext, ksk_b, ksk_a, n):
        self.rns, self.n = rns, n
        pb, cnt, mem, _ = _buf(ksk_b)
        pa, cnta, mema, _ = _buf(ksk_a)
        assert cnt == cnta == (rns.L + rns.K) * n and mem == mema
        self._h = C.c_void_p()
        L.check(L.lib().fhe_ckks_ksk_prepare(rns.handle, pb, pa, n, mem, C.byref(self._h)), "fhe_ckks_ksk_prepare")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and L is not None and getattr(L, "lib", None):  # (module globals are gone at interpreter shutdown)
            L.lib().fhe_ckks_key_destroy(h)

    def key_switch_(self, ct_b, ct_a):
        """scheme/ckks/src/ckks.rs:284-293, in place on [batch][L][n] b/a."""
        pb, cnt, mem, st = _buf(ct_b)
        pa, _, _, _ = _buf(ct_a)
        batch = cnt // (self.rns.L * self.n)
        L.check(L.lib().fhe_ckks_key_switch(self.rns.handle, self._h, pb, pa, batch, mem, st), "fhe_ckks_key_switch")

    def rotate_(self, t, ct_b, ct_a):
        """scheme/ckks/src/ckks.rs:274-282 (rotate: t = 5^j mod 2n; conjugate: t = -1), in place."""
        pb, cnt, mem, st = _buf(ct_b)
        pa, _, _, _ = _buf(ct_a)
        batch = cnt // (self.rns.L * self.n)
        L.check(L.lib().fhe_ckks_rotate(self.rns.handle, self._h, t, pb, pa, batch, mem, st), "fhe_ckks_rotate")

    def mul(self, ct0_b, ct0_a, ct1_b, ct1_a):
        """scheme/ckks/src/ckks.rs:250-263 with this key as the relinearisation key -> (b, a) on L-1 limbs."""
        p0b, cnt, mem, st = _buf(ct0_b)
        p0a, p1b, p1a = _buf(ct0_a)[0], _buf(ct1_b)[0], _buf(ct1_a)[0]
        batch = cnt // (self.rns.L * self.n)
        ob, oa = _like(ct0_b, (batch, self.rns.L - 1, self.n)), _like(ct0_b, (batch, self.rns.L - 1, self.n))
        L.check(L.lib().fhe_ckks_mul(self.rns.handle, self._h, p0b, p0a, p1b, p1a, _buf(ob)[0], _buf(oa)[0], batch, mem, st), "fhe_ckks_mul")
        return ob, oa


class CkksShard:
    """One device's part of a limb-sharded key switch (include/fhe_ring.h fhe_ckks_shard_*): owns the
    q-limbs [q_lo, q_hi) and the p-limbs [p_lo, p_hi) of `key.rns`."""

    def __init__(self, key: CkksKey, q_lo, q_hi, p_lo, p_hi):
        self.key, self.rns, self.n = key, key.rns, key.n
        self.nq, self.np = q_hi - q_lo, p_hi - p_lo
        self._h = C.c_void_p()
        L.check(L.lib().fhe_ckks_shard_create(key.rns.handle, key._h, q_lo, q_hi, p_lo, p_hi, C.byref(self._h)), "fhe_ckks_shard_create")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and L is not None and getattr(L, "lib", None):
            L.lib().fhe_ckks_shard_destroy(h)

    def products(self, ct_a):
        """stage 1: ct_a [batch][L][n] (replicated) -> (prod_q [2][batch][nq][n], prod_p [2][batch][np][n])"""
        pa, cnt, mem, st = _buf(ct_a)
        batch = cnt // (self.rns.L * self.n)
        pq, pp = _like(ct_a, (2, batch, self.nq, self.n)), _like(ct_a, (2, batch, self.np, self.n))
        L.check(L.lib().fhe_ckks_shard_products(self._h, pa, _buf(pq)[0], _buf(pp)[0], batch, mem, st), "fhe_ckks_shard_products")
        return pq, pp

    def finish(self, prod_q, gathered_p, ct_b=None):
        """stage 2: prod_q, gathered_p [K / np][2][batch][np][n], ct_b [batch][nq][n] or None -> (out_b, out_a) [batch][nq][n]"""
        pq, cnt, mem, st = _buf(prod_q)
        batch = cnt // (2 * self.nq * self.n)
        ob, oa = _like(prod_q, (batch, self.nq, self.n)), _like(prod_q, (batch, self.nq, self.n))
        pb = _buf(ct_b)[0] if ct_b is not None else None
        L.check(L.lib().fhe_ckks_shard_finish(self._h, pq, _buf(gathered_p)[0], pb, _buf(ob)[0], _buf(oa)[0], batch, mem, st), "fhe_ckks_shard_finish")
        return ob, oa


# ---- row T: TFHE torus path ------------------------------------------------------------------------------


def torus_decompose(log_b, d, a, n):
    """util/src/misc/decompose.rs:114-135 on T64: [polys][n] -> [polys][d][n]."""
    p, cnt, mem, st = _buf(a)
    out = _like(a, (cnt // n, d, n))
    po, _, _, _ = _buf(out)
    L.check(L.lib().fhe_torus_decompose(log_b, d, p, n, cnt // n, po, mem, st), "fhe_torus_decompose")
    return out


class TorusContext:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        L.check(L.lib().fhe_torus_ctx_create(device, C.byref(self._h)), "fhe_torus_ctx_create")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and L is not None and getattr(L, "lib", None):  # (module globals are gone at interpreter shutdown)
            L.lib().fhe_torus_ctx_destroy(h)

    @property
    def handle(self):
        return self._h

    def mul_(self, a, b, log_bound_b, n):
        """a <- a * b in Z_{2^64}[X]/(X^n+1), exact (ring.rs:315-320)."""
        pa, cnt, mem, st = _buf(a)
        pb, _, _, _ = _buf(b)
        L.check(L.lib().fhe_torus_mul(self._h, pa, pb, log_bound_b, n, cnt // n, mem, st), "fhe_torus_mul")
        return a

    @staticmethod
    def mod_switch(v, big_n):
        p, cnt, mem, st = _buf(v)
        out = _like(v, tuple(v.shape))
        po, _, _, _ = _buf(out)
        L.check(L.lib().fhe_tfhe_mod_switch(p, po, cnt, big_n, mem, st), "fhe_tfhe_mod_switch")
        return out


class TggswKey:
    """Prepared TGGSW ciphertexts, k = 1 (scheme/tfhe/src/tggsw.rs:44-88)."""

    def __init__(self, t: TorusContext, log_b, d, rows_a, rows_b, n, fft64=False):
        """fft64: the f64 FFT product of the reference (util/src/ring/fft/c64.rs) instead of the exact one -- within the reference's
        error bound of the exact results, not bit-identical to anything."""
        self.t, self.log_b, self.d, self.n, self.fft64 = t, log_b, d, n, fft64
        pa, cnt, mem, _ = _buf(rows_a)
        pb, _, _, _ = _buf(rows_b)
        self.count = cnt // (2 * d * n)
        self._h = C.c_void_p()
        fn = L.lib().fhe_tggsw_prepare_fft64 if fft64 else L.lib().fhe_tggsw_prepare
        L.check(fn(t.handle, log_b, d, pa, pb, n, self.count, mem, C.byref(self._h)), "fhe_tggsw_prepare")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and L is not None and getattr(L, "lib", None):  # (module globals are gone at interpreter shutdown)
            L.lib().fhe_tggsw_key_destroy(h)

    def external_product_(self, index, ct_a, ct_b):
        pa, cnt, mem, st = _buf(ct_a)
        pb, _, _, _ = _buf(ct_b)
        L.check(L.lib().fhe_tggsw_external_product(self.t.handle, self._h, index, pa, pb, cnt // self.n, mem, st),
                "fhe_tggsw_external_product")

    def cmux(self, index, ct0_a, ct0_b, ct1_a, ct1_b):
        """scheme/tfhe/src/tggsw.rs:114-121: ct0 + external_product(key[index], ct1 - ct0) -> (a, b), [batch][n]."""
        p0a, cnt, mem, st = _buf(ct0_a)
        out_a, out_b = _like(ct0_a, (cnt // self.n, self.n)), _like(ct0_a, (cnt // self.n, self.n))
        L.check(L.lib().fhe_tggsw_cmux(self.t.handle, self._h, index, p0a, _buf(ct0_b)[0], _buf(ct1_a)[0], _buf(ct1_b)[0], _buf(out_a)[0], _buf(out_b)[0],
                                       cnt // self.n, mem, st), "fhe_tggsw_cmux")
        return out_a, out_b

    def blind_rotate(self, a_tilde, b_tilde, v):
        """scheme/tfhe/src/bootstrapping.rs:84-96."""
        pa, cnt, mem, st = _buf(a_tilde)
        pb, batch, _, _ = _buf(b_tilde)
        pv, _, _, _ = _buf(v)
        out_a, out_b = _like(a_tilde, (batch, self.n)), _like(a_tilde, (batch, self.n))
        poa, _, _, _ = _buf(out_a)
        pob, _, _, _ = _buf(out_b)
        L.check(L.lib().fhe_tfhe_blind_rotate(self.t.handle, self._h, pa, pb, pv, poa, pob, batch, mem, st), "fhe_tfhe_blind_rotate")
        return out_a, out_b

    def bootstrap(self, ks_log_b, ks_d, ksk_a, ksk_b, v, lwe_a, lwe_b):
        """scheme/tfhe/src/bootstrapping.rs:78-82 in one call: lwe_a [batch][n_lwe], lwe_b [batch] -> (a [batch][n_lwe], b [batch])."""
        pka, _, mem, st = _buf(ksk_a)
        pkb, _, _, _ = _buf(ksk_b)
        pv, _, _, _ = _buf(v)
        pa, _, _, _ = _buf(lwe_a)
        pb, batch, _, _ = _buf(lwe_b)
        out_a, out_b = _like(lwe_a, (batch, self.count)), _like(lwe_a, (batch,))
        poa, _, _, _ = _buf(out_a)
        pob, _, _, _ = _buf(out_b)
        L.check(L.lib().fhe_tfhe_bootstrap(self.t.handle, self._h, ks_log_b, ks_d, pka, pkb, pv, pa, pb, poa, pob, batch, mem, st),
                "fhe_tfhe_bootstrap")
        return out_a, out_b


class TggswKeyK:
    """Prepared TGGSW ciphertexts of any TGLWE rank k (scheme/tfhe/src/tggsw.rs:44-88; the reference's `TglweParam::n`).
    rows [count][(k + 1) d][k + 1][n]; ciphertexts are single buffers [batch][k + 1][n] (a_0 .. a_{k-1}, b)."""

    def __init__(self, t: TorusContext, k, log_b, d, rows, n):
        self.t, self.k, self.log_b, self.d, self.n = t, k, log_b, d, n
        pr, cnt, mem, _ = _buf(rows)
        self.count = cnt // ((k + 1) * d * (k + 1) * n)
        self._h = C.c_void_p()
        L.check(L.lib().fhe_tggswk_prepare(t.handle, k, log_b, d, pr, n, self.count, mem, C.byref(self._h)), "fhe_tggswk_prepare")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and L is not None and getattr(L, "lib", None):
            L.lib().fhe_tggswk_key_destroy(h)

    def _batch(self, cnt):
        return cnt // ((self.k + 1) * self.n)

    def external_product_(self, index, ct):
        """tggsw.rs:100-112, in place."""
        pc, cnt, mem, st = _buf(ct)
        L.check(L.lib().fhe_tggswk_external_product(self.t.handle, self._h, index, pc, self._batch(cnt), mem, st), "fhe_tggswk_external_product")

    def cmux(self, index, ct0, ct1):
        """tggsw.rs:114-121."""
        p0, cnt, mem, st = _buf(ct0)
        out = _like(ct0, (self._batch(cnt), self.k + 1, self.n))
        L.check(L.lib().fhe_tggswk_cmux(self.t.handle, self._h, index, p0, _buf(ct1)[0], _buf(out)[0], self._batch(cnt), mem, st), "fhe_tggswk_cmux")
        return out

    def blind_rotate(self, a_tilde, b_tilde, v):
        """bootstrapping.rs:84-96 -> [batch][k + 1][n]."""
        pa, _, mem, st = _buf(a_tilde)
        pb, batch, _, _ = _buf(b_tilde)
        out = _like(a_tilde, (batch, self.k + 1, self.n))
        L.check(L.lib().fhe_tfhek_blind_rotate(self.t.handle, self._h, pa, pb, _buf(v)[0], _buf(out)[0], batch, mem, st), "fhe_tfhek_blind_rotate")
        return out

    def bootstrap(self, ks_log_b, ks_d, ksk_a, ksk_b, v, lwe_a, lwe_b):
        """bootstrapping.rs:78-82 in one call: lwe_a [batch][n_lwe], lwe_b [batch] -> (a [batch][n_lwe], b [batch])."""
        pka, _, mem, st = _buf(ksk_a)
        pb, batch, _, _ = _buf(lwe_b)
        out_a, out_b = _like(lwe_a, (batch, self.count)), _like(lwe_a, (batch,))
        L.check(L.lib().fhe_tfhek_bootstrap(self.t.handle, self._h, ks_log_b, ks_d, pka, _buf(ksk_b)[0], _buf(v)[0], _buf(lwe_a)[0], pb, _buf(out_a)[0],
                                            _buf(out_b)[0], batch, mem, st), "fhe_tfhek_bootstrap")
        return out_a, out_b


def tglwek_rotate(ct, k, n, i):
    """scheme/tfhe/src/tglwe.rs:61-66 for [batch][k + 1][n] ciphertexts."""
    pc, cnt, mem, st = _buf(ct)
    batch = cnt // ((k + 1) * n)
    out = _like(ct, (batch, k + 1, n))
    L.check(L.lib().fhe_tglwek_rotate(pc, k, n, int(i), _buf(out)[0], batch, mem, st), "fhe_tglwek_rotate")
    return out


def tglwek_sample_extract(ct, k, n, index):
    """scheme/tfhe/src/tglwe.rs:115-127 -> (a [batch][k n], b [batch])."""
    pc, cnt, mem, st = _buf(ct)
    batch = cnt // ((k + 1) * n)
    out_a, out_b = _like(ct, (batch, k * n)), _like(ct, (batch,))
    L.check(L.lib().fhe_tglwek_sample_extract(pc, k, n, index, _buf(out_a)[0], _buf(out_b)[0], batch, mem, st), "fhe_tglwek_sample_extract")
    return out_a, out_b


def tglwek_sk_encrypt(t, k, sk, pt, n, rows, std_dev, seed, stream_id):
    """scheme/tfhe/src/tglwe.rs:91-103 at rank k: sk [k n] -> ct [rows][k + 1][n]"""
    ps, _, mem, st = _buf(sk)
    pp = _buf(pt)[0] if pt is not None else None
    ct = _like(sk, (rows, k + 1, n))
    L.check(L.lib().fhe_tglwek_sk_encrypt(t.handle, k, ps, pp, n, rows, std_dev, _rng(seed), stream_id, _buf(ct)[0], mem, st), "fhe_tglwek_sk_encrypt")
    return ct


def tggswk_encrypt(t, k, log_b, d, sk, pt, n, std_dev, seed, stream_id):
    """scheme/tfhe/src/tggsw.rs:73-88 at rank k for pt [count][n] -> rows [count][(k + 1) d][k + 1][n]"""
    ps, _, mem, st = _buf(sk)
    pp, cnt, _, _ = _buf(pt)
    count = cnt // n
    rows = _like(sk, (count, (k + 1) * d, k + 1, n))
    L.check(L.lib().fhe_tggswk_encrypt(t.handle, k, log_b, d, ps, pp, n, count, std_dev, _rng(seed), stream_id, _buf(rows)[0], mem, st), "fhe_tggswk_encrypt")
    return rows


def tglwe_sample_extract(ct_a, ct_b, n, index):
    pa, cnt, mem, st = _buf(ct_a)
    pb, _, _, _ = _buf(ct_b)
    batch = cnt // n
    out_a, out_b = _like(ct_a, (batch, n)), _like(ct_a, (batch,))
    poa, _, _, _ = _buf(out_a)
    pob, _, _, _ = _buf(out_b)
    L.check(L.lib().fhe_tglwe_sample_extract(pa, pb, n, index, poa, pob, batch, mem, st), "fhe_tglwe_sample_extract")
    return out_a, out_b


def tlwe_key_switch(log_b, d, ksk_a, ksk_b, ct_a, ct_b, n_in, n_out):
    pka, _, mem, st = _buf(ksk_a)
    pkb, _, _, _ = _buf(ksk_b)
    pa, cnt, _, _ = _buf(ct_a)
    pb, _, _, _ = _buf(ct_b)
    batch = cnt // n_in
    out_a, out_b = _like(ct_a, (batch, n_out)), _like(ct_a, (batch,))
    poa, _, _, _ = _buf(out_a)
    pob, _, _, _ = _buf(out_b)
    L.check(L.lib().fhe_tlwe_key_switch(log_b, d, pka, pkb, pa, pb, n_in, n_out, poa, pob, batch, mem, st), "fhe_tlwe_key_switch")
    return out_a, out_b


class Fhew:
    """Host mirror of `Fhew` (scheme/fhew/src/fhew.rs:16-70) over batches of LWE ciphertexts (a [batch][N], b [batch]) under the
    ring key: each gate is a linear combination, one `BootstrapKey.bootstrap` with the gate's table, and + Q/8."""

    TABLES = {"and": (0, 0, 0, 1), "nand": (1, 1, 1, 0), "or": (0, 1, 1, 1), "nor": (1, 0, 0, 0), "xor": (0, 1, 1, 1),
              "xnor": (1, 0, 0, 0), "majority": (0, 0, 0, 1)}

    def __init__(self, bk, q_ks, ks_log_b, ks_d, lwe_ksk_a, lwe_ksk_b):
        self.bk, self.q_ks, self.ks_log_b, self.ks_d, self.ksk_a, self.ksk_b = bk, q_ks, ks_log_b, ks_d, lwe_ksk_a, lwe_ksk_b
        self.big_q, self.n = bk.ctx.q, bk.brk.n
        self.big_q_by_8 = self._round_div(self.big_q, 8)
        self.big_q_by_4 = self._round_div(self.big_q, 4)
        self._f = {}

    @staticmethod
    def _round_div(q, k):
        # BootstrappingParam::big_q_by_8 (scheme/fhew/src/bootstrapping.rs:73-79): Zq::from_f64(q, q as f64 / k as f64)
        import math
        x = float(q) / float(k)
        fl = math.floor(x)
        return (int(fl) + (1 if x - fl >= 0.5 else 0)) % q  # f64::round: half away from zero (x > 0)

    def table_poly(self, table, like):
        """fhew.rs:32-37: f = table.flat_map(|out| repeat(+-Q/8).take(q_by_8)), q = 2N"""
        key = (tuple(table), _is_torch(like))
        if key not in self._f:
            vals = [(self.big_q - self.big_q_by_8) % self.big_q, self.big_q_by_8]
            f = np.array([vals[o] for o in table for _ in range(2 * self.n // 8)], dtype=np.uint64)
            if _is_torch(like):
                import torch
                f = torch.from_numpy(f.view(np.int64)).to(like.device)
            self._f[key] = f
        return self._f[key]

    def not_(self, ct):
        a, b = ct
        return lwe_lincomb(self.big_q, [-1], [a]), lwe_lincomb(self.big_q, [-1], [b], self.big_q_by_4)

    def _op(self, name, coefs, cts):
        a = lwe_lincomb(self.big_q, coefs, [c[0] for c in cts])
        b = lwe_lincomb(self.big_q, coefs, [c[1] for c in cts])
        return self.bk.bootstrap(self.q_ks, self.ks_log_b, self.ks_d, self.ksk_a, self.ksk_b, self.table_poly(self.TABLES[name], a), a, b,
                                 addend=self.big_q_by_8)

    def and_(self, c0, c1): return self._op("and", [1, 1], [c0, c1])          # noqa: E704
    def nand(self, c0, c1): return self._op("nand", [1, 1], [c0, c1])         # noqa: E704
    def or_(self, c0, c1): return self._op("or", [1, 1], [c0, c1])            # noqa: E704
    def nor(self, c0, c1): return self._op("nor", [1, 1], [c0, c1])           # noqa: E704
    def xor(self, c0, c1): return self._op("xor", [2, -2], [c0, c1])          # noqa: E704
    def xnor(self, c0, c1): return self._op("xnor", [2, -2], [c0, c1])        # noqa: E704
    def majority(self, c0, c1, c2): return self._op("majority", [1, 1, 1], [c0, c1, c2])  # noqa: E704


# ---- key material on the device ------------------------------------------------------------------


STREAM_AUTO = (1 << 64) - 1  # FHE_STREAM_AUTO: the generator numbers the call itself


class Rng:
    """`fhe_rng` (include/fhe_ring.h): the 256-bit ChaCha20 key every key-material producer draws under.  Rng(key=<32 bytes>) takes
    the caller's entropy, Rng() 32 bytes from the operating system; Rng(seed=<int>) is the 64-bit test / reproducibility form."""

    def __init__(self, key=None, seed=None):
        h = C.c_void_p()
        if seed is not None:
            L.check(L.lib().fhe_rng_create_from_seed(int(seed), C.byref(h)), "fhe_rng_create_from_seed")
        else:
            if key is not None and len(key) != 32:
                raise ValueError("an fhe_rng key is 32 bytes")
            L.check(L.lib().fhe_rng_create(bytes(key) if key is not None else None, C.byref(h)), "fhe_rng_create")
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            L.lib().fhe_rng_destroy(self._h)
            self._h = None


_SEED_RNGS = {}


def _rng(seed):
    """every wrapper below takes `seed`: an Rng, or an int -- the deterministic test form (one cached generator per value)"""
    if isinstance(seed, Rng):
        return seed._h
    r = _SEED_RNGS.get(int(seed))
    if r is None:
        r = _SEED_RNGS[int(seed)] = Rng(seed=int(seed))
    return r._h


def chacha20_block(key: bytes, nonce: int, counter: int) -> bytes:
    out = C.create_string_buffer(64)
    L.check(L.lib().fhe_chacha20_block(bytes(key), nonce, counter, out), "fhe_chacha20_block")
    return out.raw


def sample_uniform(q, seed, stream_id, like, shape):
    """util/src/zq.rs:91-93: uniform in [0, q); `like` picks host (numpy) or device (torch) output."""
    out = _like(like, tuple(shape))
    p, cnt, mem, st = _buf(out)
    L.check(L.lib().fhe_sample_uniform(q, _rng(seed), stream_id, p, cnt, mem, st), "fhe_sample_uniform")
    return out


def sample_torus(seed, stream_id, like, shape):
    out = _like(like, tuple(shape))
    p, cnt, mem, st = _buf(out)
    L.check(L.lib().fhe_sample_torus(_rng(seed), stream_id, p, cnt, mem, st), "fhe_sample_torus")
    return out


def sample_dg(q, std_dev, n_sigma, seed, stream_id, like, shape):
    """util/src/misc/distribution.rs:23-46 `dg(std_dev, n)` as Zq values (q = 0: two's-complement integers)."""
    out = _like(like, tuple(shape))
    p, cnt, mem, st = _buf(out)
    L.check(L.lib().fhe_sample_dg(q, float(std_dev), n_sigma, _rng(seed), stream_id, p, cnt, mem, st), "fhe_sample_dg")
    return out


def power_up(q, log_b, d, a, n):
    """util/src/misc/decompose.rs:35-40: [polys][n] -> [polys][d][n]."""
    p, cnt, mem, st = _buf(a)
    polys = cnt // n
    out = _like(a, (polys, d, n))
    po, _, _, _ = _buf(out)
    L.check(L.lib().fhe_power_up(q, log_b, d, p, n, polys, po, mem, st), "fhe_power_up")
    return out


def rlwe_sk_encrypt(ctx: NttContext, sk, pt, n, batch, seed, stream_id):
    """scheme/fhew/src/rlwe.rs:146-156; sk [n] as Zq values, pt [batch][n] or None (zeros) -> (a, b)."""
    ps, _, mem, st = _buf(sk)
    pp = _buf(pt)[0] if pt is not None else None
    a, b = _like(sk, (batch, n)), _like(sk, (batch, n))
    pa, _, _, _ = _buf(a)
    pb, _, _, _ = _buf(b)
    L.check(L.lib().fhe_rlwe_sk_encrypt(ctx.handle, ps, pp, n, batch, _rng(seed), stream_id, pa, pb, mem, st), "fhe_rlwe_sk_encrypt")
    return a, b


def lwe_share_encrypt(q, a, sk, pt, n, seed, stream_id):
    """scheme/fhew/src/lwe.rs:169-183 / 197-207: b[r] = <a[r], sk> + pt[r] + e[r] for given masks a [rows][n] -> b [rows]."""
    pa, cnt, mem, st = _buf(a)
    rows = cnt // n
    pp = _buf(pt)[0] if pt is not None else None
    b = _like(a, (rows,))
    L.check(L.lib().fhe_lwe_share_encrypt(q, pa, _buf(sk)[0], pp, n, rows, _rng(seed), stream_id, _buf(b)[0], mem, st), "fhe_lwe_share_encrypt")
    return b


def lwe_ksk_share_gen(q, log_b, d, crs, sk0, sk1, seed, stream_id):
    """scheme/fhew/src/lwe.rs:214-226: crs [n1 d][n0] -> b [n1 d]."""
    pc, _, mem, st = _buf(crs)
    p0, n0, _, _ = _buf(sk0)
    p1, n1, _, _ = _buf(sk1)
    b = _like(crs, (n1 * d,))
    L.check(L.lib().fhe_lwe_ksk_share_gen(q, log_b, d, pc, p0, n0, p1, n1, _rng(seed), stream_id, _buf(b)[0], mem, st), "fhe_lwe_ksk_share_gen")
    return b


def rgsw_pk_encrypt(ctx: NttContext, log_b, d, pk_a, pk_b, pt, n, seed, stream_id):
    """scheme/fhew/src/rgsw.rs:75-83; pt [count][n] -> (rows_a, rows_b) [count][2d][n]."""
    pa, _, mem, st = _buf(pk_a)
    pp, cnt, _, _ = _buf(pt)
    count = cnt // n
    ra, rb = _like(pk_a, (count, 2 * d, n)), _like(pk_a, (count, 2 * d, n))
    L.check(L.lib().fhe_rgsw_pk_encrypt(ctx.handle, log_b, d, pa, _buf(pk_b)[0], pp, n, count, _rng(seed), stream_id, _buf(ra)[0], _buf(rb)[0], mem, st),
            "fhe_rgsw_pk_encrypt")
    return ra, rb


def rlwe_share_encrypt(ctx: NttContext, a, sk, pt, n, rows, seed, stream_id):
    """scheme/fhew/src/rlwe.rs:237-249: b = a sk + e + pt for a given mask a ([rows][n], or [n] shared by every row) -> b [rows][n]."""
    pa, cnt, mem, st = _buf(a)
    pp = _buf(pt)[0] if pt is not None else None
    b = _like(sk, (rows, n))
    L.check(L.lib().fhe_rlwe_share_encrypt(ctx.handle, pa, cnt // n, _buf(sk)[0], pp, n, rows, _rng(seed), stream_id, _buf(b)[0], mem, st), "fhe_rlwe_share_encrypt")
    return b


def rlwe_pk_encrypt(ctx: NttContext, pk_a, pk_b, pt, n, batch, seed, stream_id):
    """scheme/fhew/src/rlwe.rs:158-170 -> (a, b) [batch][n]."""
    pa, _, mem, st = _buf(pk_a)
    pp = _buf(pt)[0] if pt is not None else None
    a, b = _like(pk_a, (batch, n)), _like(pk_a, (batch, n))
    L.check(L.lib().fhe_rlwe_pk_encrypt(ctx.handle, pa, _buf(pk_b)[0], pp, n, batch, _rng(seed), stream_id, _buf(a)[0], _buf(b)[0], mem, st), "fhe_rlwe_pk_encrypt")
    return a, b


def rlwe_decrypt(ctx: NttContext, sk, ct_a, ct_b, n):
    """scheme/fhew/src/rlwe.rs:172-175: b - a sk -> [batch][n]."""
    pa, cnt, mem, st = _buf(ct_a)
    out = _like(ct_a, (cnt // n, n))
    L.check(L.lib().fhe_rlwe_decrypt(ctx.handle, _buf(sk)[0], pa, _buf(ct_b)[0], n, cnt // n, _buf(out)[0], mem, st), "fhe_rlwe_decrypt")
    return out


def rgsw_encrypt(ctx: NttContext, log_b, d, sk, pt, n, seed, stream_id):
    """scheme/fhew/src/rgsw.rs:84-105; pt [count][n] -> (rows_a, rows_b) [count][2d][n]."""
    ps, _, mem, st = _buf(sk)
    pp, cnt, _, _ = _buf(pt)
    count = cnt // n
    ra, rb = _like(sk, (count, 2 * d, n)), _like(sk, (count, 2 * d, n))
    pa, _, _, _ = _buf(ra)
    pb, _, _, _ = _buf(rb)
    L.check(L.lib().fhe_rgsw_encrypt(ctx.handle, log_b, d, ps, pp, n, count, _rng(seed), stream_id, pa, pb, mem, st), "fhe_rgsw_encrypt")
    return ra, rb


def rlwe_ksk_gen(ctx: NttContext, log_b, d, sk0, sk1, t, n, seed, stream_id):
    """scheme/fhew/src/rlwe.rs:109-132: t = 0: key-switching key sk1 -> sk0; t != 0: automorphism key of sk0 for X -> X^t."""
    p0, _, mem, st = _buf(sk0)
    p1 = _buf(sk1)[0] if sk1 is not None else None
    ra, rb = _like(sk0, (d, n)), _like(sk0, (d, n))
    pa, _, _, _ = _buf(ra)
    pb, _, _, _ = _buf(rb)
    L.check(L.lib().fhe_rlwe_ksk_gen(ctx.handle, log_b, d, p0, p1, t, n, _rng(seed), stream_id, pa, pb, mem, st), "fhe_rlwe_ksk_gen")
    return ra, rb


def lwe_sk_encrypt(q, sk, pt, n, rows, seed, stream_id):
    """scheme/fhew/src/lwe.rs:128-139: sk [n], pt [rows] or None -> (a [rows][n], b [rows])."""
    ps, _, mem, st = _buf(sk)
    pp = _buf(pt)[0] if pt is not None else None
    a, b = _like(sk, (rows, n)), _like(sk, (rows,))
    pa, _, _, _ = _buf(a)
    pb, _, _, _ = _buf(b)
    L.check(L.lib().fhe_lwe_sk_encrypt(q, ps, pp, n, rows, _rng(seed), stream_id, pa, pb, mem, st), "fhe_lwe_sk_encrypt")
    return a, b


def lwe_ksk_gen(q, log_b, d, sk0, sk1, seed, stream_id):
    """scheme/fhew/src/lwe.rs:108-119: -> (ksk_a [n1 d][n0], ksk_b [n1 d])."""
    p0, n0, mem, st = _buf(sk0)
    p1, n1, _, _ = _buf(sk1)
    ka, kb = _like(sk0, (n1 * d, n0)), _like(sk0, (n1 * d,))
    pa, _, _, _ = _buf(ka)
    pb, _, _, _ = _buf(kb)
    L.check(L.lib().fhe_lwe_ksk_gen(q, log_b, d, p0, n0, p1, n1, _rng(seed), stream_id, pa, pb, mem, st), "fhe_lwe_ksk_gen")
    return ka, kb


def rq_sum(q, a, n):
    """util/src/ring.rs:328-341 `Rq: Sum`: [count][n] -> [n]."""
    p, cnt, mem, st = _buf(a)
    out = _like(a, (n,))
    po, _, _, _ = _buf(out)
    L.check(L.lib().fhe_rq_sum(q, p, n, cnt // n, po, mem, st), "fhe_rq_sum")
    return out


def tglwe_rotate(ct_a, ct_b, n, i):
    """scheme/tfhe/src/tglwe.rs:61-66: (a, b) X^i for [batch][n] ciphertexts."""
    pa, cnt, mem, st = _buf(ct_a)
    out_a, out_b = _like(ct_a, (cnt // n, n)), _like(ct_a, (cnt // n, n))
    L.check(L.lib().fhe_tglwe_rotate(pa, _buf(ct_b)[0], n, int(i), _buf(out_a)[0], _buf(out_b)[0], cnt // n, mem, st), "fhe_tglwe_rotate")
    return out_a, out_b


def sample_tdg(std_dev, seed, stream_id, like, count):
    """util/src/misc/distribution.rs:49-54"""
    out = _like(like, (count,))
    po, _, mem, st = _buf(out)
    L.check(L.lib().fhe_sample_tdg(std_dev, _rng(seed), stream_id, po, count, mem, st), "fhe_sample_tdg")
    return out


def sample_binary(seed, stream_id, like, count):
    """distribution.rs `binary()`"""
    out = _like(like, (count,))
    po, _, mem, st = _buf(out)
    L.check(L.lib().fhe_sample_binary(_rng(seed), stream_id, po, count, mem, st), "fhe_sample_binary")
    return out


def tlwe_sk_encrypt(sk, pt, n, rows, std_dev, seed, stream_id):
    """scheme/tfhe/src/tlwe.rs:122-132 -> (a [rows][n], b [rows])"""
    ps, _, mem, st = _buf(sk)
    pp = _buf(pt)[0] if pt is not None else None
    a, b = _like(sk, (rows, n)), _like(sk, (rows,))
    L.check(L.lib().fhe_tlwe_sk_encrypt(ps, pp, n, rows, std_dev, _rng(seed), stream_id, _buf(a)[0], _buf(b)[0], mem, st), "fhe_tlwe_sk_encrypt")
    return a, b


def tlwe_ksk_gen(log_b, d, sk0, sk1, std_dev, seed, stream_id):
    """scheme/tfhe/src/tlwe.rs:100-111 -> (ksk_a [n1 d][n0], ksk_b [n1 d])"""
    p0, n0, mem, st = _buf(sk0)
    p1, n1, _, _ = _buf(sk1)
    ka, kb = _like(sk0, (n1 * d, n0)), _like(sk0, (n1 * d,))
    L.check(L.lib().fhe_tlwe_ksk_gen(log_b, d, p0, n0, p1, n1, std_dev, _rng(seed), stream_id, _buf(ka)[0], _buf(kb)[0], mem, st), "fhe_tlwe_ksk_gen")
    return ka, kb


def tglwe_sk_encrypt(t, sk, pt, n, rows, std_dev, seed, stream_id):
    """scheme/tfhe/src/tglwe.rs:91-103 -> (a, b) [rows][n]"""
    ps, _, mem, st = _buf(sk)
    pp = _buf(pt)[0] if pt is not None else None
    a, b = _like(sk, (rows, n)), _like(sk, (rows, n))
    L.check(L.lib().fhe_tglwe_sk_encrypt(t.handle, ps, pp, n, rows, std_dev, _rng(seed), stream_id, _buf(a)[0], _buf(b)[0], mem, st), "fhe_tglwe_sk_encrypt")
    return a, b


def tggsw_encrypt(t, log_b, d, sk, pt, n, std_dev, seed, stream_id):
    """scheme/tfhe/src/tggsw.rs:73-88 for pt [count][n] -> (rows_a, rows_b) [count][2d][n]"""
    ps, _, mem, st = _buf(sk)
    pp, cnt, _, _ = _buf(pt)
    count = cnt // n
    ra, rb = _like(sk, (count, 2 * d, n)), _like(sk, (count, 2 * d, n))
    L.check(L.lib().fhe_tggsw_encrypt(t.handle, log_b, d, ps, pp, n, count, std_dev, _rng(seed), stream_id, _buf(ra)[0], _buf(rb)[0], mem, st), "fhe_tggsw_encrypt")
    return ra, rb


def sample_zo(rho, seed, stream_id, like, count):
    """util/src/misc/distribution.rs:10-21 as two's-complement i64"""
    out = _like(like, (count,))
    po, _, mem, st = _buf(out)
    L.check(L.lib().fhe_sample_zo(rho, _rng(seed), stream_id, po, count, mem, st), "fhe_sample_zo")
    return out
