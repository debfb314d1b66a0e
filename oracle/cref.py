"""ctypes binding of oracle/_build/libref_ring.so (the C restatement).  TEST INFRASTRUCTURE ONLY
-- see the header of oracle/ref_ring.c.  numpy uint64 arrays in, numpy uint64 arrays out."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libref_ring.so")
_lib = None

u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ref_ring.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.ref_generator.restype = C.c_uint64
        _lib.ref_generator.argtypes = [C.c_uint64]
        _lib.ref_mod_switch.restype = C.c_uint64
        _lib.ref_mod_switch.argtypes = [C.c_uint64] * 3
        _lib.ref_mod_switch_odd.restype = C.c_uint64
        _lib.ref_mod_switch_odd.argtypes = [C.c_uint64] * 3
        _lib.ref_blind_rotate_schedule.restype = C.c_size_t
    return _lib


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


def is_prime(q):
    return bool(lib().ref_is_prime(C.c_uint64(q)))


def two_adic_primes(bits, log_n, count):
    out = np.zeros(count, dtype=np.uint64)
    k = lib().ref_two_adic_primes(bits, log_n, count, _p(out))
    return [int(v) for v in out[:k]]


def twiddle_info(q, count):
    s, g, w = C.c_int(), C.c_uint64(), C.c_uint64()
    tw, twi = np.zeros(count, dtype=np.uint64), np.zeros(count, dtype=np.uint64)
    rc = lib().ref_twiddle_info(C.c_uint64(q), C.byref(s), C.byref(g), C.byref(w), _p(tw), _p(twi), C.c_size_t(count))
    if rc:
        raise ValueError("q not prime")
    return s.value, g.value, w.value, tw, twi


def ntt_fwd(q, a, n, threads=1):
    a = _arr(a).copy()
    batch = a.size // n
    rc = lib().ref_ntt_fwd(C.c_uint64(q), _p(a), C.c_size_t(n), C.c_size_t(batch), threads)
    if rc:
        raise ValueError("ref_ntt_fwd: invalid (q, n)")
    return a


def ntt_inv(q, a, n, threads=1):
    a = _arr(a).copy()
    batch = a.size // n
    rc = lib().ref_ntt_inv(C.c_uint64(q), _p(a), C.c_size_t(n), C.c_size_t(batch), threads)
    if rc:
        raise ValueError("ref_ntt_inv: invalid (q, n)")
    return a


def ntt_fwd_inplace(q, a, n, threads=1):
    """In place on a caller-owned uint64 array (used by the timed cpu_baseline leg)."""
    return lib().ref_ntt_fwd(C.c_uint64(q), _p(a), C.c_size_t(n), C.c_size_t(a.size // n), threads)


def ntt_inv_inplace(q, a, n, threads=1):
    return lib().ref_ntt_inv(C.c_uint64(q), _p(a), C.c_size_t(n), C.c_size_t(a.size // n), threads)


def ntt_mul(q, a, b, n):
    a, b = _arr(a).copy(), _arr(b)
    rc = lib().ref_ntt_mul(C.c_uint64(q), _p(a), _p(b), C.c_size_t(n), C.c_size_t(a.size // n))
    if rc:
        raise ValueError("ref_ntt_mul: invalid (q, n)")
    return a


def pointwise_mul(q, a, b):
    a, b = _arr(a).copy(), _arr(b)
    lib().ref_pointwise_mul(C.c_uint64(q), _p(a), _p(b), C.c_size_t(a.size))
    return a


def schoolbook_mul(q, a, b):
    a, b = _arr(a), _arr(b)
    c = np.zeros_like(a)
    lib().ref_schoolbook_mul(C.c_uint64(q), _p(a), _p(b), _p(c), C.c_size_t(a.size))
    return c


def automorphism(q, t, a):
    a = _arr(a)
    out = np.zeros_like(a)
    lib().ref_automorphism(C.c_uint64(q), C.c_int64(t), _p(a), _p(out), C.c_size_t(a.size))
    return out


def monomial_mul(q, k, a):
    a = _arr(a)
    out = np.zeros_like(a)
    lib().ref_monomial_mul(C.c_uint64(q), C.c_int64(k), _p(a), _p(out), C.c_size_t(a.size))
    return out


def decompose(q, log_b, d, a):
    """-> [d][n] digit-major."""
    a = _arr(a)
    out = np.zeros((d, a.size), dtype=np.uint64)
    lib().ref_decompose(C.c_uint64(q), log_b, d, _p(a), C.c_size_t(a.size), _p(out))
    return out


def rlwe_key_switch(q, log_b, d, ksk_a, ksk_b, ct_a, ct_b):
    ksk_a, ksk_b, ct_a, ct_b = _arr(ksk_a), _arr(ksk_b), _arr(ct_a).copy(), _arr(ct_b).copy()
    rc = lib().ref_rlwe_key_switch(C.c_uint64(q), log_b, d, _p(ksk_a), _p(ksk_b), _p(ct_a), _p(ct_b), C.c_size_t(ct_a.size))
    assert rc == 0
    return ct_a, ct_b


def rlwe_automorphism(q, log_b, d, t, ak_a, ak_b, ct_a, ct_b):
    ak_a, ak_b, ct_a, ct_b = _arr(ak_a), _arr(ak_b), _arr(ct_a).copy(), _arr(ct_b).copy()
    rc = lib().ref_rlwe_automorphism(C.c_uint64(q), log_b, d, C.c_int64(t), _p(ak_a), _p(ak_b), _p(ct_a), _p(ct_b),
                                     C.c_size_t(ct_a.size))
    assert rc == 0
    return ct_a, ct_b


def external_product(q, log_b, d, rgsw_a, rgsw_b, ct_a, ct_b):
    rgsw_a, rgsw_b, ct_a, ct_b = _arr(rgsw_a), _arr(rgsw_b), _arr(ct_a).copy(), _arr(ct_b).copy()
    rc = lib().ref_external_product(C.c_uint64(q), log_b, d, _p(rgsw_a), _p(rgsw_b), _p(ct_a), _p(ct_b),
                                    C.c_size_t(ct_a.size))
    assert rc == 0
    return ct_a, ct_b


def blind_rotate_schedule(n, w, a):
    a = _arr(a)
    ops = np.zeros(a.size + n + 4, dtype=np.uint64)
    k = lib().ref_blind_rotate_schedule(C.c_size_t(n), w, _p(a), C.c_size_t(a.size), _p(ops))
    return [("ak" if int(o) >> 32 else "ep", int(o) & 0xFFFFFFFF) for o in ops[:k]]


def blind_rotate(q, n, w, log_b, d, ks_log_b, ks_d, brk, ak, ak_t, f, lwe_a, lwe_b):
    """brk: [n_lwe][2][2d][n]; ak: [w+1][2][ks_d][n]."""
    brk, ak, f, lwe_a = _arr(brk), _arr(ak), _arr(f), _arr(lwe_a)
    akt = np.ascontiguousarray(np.asarray(ak_t, dtype=np.int64))
    oa, ob = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
    rc = lib().ref_blind_rotate(C.c_uint64(q), C.c_size_t(n), w, log_b, d, ks_log_b, ks_d, _p(brk), _p(ak),
                                akt.ctypes.data_as(i64p), _p(f), _p(lwe_a), C.c_uint64(lwe_b),
                                C.c_size_t(lwe_a.size), _p(oa), _p(ob))
    assert rc == 0
    return oa, ob


def sample_extract(q, ct_a, ct_b, i):
    ct_a, ct_b = _arr(ct_a), _arr(ct_b)
    la, lb = np.zeros_like(ct_a), C.c_uint64()
    lib().ref_sample_extract(C.c_uint64(q), _p(ct_a), _p(ct_b), C.c_size_t(ct_a.size), C.c_size_t(i), _p(la), C.byref(lb))
    return la, lb.value


def mod_switch(q, v, q_prime):
    return lib().ref_mod_switch(q, v, q_prime)


def mod_switch_odd(q, v, q_prime):
    return lib().ref_mod_switch_odd(q, v, q_prime)


def lwe_key_switch(q, log_b, d, ksk_a, ksk_b, ct_a, ct_b):
    ksk_a, ksk_b, ct_a = _arr(ksk_a), _arr(ksk_b), _arr(ct_a)
    n_in, n_out = ct_a.size, ksk_a.shape[-1]
    oa, ob = np.zeros(n_out, dtype=np.uint64), C.c_uint64()
    lib().ref_lwe_key_switch(C.c_uint64(q), log_b, d, _p(ksk_a), _p(ksk_b), _p(ct_a), C.c_uint64(ct_b),
                             C.c_size_t(n_in), C.c_size_t(n_out), _p(oa), C.byref(ob))
    return oa, ob.value


def rns_extend_bases(qs, ps, limbs):
    qs_, ps_, limbs = _arr(qs), _arr(ps), _arr(limbs)
    n = limbs.shape[-1]
    out = np.zeros((len(ps), n), dtype=np.uint64)
    rc = lib().ref_rns_extend_bases(_p(qs_), len(qs), _p(ps_), len(ps), _p(limbs), _p(out), C.c_size_t(n))
    assert rc == 0
    return out


def rns_rescale_k(qps, k, limbs):
    qps_, limbs = _arr(qps), _arr(limbs).copy()
    n = limbs.shape[-1]
    rc = lib().ref_rns_rescale_k(_p(qps_), len(qps), k, _p(limbs), C.c_size_t(n))
    assert rc == 0
    return limbs[: len(qps) - k]


def ckks_key_switch(qs, ps, ksk_b, ksk_a, ct_b, ct_a):
    qs_, ps_ = _arr(qs), _arr(ps)
    ksk_b, ksk_a, ct_b, ct_a = _arr(ksk_b), _arr(ksk_a), _arr(ct_b).copy(), _arr(ct_a).copy()
    n = ct_a.shape[-1]
    rc = lib().ref_ckks_key_switch(_p(qs_), len(qs), _p(ps_), len(ps), _p(ksk_b), _p(ksk_a), _p(ct_b), _p(ct_a),
                                   C.c_size_t(n))
    assert rc == 0
    return ct_b, ct_a


def rns_mul(qs, a, b):
    """util/src/ring/rns.rs:148-158 on coefficient-domain limbs: [L][n] x [L][n]"""
    a, b = _arr(a).copy(), _arr(b)
    for i, q in enumerate(qs):
        a[i] = ntt_mul(q, a[i], b[i], a.shape[-1])
    return a


def rns_automorphism(qs, limbs, t):
    limbs = _arr(limbs)
    return np.stack([automorphism(q, t, limbs[i]) for i, q in enumerate(qs)])


def _rns_add(qs, a, b):
    qv = np.array(qs, dtype=np.uint64)[:, None]
    s = a + b  # < 2^63 for the 62-bit moduli of this code base
    return np.where(s >= qv, s - qv, s)


def ckks_rotate(qs, ps, key_b, key_a, t, ct_b, ct_a):
    """scheme/ckks/src/ckks.rs:274-282"""
    return ckks_key_switch(qs, ps, key_b, key_a, rns_automorphism(qs, ct_b, t), rns_automorphism(qs, ct_a, t))


def ckks_mul(qs, ps, rlk_b, rlk_a, ct0_b, ct0_a, ct1_b, ct1_a):
    """scheme/ckks/src/ckks.rs:250-272 composed from the C primitives"""
    ct0_b, ct0_a, ct1_b, ct1_a = _arr(ct0_b), _arr(ct0_a), _arr(ct1_b), _arr(ct1_a)
    d0 = rns_mul(qs, ct0_b, ct1_b)
    d1 = _rns_add(qs, rns_mul(qs, ct0_b, ct1_a), rns_mul(qs, ct0_a, ct1_b))
    d2 = rns_mul(qs, ct0_a, ct1_a)
    kb, ka = ckks_key_switch(qs, ps, rlk_b, rlk_a, np.zeros_like(d2), d2)
    return rns_rescale_k(qs, 1, _rns_add(qs, d0, kb)), rns_rescale_k(qs, 1, _rns_add(qs, d1, ka))


def rns_from_i64(qs, v):
    """util/src/ring/rns.rs `RnsRq::from_i64`: a two's-complement i64 vector over every modulus -> [L][n]"""
    v = np.asarray(v).view(np.int64) if np.asarray(v).dtype == np.uint64 else np.asarray(v, dtype=np.int64)
    return np.stack([np.mod(v, np.int64(q)).astype(np.uint64) if q < (1 << 63) else None for q in qs])


def ckks_decrypt(qs, sk, ct_b, ct_a):
    """scheme/ckks/src/ckks.rs:240-248: b + a * sk"""
    return _rns_add(qs, _arr(ct_b), rns_mul(qs, ct_a, rns_from_i64(qs, sk)))


def ckks_mul_plain(qs, pt, ct_b, ct_a):
    """scheme/ckks/src/ckks.rs:250-253 after `encode`: (pt * b, pt * a).rescale()"""
    return rns_rescale_k(qs, 1, rns_mul(qs, ct_b, pt)), rns_rescale_k(qs, 1, rns_mul(qs, ct_a, pt))


def num_threads():
    return lib().ref_num_threads()


# ---- TFHE torus path (row T): exact product = the checker; fft64 = the reference's floating-point algorithm (CPU baseline) ----

def torus_mul_exact(a, b):
    a, b = _arr(a), _arr(b)
    c = np.zeros_like(a)
    lib().ref_torus_mul_exact(_p(a), _p(b), _p(c), C.c_size_t(a.size))
    return c


def torus_mul_fft64(a, b):
    a, b = _arr(a).copy(), _arr(b)
    lib().ref_torus_mul_fft64(_p(a), _p(b), C.c_size_t(a.size))
    return a


def torus_monomial_mul(a, k):
    a = _arr(a)
    out = np.zeros_like(a)
    lib().ref_torus_monomial_mul(C.c_int64(k), _p(a), _p(out), C.c_size_t(a.size))
    return out


def torus_decompose(log_b, d, a):
    a = _arr(a)
    out = np.zeros((d, a.size), dtype=np.uint64)
    lib().ref_torus_decompose(log_b, d, _p(a), C.c_size_t(a.size), _p(out))
    return out


def tggsw_external_product(log_b, d, rows_a, rows_b, ct_a, ct_b, fft=False):
    """rows_*: [2d][n] -> (a', b')"""
    rows_a, rows_b, ct_a, ct_b = _arr(rows_a), _arr(rows_b), _arr(ct_a).copy(), _arr(ct_b).copy()
    rc = lib().ref_tggsw_external_product(log_b, d, _p(rows_a), _p(rows_b), _p(ct_a), _p(ct_b), C.c_size_t(ct_a.size), int(fft))
    assert rc == 0
    return ct_a, ct_b


def tfhe_mod_switch(values, big_n):
    v = _arr(values)
    out = np.zeros_like(v)
    lib().ref_tfhe_mod_switch(_p(v), _p(out), C.c_size_t(v.size), C.c_size_t(big_n))
    return out


def tfhe_blind_rotate(log_b, d, brk_a, brk_b, v, a_tilde, b_tilde, threads=1, fft=False):
    """brk_*: [n_lwe][2d][n]; a_tilde: [batch][n_lwe]; b_tilde: [batch] -> (acc_a, acc_b) each [batch][n]"""
    brk_a, brk_b, v, a_tilde, b_tilde = _arr(brk_a), _arr(brk_b), _arr(v), _arr(a_tilde), _arr(b_tilde)
    n, batch, n_lwe = v.size, b_tilde.size, brk_a.shape[0]
    oa, ob = np.zeros((batch, n), dtype=np.uint64), np.zeros((batch, n), dtype=np.uint64)
    rc = lib().ref_tfhe_blind_rotate(log_b, d, _p(brk_a), _p(brk_b), C.c_size_t(n_lwe), _p(v), _p(a_tilde), _p(b_tilde), _p(oa), _p(ob),
                                     C.c_size_t(n), C.c_size_t(batch), threads, int(fft))
    assert rc == 0
    return oa, ob


def tglwe_sample_extract(ct_a, ct_b, i):
    ct_a, ct_b = _arr(ct_a), _arr(ct_b)
    la, lb = np.zeros_like(ct_a), C.c_uint64()
    lib().ref_tglwe_sample_extract(_p(ct_a), _p(ct_b), C.c_size_t(ct_a.size), C.c_size_t(i), _p(la), C.byref(lb))
    return la, lb.value


def tlwe_key_switch(log_b, d, ksk_a, ksk_b, ct_a, ct_b):
    ksk_a, ksk_b, ct_a = _arr(ksk_a), _arr(ksk_b), _arr(ct_a)
    n_in, n_out = ct_a.size, ksk_a.shape[-1]
    oa, ob = np.zeros(n_out, dtype=np.uint64), C.c_uint64()
    lib().ref_tlwe_key_switch(log_b, d, _p(ksk_a), _p(ksk_b), _p(ct_a), C.c_uint64(ct_b), C.c_size_t(n_in), C.c_size_t(n_out), _p(oa),
                              C.byref(ob))
    return oa, ob.value


def tfhe_bootstrap(log_b, d, ks_log_b, ks_d, brk_a, brk_b, ksk_a, ksk_b, v, lwe_a, lwe_b, threads=1, fft=False):
    """tfhe/bootstrapping.rs:78-82 on a batch: lwe_a [batch][n_lwe], lwe_b [batch] -> (a [batch][n_lwe], b [batch])"""
    brk_a, brk_b, ksk_a, ksk_b, v, lwe_a, lwe_b = map(_arr, (brk_a, brk_b, ksk_a, ksk_b, v, lwe_a, lwe_b))
    n, batch, n_lwe = v.size, lwe_b.size, brk_a.shape[0]
    oa, ob = np.zeros((batch, n_lwe), dtype=np.uint64), np.zeros(batch, dtype=np.uint64)
    rc = lib().ref_tfhe_bootstrap(log_b, d, ks_log_b, ks_d, _p(brk_a), _p(brk_b), _p(ksk_a), _p(ksk_b), C.c_size_t(n_lwe), _p(v), _p(lwe_a),
                                  _p(lwe_b), _p(oa), _p(ob), C.c_size_t(n), C.c_size_t(batch), threads, int(fft))
    assert rc == 0
    return oa, ob


# ---- row T at any TGLWE rank k: ciphertexts are [k + 1][n] buffers, TGGSW rows [(k + 1) d][k + 1][n] ----

def tggswk_external_product(k, log_b, d, rows, ct):
    rows, ct = _arr(rows), _arr(ct).copy()
    n = ct.shape[-1]
    assert ct.shape == (k + 1, n) and rows.shape == ((k + 1) * d, k + 1, n)
    assert lib().ref_tggswk_external_product(k, log_b, d, _p(rows), _p(ct), C.c_size_t(n)) == 0
    return ct


def tggswk_cmux(k, log_b, d, rows, ct0, ct1):
    rows, ct0, ct1 = _arr(rows), _arr(ct0), _arr(ct1)
    out = np.zeros_like(ct0)
    assert lib().ref_tggswk_cmux(k, log_b, d, _p(rows), _p(ct0), _p(ct1), _p(out), C.c_size_t(ct0.shape[-1])) == 0
    return out


def tfhek_blind_rotate(k, log_b, d, brk, v, a_tilde, b_tilde, threads=1):
    """brk [n_lwe][(k+1)d][k+1][n]; a_tilde [batch][n_lwe]; b_tilde [batch] -> [batch][k+1][n]"""
    brk, v, a_tilde, b_tilde = _arr(brk), _arr(v), _arr(a_tilde), _arr(b_tilde)
    n, batch, n_lwe = v.size, b_tilde.size, brk.shape[0]
    out = np.zeros((batch, k + 1, n), dtype=np.uint64)
    rc = lib().ref_tfhek_blind_rotate(k, log_b, d, _p(brk), C.c_size_t(n_lwe), _p(v), _p(a_tilde), _p(b_tilde), _p(out), C.c_size_t(n),
                                      C.c_size_t(batch), threads)
    assert rc == 0
    return out


def tglwek_sample_extract(k, ct, i):
    ct = _arr(ct)
    n = ct.shape[-1]
    la, lb = np.zeros(k * n, dtype=np.uint64), C.c_uint64()
    lib().ref_tglwek_sample_extract(k, _p(ct), C.c_size_t(n), C.c_size_t(i), _p(la), C.byref(lb))
    return la, lb.value


def tfhek_bootstrap(k, log_b, d, ks_log_b, ks_d, brk, ksk_a, ksk_b, v, lwe_a, lwe_b, threads=1):
    brk, ksk_a, ksk_b, v, lwe_a, lwe_b = map(_arr, (brk, ksk_a, ksk_b, v, lwe_a, lwe_b))
    n, batch, n_lwe = v.size, lwe_b.size, brk.shape[0]
    oa, ob = np.zeros((batch, n_lwe), dtype=np.uint64), np.zeros(batch, dtype=np.uint64)
    rc = lib().ref_tfhek_bootstrap(k, log_b, d, ks_log_b, ks_d, _p(brk), _p(ksk_a), _p(ksk_b), C.c_size_t(n_lwe), _p(v), _p(lwe_a), _p(lwe_b),
                                   _p(oa), _p(ob), C.c_size_t(n), C.c_size_t(batch), threads)
    assert rc == 0
    return oa, ob
