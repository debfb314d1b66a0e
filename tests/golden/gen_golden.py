#!/usr/bin/env python3
"""Regenerates tests/golden/*.json and ntt_2p14.npz from the Python big-int restatement
(oracle/pyref.py), fixed seeds.  The reference holds no fixed vectors (all its tests draw from
thread_rng()) and cannot be built here, so these are restatement-derived fixtures: they pin the
oracle and the HIP path against regressions and against each other, not against a Rust binary.

Run from the repo root:  python tests/golden/gen_golden.py
"""
import json
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyref as P  # noqa: E402


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")


def moduli():
    out = []
    for label, bits, log_n in [("cfg1", 30, 11), ("cfg2", 60, 15), ("cfg3", 54, 11), ("t28", 28, 10), ("t54", 54, 10),
                               ("t55", 55, 12), ("t45", 45, 10)]:
        q = next(P.two_adic_primes(bits, log_n))
        tw, twi = P.twiddle(q)
        s = ((q - 1) & -(q - 1)).bit_length() - 1
        out.append({"label": label, "bits": bits, "log_n": log_n, "q": q, "s": s, "g": P.generator(q),
                    "omega": P.two_adic_generator(q, s), "tw": tw[:16], "twi": twi[:16]})
    qs, ps = P.ckks_primes(15, 60, 8)
    return {"moduli": out, "cfg4_qs": qs, "cfg4_ps": ps}


def ntt_vectors():
    rng = P.SplitMix64(1)
    out = []
    q1 = next(P.two_adic_primes(30, 11))
    out.append({"q": q1, "n": 8, "a": list(range(8)), "ntt": P.nega_cyclic_ntt(q1, list(range(8)))})
    out.append({"q": q1, "n": 8, "a": [0, 1, 0, 0, 0, 0, 0, 0], "ntt": P.nega_cyclic_ntt(q1, [0, 1, 0, 0, 0, 0, 0, 0])})
    a = rng.uniform(q1, 1024)  # cfg1: N = 1024, 30-bit prime, seed 1
    out.append({"q": q1, "n": 1024, "a": a, "ntt": P.nega_cyclic_ntt(q1, a)})
    q3 = next(P.two_adic_primes(54, 11))
    a = rng.uniform(q3, 1024)
    out.append({"q": q3, "n": 1024, "a": a, "ntt": P.nega_cyclic_ntt(q3, a)})
    for log_n in range(1, 8):  # every small size once, 45-bit primes as in the reference tests
        q = next(P.two_adic_primes(45, log_n + 1))
        a = rng.uniform(q, 1 << log_n)
        out.append({"q": q, "n": 1 << log_n, "a": a, "ntt": P.nega_cyclic_ntt(q, a)})
    # products (mathematically unique): a * b == schoolbook
    prods = []
    for log_n in (3, 6):
        q = next(P.two_adic_primes(45, log_n + 1))
        a, b = rng.uniform(q, 1 << log_n), rng.uniform(q, 1 << log_n)
        c = P.nega_cyclic_ntt_mul(q, a, b)
        assert c == P.nega_cyclic_schoolbook_mul(q, a, b)
        prods.append({"q": q, "n": 1 << log_n, "a": a, "b": b, "c": c})
    # cfg2: one N = 2^14 vector, 60-bit prime, seed 2 (binary, 2 x 128 KiB)
    q2 = next(P.two_adic_primes(60, 15))
    a = P.SplitMix64(2).uniform(q2, 1 << 14)
    np.savez_compressed(os.path.join(HERE, "ntt_2p14.npz"), q=np.uint64(q2), a=np.array(a, dtype=np.uint64),
                        ntt=np.array(P.nega_cyclic_ntt(q2, a), dtype=np.uint64))
    print("wrote ntt_2p14.npz")
    return {"ntt": out, "mul": prods}


def decompose_vectors():
    rnd = random.Random(3)
    out = []
    for bits, log_n, log_b, d in [(54, 11, 6, 9), (28, 10, 7, 4), (54, 10, 6, 9), (45, 10, 5, 9), (55, 12, 11, 5)]:
        q = next(P.two_adic_primes(bits, log_n))
        dec = P.Base2Decomposor(q, log_b, d)
        v = [0, 1, q - 1, q >> 1, (q >> 1) - 1, (q >> 1) + 1, (1 << log_b) - 1, 1 << (log_b - 1)]
        v += [rnd.randrange(q) for _ in range(24)]
        out.append({"q": q, "log_b": log_b, "d": d, "rounding_bits": dec.rounding_bits, "in": v, "digits": dec.decompose(v)})
    q = 1 << 16  # LWE key-switch modulus of the FHEW parameter sets (not prime)
    dec = P.Base2Decomposor(q, 4, 4)
    v = [0, 1, q - 1, q >> 1, (q >> 1) - 1, 12345, 54321, 40000]
    out.append({"q": q, "log_b": 4, "d": 4, "rounding_bits": dec.rounding_bits, "in": v, "digits": dec.decompose(v)})
    return out


def automorphism_vectors():
    rng = P.SplitMix64(4)
    q = next(P.two_adic_primes(54, 11))
    a = rng.uniform(q, 16)
    out = [{"q": q, "t": t, "in": a, "out": P.automorphism(q, a, t)} for t in (5, -5, 25, 1, -1, 31)]
    mono = [{"q": q, "k": k, "in": a, "out": P.monomial_mul(q, a, k)} for k in (0, 1, -1, 15, 16, 17, 31, -16, 53)]
    return {"automorphism": out, "monomial": mono}


def rlwe_vectors():
    """uniform-random key rows: noise/validity is irrelevant for bit parity."""
    rng = P.SplitMix64(5)
    n, log_b, d = 128, 6, 3
    q = next(P.two_adic_primes(54, 11))
    dec = P.Base2Decomposor(q, log_b, d)
    ra = [rng.uniform(q, n) for _ in range(2 * d)]
    rb = [rng.uniform(q, n) for _ in range(2 * d)]
    ca, cb = rng.uniform(q, n), rng.uniform(q, n)
    oa, ob = P.rgsw_external_product(q, dec, ra, rb, ca, cb)
    ka = [rng.uniform(q, n) for _ in range(d)]
    kb = [rng.uniform(q, n) for _ in range(d)]
    sa, sb = P.rlwe_key_switch(q, dec, ka, kb, ca, cb)
    ta, tb = P.rlwe_automorphism(q, dec, -5, ka, kb, ca, cb)
    return {"q": q, "n": n, "log_b": log_b, "d": d, "rgsw_a": ra, "rgsw_b": rb, "ct_a": ca, "ct_b": cb,
            "ext_a": oa, "ext_b": ob, "ksk_a": ka, "ksk_b": kb, "ks_a": sa, "ks_b": sb, "auto_t": -5, "auto_a": ta,
            "auto_b": tb}


def blind_rotate_vector():
    rng = P.SplitMix64(6)
    n, log_b, d, w, n_lwe = 128, 6, 3, 3, 6
    q = next(P.two_adic_primes(54, 11))
    dec = P.Base2Decomposor(q, log_b, d)
    brk = [([rng.uniform(q, n) for _ in range(2 * d)], [rng.uniform(q, n) for _ in range(2 * d)]) for _ in range(n_lwe)]
    ts = P.ak_t(n, w)
    ak = [(t, [rng.uniform(q, n) for _ in range(d)], [rng.uniform(q, n) for _ in range(d)]) for t in ts]
    f = rng.uniform(q, n)
    lwe_a = [(rng.next() % n) * 2 + 1 for _ in range(n_lwe)]
    lwe_a[2] = 0  # a zero coefficient is skipped (bootstrapping.rs:220)
    lwe_b = rng.next() % (2 * n)
    oa, ob = P.blind_rotate(q, n, w, dec, dec, brk, ak, f, lwe_a, lwe_b)
    return {"q": q, "n": n, "log_b": log_b, "d": d, "w": w, "ak_t": ts, "brk": [[a, b] for a, b in brk],
            "ak": [[a, b] for _, a, b in ak], "f": f, "lwe_a": lwe_a, "lwe_b": lwe_b,
            "schedule": [[k, i] for k, i in P.blind_rotate_schedule(n, w, lwe_a)], "out_a": oa, "out_b": ob}


def rns_vectors():
    rng = P.SplitMix64(7)
    n = 8
    gen = P.two_adic_primes(55, 4)
    qs = [next(gen) for _ in range(2)]
    ps = [next(gen) for _ in range(2)]
    limbs = [rng.uniform(qi, n) for qi in qs]
    ext = P.rns_extend_bases(qs, limbs, ps)
    qps = qs + ps
    full = [rng.uniform(qi, n) for qi in qps]
    r2 = P.rns_rescale_k(qps, full, 2)
    r1 = P.rns_rescale_k(qps[:3], full[:3], 1)
    kb, ka = [rng.uniform(qi, n) for qi in qps], [rng.uniform(qi, n) for qi in qps]
    cb, ca = [rng.uniform(qi, n) for qi in qs], [rng.uniform(qi, n) for qi in qs]
    ob, oa = P.ckks_key_switch(qs, ps, kb, ka, cb, ca)
    return {"n": n, "qs": qs, "ps": ps, "limbs": limbs, "extended": ext, "full": full, "rescale_k2": r2,
            "rescale_k1": r1, "ksk_b": kb, "ksk_a": ka, "ct_b": cb, "ct_a": ca, "ks_b": ob, "ks_a": oa}


def lwe_vectors():
    rnd = random.Random(8)
    big_q = next(P.two_adic_primes(54, 11))
    ms = []
    for _ in range(64):
        v = rnd.randrange(big_q)
        ms.append({"v": v, "to_2p16": P.zq_mod_switch(big_q, v, 1 << 16)})
    odd = []
    for v in [0, 1, 2, 31, 32, 33, 65535, 65534, 40000, 12345, 17, 16, 15]:
        odd.append({"v": v, "to_2048": P.zq_mod_switch_odd(1 << 16, v, 2048)})
    return {"big_q": big_q, "mod_switch": ms, "q_ks": 1 << 16, "mod_switch_odd": odd}


if __name__ == "__main__":
    dump("moduli.json", moduli())
    dump("ntt.json", ntt_vectors())
    dump("decompose.json", decompose_vectors())
    dump("automorphism.json", automorphism_vectors())
    dump("rlwe.json", rlwe_vectors())
    dump("blind_rotate.json", blind_rotate_vector())
    dump("rns.json", rns_vectors())
    dump("lwe.json", lwe_vectors())
