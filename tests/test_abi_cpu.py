"""No-GPU checks of the product library: it loads, exports every symbol include/*.h declares, and its
host-side set-up logic (primes, generator, omega, twiddle tables, status codes) matches the oracle and the
golden vectors.  No compute entry point is exercised here (they need a GPU and have no CPU fallback)."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(fhe_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(fhe):
    lib = fhe.lib()
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert b"gfx950" in lib.fhe_version()


def test_host_setup_matches_golden(fhe):
    g = load_golden("moduli.json")
    for m in g["moduli"]:
        ctx = fhe.NttContext(m["q"], device=-1)
        info = ctx.info()
        assert (info["q"], info["s"], info["g"], info["omega"]) == (m["q"], m["s"], m["g"], m["omega"])
        assert [int(x) for x in ctx.twiddles(16)] == m["tw"]
        assert [int(x) for x in ctx.twiddles(16, inverse=True)] == m["twi"]
    out = (C.c_uint64 * 16)()
    assert fhe.lib().fhe_two_adic_primes(60, 16, 16, out) == 16
    assert list(out) == g["cfg4_qs"] + g["cfg4_ps"]


def test_twiddle_table_matches_oracle_full(fhe, cref):
    q = 1073707009  # s = 11: the whole 2^10-entry table
    ctx = fhe.NttContext(q, device=-1)
    s, g, w, tw, twi = cref.twiddle_info(q, 1 << 10)
    assert np.array_equal(ctx.twiddles(1 << 10), tw)
    assert np.array_equal(ctx.twiddles(1 << 10, inverse=True), twi)
    # a prime whose 2-adicity exceeds the cap still serves the reference's prefix (tables are prefix-nested)
    q = 1152921504606584833  # first cfg4 prime
    ctx = fhe.NttContext(q, device=-1)
    s, g, w, tw, twi = cref.twiddle_info(q, 4096)
    assert ctx.info()["s"] == s
    assert np.array_equal(ctx.twiddles(4096), tw) and np.array_equal(ctx.twiddles(4096, inverse=True), twi)


def test_status_codes(fhe):
    lib = fhe.lib()
    h = C.c_void_p()
    assert lib.fhe_ctx_create(C.c_uint64(1 << 16), -1, C.byref(h)) == 2  # FHE_ERR_NOT_PRIME (fft/zq.rs:44)
    assert lib.fhe_ctx_create(C.c_uint64(15), -1, C.byref(h)) == 2
    assert lib.fhe_ctx_create(C.c_uint64((1 << 63) - 25), -1, C.byref(h)) in (2, 6)  # >= 2^62: unsupported
    assert lib.fhe_is_prime(C.c_uint64(18014398509404161)) == 1
    assert lib.fhe_is_prime(C.c_uint64(18014398509404163)) == 0
    ctx = fhe.NttContext(1073707009, device=-1)
    a = np.arange(8, dtype=np.uint64)
    with pytest.raises(fhe.FheError) as e:
        ctx.ntt_(a, 8)
    assert e.value.code == 7  # FHE_ERR_NO_DEVICE: host-only context, no CPU fallback
    rc = lib.fhe_ntt_fwd(ctx.handle, a.ctypes.data_as(C.c_void_p), 6, 1, 0, None)
    assert rc == 1  # not a power of two (ring.rs:62)
    big = np.zeros(2048, dtype=np.uint64)
    rc = lib.fhe_ntt_fwd(ctx.handle, big.ctypes.data_as(C.c_void_p), 2048, 1, 0, None)
    assert rc == 3  # n > 2^(s-1): no 2n-th root (fft.rs:45)


def test_host_mirror_of_the_gate_logic(fhe):
    """Host-side pieces of `Bootstrapping`/`Fhew` that need no device: automorphism exponents (bootstrapping.rs:86-89), Q/8 and
    Q/4 (bootstrapping.rs:62-68), the gate look-up polynomial (fhew.rs:31-36) -- against the oracle's restatement."""
    from oracle import pyref as P
    for n, w in [(8, 2), (128, 3), (512, 10), (1024, 10), (2048, 10)]:
        assert fhe.ak_t(n, w) == P.ak_t(n, w)
    for q in (268369921, 18014398509404161, 1152921504606748673):
        assert fhe.Fhew._round_div(q, 8) == P.zq_from_f64(q, float(q) / 8.0)
        assert fhe.Fhew._round_div(q, 4) == P.zq_from_f64(q, float(q) / 4.0)

    class _Key:  # the two attributes Fhew reads from a BootstrapKey
        class ctx:
            q = 268369921

        class brk:
            n = 512

    ev = fhe.Fhew(_Key, 1 << 16, 4, 4, None, None)
    f = ev.table_poly(fhe.Fhew.TABLES["nand"], np.zeros(1, dtype=np.uint64))
    q8 = ev.big_q_by_8
    assert f.shape == (512,) and list(f[:128]) == [q8] * 128 and list(f[384:]) == [268369921 - q8] * 128


def test_header_is_plain_c_and_links(tmp_path, fhe):
    """include/fhe_ring.h is the drop-in boundary: it must compile as C99 (no C++-isms) and every declared function must
    resolve against the built library from a C translation unit (no compute call: there is no GPU here)."""
    import re
    import subprocess
    header = os.path.join(ROOT, "include", "fhe_ring.h")
    names = sorted(set(re.findall(r"\b(fhe_[a-z0-9_]+)\s*\(", open(header).read())))
    src = tmp_path / "abi.c"
    src.write_text('#include "fhe_ring.h"\n#include <stdio.h>\nint main(void) {\n  void (*f[])(void) = {%s};\n'
                   '  unsigned n = 0; for (unsigned i = 0; i < sizeof f / sizeof f[0]; ++i) n += f[i] != 0;\n'
                   '  printf("%%u %%s\\n", n, fhe_version());\n  return 0;\n}\n' % ", ".join("(void (*)(void))%s" % x for x in names))
    lib_dir = os.path.dirname(fhe.lib_path())
    exe = tmp_path / "abi"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", lib_dir,
           "-lfhe_ring", "-Wl,--allow-shlib-undefined", "-Wl,-rpath," + lib_dir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert len(names) >= 50


def test_tiny_moduli_follow_the_reference(fhe):
    """util/src/zq.rs:99-105 searches the generator in 1..q-1 (exclusive): q = 3 has no candidate and the reference panics there;
    the boundary returns a status.  q = 5, 7, 17: smallest non-residue, 2-adicity and root as the oracle derives them."""
    from oracle import pyref as P
    with pytest.raises(fhe.FheError):
        fhe.NttContext(3, device=-1)
    for q in (5, 7, 17, 97, 12289):
        info = fhe.NttContext(q, device=-1).info()
        g = P.generator(q)
        s = (q - 1 & -(q - 1)).bit_length() - 1
        assert (info["g"], info["s"], info["omega"]) == (g, s, pow(g, (q - 1) >> s, q))


def _chacha20_block_py(key: bytes, nonce: int, counter: int) -> bytes:
    """An independent ChaCha20 block (original layout: 64-bit block counter, 64-bit nonce), plain Python."""
    import struct
    m32 = 0xFFFFFFFF
    rot = lambda x, r: ((x << r) | (x >> (32 - r))) & m32  # noqa: E731
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(struct.unpack("<8I", key)) + [counter & m32, counter >> 32, nonce & m32, nonce >> 32]
    x = list(s)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & m32; x[d] = rot(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & m32; x[b] = rot(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & m32; x[d] = rot(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & m32; x[b] = rot(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return struct.pack("<16I", *[(a + b) & m32 for a, b in zip(x, s)])


def test_chacha20_block_known_answers(fhe):
    """The generator behind every key-material producer (csrc/keygen_kernels.hpp `chacha20_block`, host-callable through
    fhe_chacha20_block): the published ChaCha20 test vectors (RFC 8439 appendix A.1 #1 - #3: all-zero key, block counters 0 and 1;
    key 00..01, counter 1 -- with a zero nonce the original 64/64 layout and the RFC's 32/96 layout coincide), and an independent
    Python implementation of the block function on random keys, nonces and 64-bit counters (the layout the library uses)."""
    z = bytes(32)
    assert fhe.chacha20_block(z, 0, 0).hex() == ("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                                                 "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    assert fhe.chacha20_block(z, 0, 1).hex() == ("9f07e7be5551387a98ba977c732d080dcb0f29a048e3656912c6533e32ee7aed"
                                                 "29b721769ce64e43d57133b074d839d531ed1f28510afb45ace10a1f4b794d6f")
    assert fhe.chacha20_block(bytes(31) + b"\x01", 0, 1).hex() == ("3aeb5224ecf849929b9d828db1ced4dd832025e8018b8160b82284f3c949aa5a"
                                                                  "8eca00bbb4a73bdad192b5c42f73f2fd4e273644c8b36125a64addeb006c13a0")
    import random
    rnd = random.Random(5)
    for _ in range(50):
        key = bytes(rnd.getrandbits(8) for _ in range(32))
        nonce, counter = rnd.getrandbits(64), rnd.getrandbits(64)
        assert fhe.chacha20_block(key, nonce, counter) == _chacha20_block_py(key, nonce, counter)


def test_rng_handles(fhe):
    """fhe_rng: 32 caller bytes, operating-system entropy (NULL key), the 64-bit test form; wrong sizes and NULL are errors."""
    r1, r2 = fhe.Rng(key=bytes(range(32))), fhe.Rng()
    assert r1._h and r2._h
    with pytest.raises(ValueError):
        fhe.Rng(key=b"short")
    lib = fhe.lib()
    assert lib.fhe_rng_create(None, None) == 1  # FHE_ERR_INVALID
    out = (C.c_uint64 * 4)()
    assert lib.fhe_sample_torus(None, 0, out, 4, 0, None) == 1  # no generator: FHE_ERR_INVALID (before anything touches a device)
