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
