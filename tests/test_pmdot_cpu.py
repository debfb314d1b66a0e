"""Host check of the base conversions' arithmetic (learn-fhe_amd/csrc/pm_dot.hpp: unreduced dot products modulo pseudo-Mersenne
primes, util/src/ring/rns.rs:103-132, 331-345) against unsigned __int128 -- the header compiles for the host, so the bounds the
GPU kernels rely on are exercised here without a GPU (tests/pmdot_host_test.cpp)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_pm_dot_against_int128(tmp_path):
    exe = str(tmp_path / "pmdot_host_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "pmdot_host_test.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "passed" in out.stdout
