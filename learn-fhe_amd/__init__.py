"""learn-fhe_amd: MI355X (gfx950) polynomial-ring backend for the FHE hot path of han0110/learn-fhe.

The product is the C-ABI shared library `lib/libfhe_ring.so` (include/fhe_ring.h) built from the HIP
sources in `csrc/`.  This Python package is only the harness-side binding used by tests and bench:
it loads the library with ctypes and fails loudly if it is missing -- there is no CPU fallback.
"""
from ._lib import FheError, build, lib, lib_path, set_option  # noqa: F401
from .ring import (BootstrapKey, CkksKey, CkksShard, Fhew, GadgetKey, NttContext, RnsContext, TggswKey, TorusContext,  # noqa: F401
                   ak_t, automorphism, decompose, lwe_key_switch, lwe_lincomb, lwe_mod_switch, monomial_mul, rlwe_sample_extract, rq_add, rq_from_i64, rq_neg, rq_scalar_mul, rq_sub,
                   tglwe_sample_extract, tlwe_key_switch, torus_decompose, tglwe_rotate,
                   power_up, rgsw_encrypt, rlwe_ksk_gen, rlwe_sk_encrypt, sample_dg, sample_torus, sample_uniform,
                   lwe_ksk_gen, lwe_sk_encrypt, rq_sum,
                   sample_binary, sample_tdg, sample_zo, tggsw_encrypt, tglwe_sk_encrypt, tlwe_ksk_gen, tlwe_sk_encrypt,
                   rlwe_share_encrypt, rlwe_pk_encrypt, rlwe_decrypt, rgsw_pk_encrypt, lwe_share_encrypt, lwe_ksk_share_gen,
                   TggswKeyK, tglwek_rotate, tglwek_sample_extract, tglwek_sk_encrypt, tggswk_encrypt,
                   Rng, STREAM_AUTO, chacha20_block)
