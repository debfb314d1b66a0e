"""ctypes loader for lib/libfhe_ring.so.  Raises if the library is absent: no fallback of any kind."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# FHE_RING_LIB: developer override (tools/scripts/occ_sweep.sh: a build variant of the same library), never a fallback
_SO = os.environ.get("FHE_RING_LIB") or os.path.join(_HERE, "lib", "libfhe_ring.so")
_lib = None

STATUS = {0: "FHE_OK", 1: "FHE_ERR_INVALID", 2: "FHE_ERR_NOT_PRIME", 3: "FHE_ERR_NO_ROOT", 4: "FHE_ERR_MODULUS",
          5: "FHE_ERR_HIP", 6: "FHE_ERR_UNSUPPORTED", 7: "FHE_ERR_NO_DEVICE"}
MEM_HOST, MEM_DEVICE = 0, 1


class FheError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        extra = ""
        if code == 5 and _lib is not None:
            extra = " (hipError %d)" % _lib.fhe_last_hip_error()
        super().__init__("%s: %s%s" % (where, STATUS.get(code, str(code)), extra))


def lib_path() -> str:
    return _SO


def build(force: bool = False) -> str:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-s"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return _SO


u64p = C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise FileNotFoundError(
                "%s not found: build it with `make -C learn-fhe_amd/csrc` (or __graft_entry__.build()); "
                "there is no CPU fallback" % _SO)
        # PyTorch-ROCm bundles its own libamdhip64.so.7; whichever HIP runtime is loaded first serves the
        # whole process.  When torch is part of the process (tests, bench: it owns the HBM buffers and the
        # stream we are handed) it must be loaded first so that both sides share ONE runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_SO)
        L.fhe_version.restype = C.c_char_p
        L.fhe_is_prime.argtypes = [C.c_uint64]
        L.fhe_two_adic_primes.argtypes = [C.c_int, C.c_int, C.c_int, u64p]
        L.fhe_ctx_create.argtypes = [C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]
        L.fhe_ctx_destroy.argtypes = [C.c_void_p]
        L.fhe_ctx_destroy.restype = None
        L.fhe_ctx_info.argtypes = [C.c_void_p, u64p, C.POINTER(C.c_int), u64p, u64p]
        L.fhe_ctx_twiddles.argtypes = [C.c_void_p, C.c_int, u64p, C.c_size_t]
        for name in ("fhe_ntt_fwd", "fhe_ntt_inv"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
        L.fhe_ntt_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
        L.fhe_pointwise_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        i64p = C.POINTER(C.c_int64)
        u32p = C.POINTER(C.c_uint32)
        vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
        L.fhe_decompose.argtypes = [C.c_uint64, ci, ci, vp, sz, sz, vp, ci, vp]
        L.fhe_automorphism.argtypes = [C.c_uint64, C.c_int64, vp, vp, sz, sz, ci, vp]
        L.fhe_monomial_mul.argtypes = [C.c_uint64, C.c_int64, vp, vp, sz, sz, ci, vp]
        for name in ("fhe_rgsw_prepare", "fhe_ksk_prepare"):
            getattr(L, name).argtypes = [vp, ci, ci, vp, vp, sz, sz, ci, C.POINTER(vp)]
        L.fhe_key_destroy.argtypes = [vp]
        L.fhe_key_destroy.restype = None
        for name in ("fhe_external_product", "fhe_rlwe_key_switch", "fhe_rgsw_internal_product"):
            getattr(L, name).argtypes = [vp, vp, sz, vp, vp, sz, ci, vp]
        L.fhe_rlwe_automorphism.argtypes = [vp, vp, sz, C.c_int64, vp, vp, sz, ci, vp]
        L.fhe_bootstrap_key_create.argtypes = [vp, vp, vp, i64p, ci, C.POINTER(vp)]
        L.fhe_bootstrap_key_destroy.argtypes = [vp]
        L.fhe_bootstrap_key_destroy.restype = None
        L.fhe_bootstrap_key_status.argtypes = [vp, vp, ci]
        L.fhe_blind_rotate.argtypes = [vp, vp, vp, vp, sz, vp, vp, sz, ci, vp, u32p, u32p]
        L.fhe_rns_ctx_create.argtypes = [u64p, ci, u64p, ci, ci, C.POINTER(vp)]
        L.fhe_rns_ctx_destroy.argtypes = [vp]
        L.fhe_rns_ctx_destroy.restype = None
        L.fhe_rns_extend_bases.argtypes = [vp, vp, vp, sz, sz, ci, vp]
        L.fhe_rns_rescale_k.argtypes = [vp, vp, vp, sz, sz, ci, vp]
        L.fhe_rns_switch_bases.argtypes = [vp, ci, vp, vp, sz, sz, ci, vp]
        L.fhe_ckks_ksk_prepare.argtypes = [vp, vp, vp, sz, ci, C.POINTER(vp)]
        L.fhe_ckks_key_destroy.argtypes = [vp]
        L.fhe_ckks_key_destroy.restype = None
        L.fhe_ckks_key_switch.argtypes = [vp, vp, vp, vp, sz, ci, vp]
        L.fhe_ckks_shard_create.argtypes = [vp, vp, ci, ci, ci, ci, C.POINTER(vp)]
        L.fhe_ckks_shard_destroy.argtypes = [vp]
        L.fhe_ckks_shard_destroy.restype = None
        L.fhe_ckks_shard_products.argtypes = [vp, vp, vp, vp, sz, ci, vp]
        L.fhe_ckks_shard_finish.argtypes = [vp, vp, vp, vp, vp, vp, sz, ci, vp]
        L.fhe_lwe_mod_switch.argtypes = [C.c_uint64, C.c_uint64, vp, vp, sz, ci, ci, vp]
        L.fhe_rq_add.argtypes = [C.c_uint64, vp, vp, vp, sz, ci, vp]
        L.fhe_rq_sub.argtypes = [C.c_uint64, vp, vp, vp, sz, ci, vp]
        L.fhe_rq_neg.argtypes = [C.c_uint64, vp, vp, sz, ci, vp]
        L.fhe_rq_scalar_mul.argtypes = [C.c_uint64, vp, C.c_uint64, vp, sz, ci, vp]
        L.fhe_rq_from_i64.argtypes = [C.c_uint64, vp, vp, sz, ci, vp]
        L.fhe_rns_ntt_fwd.argtypes = [vp, ci, vp, sz, sz, ci, vp]
        L.fhe_rns_ntt_inv.argtypes = [vp, ci, vp, sz, sz, ci, vp]
        L.fhe_rns_pointwise_mul.argtypes = [vp, ci, vp, vp, sz, sz, ci, vp]
        L.fhe_lwe_lincomb.argtypes = [C.c_uint64, ci, vp, vp, C.c_uint64, vp, sz, ci, vp]
        L.fhe_lwe_key_switch.argtypes = [C.c_uint64, ci, ci, vp, vp, vp, vp, sz, sz, vp, vp, sz, ci, vp]
        L.fhe_rlwe_sample_extract.argtypes = [C.c_uint64, vp, vp, sz, sz, C.c_uint64, vp, vp, sz, ci, vp]
        L.fhe_fhew_bootstrap.argtypes = [vp, C.c_uint64, ci, ci, vp, vp, vp, sz, C.c_uint64, vp, vp, vp, vp, sz, ci, vp]
        L.fhe_torus_ctx_create.argtypes = [ci, C.POINTER(vp)]
        L.fhe_torus_ctx_destroy.argtypes = [vp]
        L.fhe_torus_ctx_destroy.restype = None
        L.fhe_torus_decompose.argtypes = [ci, ci, vp, sz, sz, vp, ci, vp]
        L.fhe_torus_mul.argtypes = [vp, vp, vp, ci, sz, sz, ci, vp]
        L.fhe_tggsw_prepare.argtypes = [vp, ci, ci, vp, vp, sz, sz, ci, C.POINTER(vp)]
        L.fhe_tggsw_prepare_fft64.argtypes = [vp, ci, ci, vp, vp, sz, sz, ci, C.POINTER(vp)]
        L.fhe_tggsw_key_destroy.argtypes = [vp]
        L.fhe_tggsw_key_destroy.restype = None
        L.fhe_tggsw_external_product.argtypes = [vp, vp, sz, vp, vp, sz, ci, vp]
        L.fhe_tfhe_mod_switch.argtypes = [vp, vp, sz, sz, ci, vp]
        L.fhe_tggsw_cmux.argtypes = [vp, vp, sz, vp, vp, vp, vp, vp, vp, sz, ci, vp]
        L.fhe_tglwe_rotate.argtypes = [vp, vp, sz, C.c_int64, vp, vp, sz, ci, vp]
        L.fhe_tfhe_blind_rotate.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, ci, vp]
        L.fhe_tglwe_sample_extract.argtypes = [vp, vp, sz, sz, vp, vp, sz, ci, vp]
        L.fhe_tlwe_key_switch.argtypes = [ci, ci, vp, vp, vp, vp, sz, sz, vp, vp, sz, ci, vp]
        L.fhe_trim.argtypes = []
        L.fhe_set_option.argtypes = [C.c_char_p, C.c_long]
        L.fhe_rng_create.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.fhe_rng_create_from_seed.argtypes = [C.c_uint64, C.POINTER(vp)]
        L.fhe_rng_destroy.argtypes = [vp]
        L.fhe_rng_destroy.restype = None
        L.fhe_chacha20_block.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_char_p]
        u64, dbl = C.c_uint64, C.c_double
        L.fhe_sample_zo.argtypes = [dbl, vp, u64, vp, sz, ci, vp]
        L.fhe_ckks_sk_encrypt.argtypes = [vp, ci, vp, vp, sz, sz, vp, u64, vp, vp, ci, vp]
        L.fhe_ckks_ksk_gen.argtypes = [vp, vp, vp, sz, vp, u64, vp, vp, ci, vp]
        L.fhe_sample_tdg.argtypes = [dbl, vp, u64, vp, sz, ci, vp]
        L.fhe_sample_binary.argtypes = [vp, u64, vp, sz, ci, vp]
        L.fhe_tlwe_sk_encrypt.argtypes = [vp, vp, sz, sz, dbl, vp, u64, vp, vp, ci, vp]
        L.fhe_tlwe_ksk_gen.argtypes = [ci, ci, vp, sz, vp, sz, dbl, vp, u64, vp, vp, ci, vp]
        L.fhe_tglwe_sk_encrypt.argtypes = [vp, vp, vp, sz, sz, dbl, vp, u64, vp, vp, ci, vp]
        L.fhe_tggsw_encrypt.argtypes = [vp, ci, ci, vp, vp, sz, sz, dbl, vp, u64, vp, vp, ci, vp]
        L.fhe_rns_rescale.argtypes = [vp, vp, vp, sz, sz, ci, vp]
        L.fhe_rns_automorphism.argtypes = [vp, C.c_int64, vp, vp, sz, sz, ci, vp]
        L.fhe_ckks_rotate.argtypes = [vp, vp, C.c_int64, vp, vp, sz, ci, vp]
        L.fhe_ckks_mul.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz, ci, vp]
        L.fhe_lwe_sk_encrypt.argtypes = [C.c_uint64, vp, vp, sz, sz, vp, C.c_uint64, vp, vp, ci, vp]
        L.fhe_lwe_ksk_gen.argtypes = [C.c_uint64, ci, ci, vp, sz, vp, sz, vp, C.c_uint64, vp, vp, ci, vp]
        L.fhe_rq_sum.argtypes = [C.c_uint64, vp, sz, sz, vp, ci, vp]
        L.fhe_sample_uniform.argtypes = [C.c_uint64, vp, C.c_uint64, vp, sz, ci, vp]
        L.fhe_sample_torus.argtypes = [vp, C.c_uint64, vp, sz, ci, vp]
        L.fhe_sample_dg.argtypes = [C.c_uint64, C.c_double, ci, vp, C.c_uint64, vp, sz, ci, vp]
        L.fhe_power_up.argtypes = [C.c_uint64, ci, ci, vp, sz, sz, vp, ci, vp]
        L.fhe_rlwe_sk_encrypt.argtypes = [vp, vp, vp, sz, sz, vp, C.c_uint64, vp, vp, ci, vp]
        L.fhe_rgsw_encrypt.argtypes = [vp, ci, ci, vp, vp, sz, sz, vp, C.c_uint64, vp, vp, ci, vp]
        L.fhe_rlwe_ksk_gen.argtypes = [vp, ci, ci, vp, vp, C.c_int64, sz, vp, C.c_uint64, vp, vp, ci, vp]
        L.fhe_tfhe_bootstrap.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, sz, ci, vp]
        L.fhe_rlwe_share_encrypt.argtypes = [vp, vp, sz, vp, vp, sz, sz, vp, u64, vp, ci, vp]
        L.fhe_rlwe_pk_encrypt.argtypes = [vp, vp, vp, vp, sz, sz, vp, u64, vp, vp, ci, vp]
        L.fhe_rlwe_decrypt.argtypes = [vp, vp, vp, vp, sz, sz, vp, ci, vp]
        L.fhe_rgsw_pk_encrypt.argtypes = [vp, ci, ci, vp, vp, vp, sz, sz, vp, u64, vp, vp, ci, vp]
        L.fhe_lwe_share_encrypt.argtypes = [u64, vp, vp, vp, sz, sz, vp, u64, vp, ci, vp]
        L.fhe_lwe_ksk_share_gen.argtypes = [u64, ci, ci, vp, vp, sz, vp, sz, vp, u64, vp, ci, vp]
        L.fhe_rns_add.argtypes = [vp, ci, vp, vp, sz, sz, ci, vp]
        L.fhe_rns_sub.argtypes = [vp, ci, vp, vp, sz, sz, ci, vp]
        L.fhe_rns_neg.argtypes = [vp, ci, vp, sz, sz, ci, vp]
        L.fhe_ckks_pk_encrypt.argtypes = [vp, vp, vp, vp, sz, sz, vp, u64, vp, vp, ci, vp]
        L.fhe_ckks_decrypt.argtypes = [vp, vp, vp, vp, sz, sz, vp, ci, vp]
        L.fhe_ckks_mul_plain.argtypes = [vp, vp, sz, vp, vp, vp, vp, sz, sz, ci, vp]
        # any TGLWE rank k (torusk_api.hip)
        L.fhe_tggswk_prepare.argtypes = [vp, ci, ci, ci, vp, sz, sz, ci, C.POINTER(vp)]
        L.fhe_tggswk_key_destroy.argtypes = [vp]
        L.fhe_tggswk_key_destroy.restype = None
        L.fhe_tggswk_external_product.argtypes = [vp, vp, sz, vp, sz, ci, vp]
        L.fhe_tggswk_cmux.argtypes = [vp, vp, sz, vp, vp, vp, sz, ci, vp]
        L.fhe_tglwek_rotate.argtypes = [vp, ci, sz, C.c_int64, vp, sz, ci, vp]
        L.fhe_tglwek_sample_extract.argtypes = [vp, ci, sz, sz, vp, vp, sz, ci, vp]
        L.fhe_tfhek_blind_rotate.argtypes = [vp, vp, vp, vp, vp, vp, sz, ci, vp]
        L.fhe_tfhek_bootstrap.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, sz, ci, vp]
        L.fhe_tglwek_sk_encrypt.argtypes = [vp, ci, vp, vp, sz, sz, dbl, vp, u64, vp, ci, vp]
        L.fhe_tggswk_encrypt.argtypes = [vp, ci, ci, ci, vp, vp, sz, sz, dbl, vp, u64, vp, ci, vp]
        _lib = L
    return _lib


def set_option(name: str, value: int) -> None:
    """fhe_set_option: a lab switch of the library (include/fhe_ring.h); none of them changes a result."""
    check(lib().fhe_set_option(name.encode(), int(value)), "fhe_set_option(%s)" % name)


def check(rc, where):
    if rc != 0:
        raise FheError(rc, where)
