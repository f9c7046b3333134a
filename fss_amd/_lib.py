"""ctypes binding of libfss_amd.so (include/fss_amd.h).

There is no CPU fallback: if the HIP library is missing or no GPU is usable the
calls raise (ImportError / RuntimeError).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("FSS_AMD_LIB") or os.path.join(_HERE, "libfss_amd.so")  # FSS_AMD_LIB: A/B builds while tuning

OK, EINVAL, EHIP, EUNSUPPORTED = 0, -1, -2, -3

_lib = None

# name -> (restype, argtypes); must list every symbol include/fss_amd.h declares.
_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
PROTOTYPES = {
    "fss_amd_ctx_create": (_i, [C.POINTER(_vp), _i, _vp, _vp]),
    "fss_amd_ctx_destroy": (None, [_vp]),
    "fss_amd_ctx_device": (_i, [_vp]),
    "fss_amd_ctx_set_prg": (_i, [_vp, _i, _vp, _i]),
    "fss_amd_strerror": (C.c_char_p, [_i]),
    "fss_amd_last_hip_error": (_i, []),
    "fss_amd_prg_gen": (_i, [_vp, _i, _vp, _i64, _vp, _vp]),
    "fss_amd_dpf_eval_points": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "fss_amd_dcf_eval_points": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "fss_amd_ht_eval_points": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "fss_amd_eval_points_host": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "fss_amd_eval_points_workspace": (_sz, [_i, _i, _i]),
    "fss_amd_ctx_set_points_plan": (_i, [_vp, _i]),
    "fss_amd_points_prefix_bits": (_i, [_i, _i, _i64]),
    "fss_amd_dpf_eval_points_shared": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i64, _vp, _i, _vp, _sz, _vp]),
    "fss_amd_dcf_eval_points_shared": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _i64, _vp, _i, _vp, _sz, _vp]),
    "fss_amd_ht_eval_points_shared": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i64, _vp, _i, _vp, _sz, _vp]),
    "fss_amd_dpf_relayout": (_i, [_vp, _i, _vp, _i64, _vp, _vp, _vp, _vp]),
    "fss_amd_dcf_relayout": (_i, [_vp, _i, _vp, _i64, _vp, _vp, _vp, _vp]),
    "fss_amd_ht_relayout": (_i, [_vp, _i, _vp, _i64, _vp, _vp, _vp]),
    "fss_amd_dpf_eval_keys": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "fss_amd_dcf_eval_keys": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "fss_amd_ht_eval_keys": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "fss_amd_eval_all_workspace": (_sz, [_i, _i, _i64]),
    "fss_amd_dpf_eval_all": (_i, [_vp, _i, _i, _i, _vp, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "fss_amd_dcf_eval_all": (_i, [_vp, _i, _i, _i, _vp, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "fss_amd_ht_eval_all": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "fss_amd_grotto_preprocess": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fss_amd_grotto_eval_table": (_i, [_vp, _i, _vp, _vp, _i, _i64, _vp, _vp]),
    "fss_amd_grotto_eval_all": (_i, [_vp, _i, _i, _vp, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "fss_amd_grotto_eval_keys": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "fss_amd_dpf_gen": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i64, _vp, _vp]),
    "fss_amd_dcf_gen": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _i64, _vp, _vp]),
    "fss_amd_ht_gen": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i64, _vp, _vp, _vp]),
    "fss_amd_eval_all_batch_workspace": (C.c_size_t, [_i, _i, _i64]),
    "fss_amd_dpf_eval_all_batch": (_i, [_vp, _i, _i, _i, _vp, _vp, _i64, _vp, _vp, C.c_size_t, _vp]),
    "fss_amd_dcf_eval_all_batch": (_i, [_vp, _i, _i, _i, _vp, _vp, _i64, _vp, _vp, C.c_size_t, _vp]),
    "fss_amd_ht_eval_all_batch": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _vp, C.c_size_t, _vp]),
    "fss_amd_dpf_key_major": (_i, [_vp, _i, _vp, _vp, _vp, _i64, _vp, _vp]),
    "fss_amd_dcf_key_major": (_i, [_vp, _i, _vp, _vp, _vp, _i64, _vp, _vp]),
    "fss_amd_ht_key_major": (_i, [_vp, _i, _vp, _vp, _i64, _vp, _vp]),
    "fss_amd_ctx_set_hash_iv": (_i, [_vp, _vp]),
    "fss_amd_ctx_set_hash_sha256": (_i, [_vp, _vp]),
    "fss_amd_ctx_set_uint_modulus": (_i, [_vp, C.c_uint64]),
    "fss_amd_vdpf_gen": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "fss_amd_vdpf_gen_level_major": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fss_amd_vdpf_relayout": (_i, [_vp, _i, _vp, _i64, _vp, _vp, _vp]),
    "fss_amd_vdpf_eval_points": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _vp]),
    "fss_amd_vdpf_eval_keys": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _vp]),
    "fss_amd_vdpf_prove": (_i, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "fss_amd_vdpf_prove_keys": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "fss_amd_dpf_gen_level_major": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i64, _vp, _vp, _vp, _vp]),
    "fss_amd_dcf_gen_level_major": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _i64, _vp, _vp, _vp, _vp]),
    "fss_amd_ht_gen_level_major": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i64, _vp, _vp, _vp, _vp]),
    "fss_amd_last_kernel": (C.c_char_p, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "fss_amd_dpf_relayout_u32": (_i, [_vp, _i, _vp, _i64, _vp, _vp, _vp, _vp]),
    "fss_amd_dpf_eval_keys_u32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "fss_amd_vdpf_relayout_u32": (_i, [_vp, _i, _vp, _i64, _vp, _vp, _vp]),
    "fss_amd_vdpf_eval_keys_u32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _vp]),
    "fss_amd_batch_slice": (_i, [_i64, _i, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "fss_amd_domain_block": (_i, [_i, _i, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "fss_amd_grotto_carries": (_i, [_vp, _i, _vp]),
    "fss_amd_grotto_apply_carry": (_i, [_vp, _vp, _i64, _i, _vp]),
    "fss_amd_ctx_scratch": (_i, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "fss_amd_ctx_scratch_release": (_i, [_vp, _vp]),
    "fss_amd_upload": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "fss_amd_ctx_read_te0": (_i, [_vp, _vp]),
}


def _bind(path):
    L = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


def lib():
    """Load libfss_amd.so; raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise ImportError(
                f"{SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). fss_amd has no CPU fallback.")
        _lib = _bind(SO)
    return _lib


DIAG_SO = os.path.join(_HERE, "libfss_amd_diag.so")
_diag = None


def diag_lib():
    """The diagnostic build of the same sources (-DFSS_DIAG_STAMPS=1: every wave of the walks records s_memtime /
    s_memrealtime; __graft_entry__.build() makes it), or None when it has not been built.  Measurement only
    (bench.py's in-kernel clock): the product library executes no stamp."""
    global _diag
    if _diag is None and os.path.exists(DIAG_SO):
        try:
            from . import _build
            stamp = DIAG_SO + ".sources.sha256"
            if not os.path.exists(stamp) or open(stamp).read().strip() != _build._sources_digest() + " FSS_DIAG_STAMPS=1":
                return None  # built from other sources than the product library: its clock would be another kernel's
            L = _bind(DIAG_SO)
            L.fss_amd_diag_read_stamps.restype = _i
            L.fss_amd_diag_read_stamps.argtypes = [_vp, _vp, _i]
            _diag = L
        except (OSError, AttributeError):
            return None
    return _diag


class FssAmdError(RuntimeError):
    pass


def check(rc, what):
    if rc != OK:
        L = lib()
        msg = L.fss_amd_strerror(rc).decode()
        extra = f" (hipError_t {L.fss_amd_last_hip_error()})" if rc == EHIP else ""
        raise FssAmdError(f"{what}: {msg}{extra}")
