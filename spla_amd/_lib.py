"""ctypes binding of libspla_amd.so's kernel-level C ABI (include/spla_hip.h).

There is no CPU fallback: if the shared library is missing this module raises at import of the symbols, and every
call checks the returned status and raises ``SplaHipError`` with the library's own message.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libspla_amd.so"
if os.environ.get("SPLA_LIB_TAG"):  # experiments only: a side-by-side build made with SPLA_BUILD_TAG
    LIB_PATH = LIB_PATH.with_name(f"libspla_amd_{os.environ['SPLA_LIB_TAG']}.so")


class SplaHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"splahip status {status}: {message}")
        self.status = status


_u32p = C.POINTER(C.c_uint32)
_vp = C.c_void_p

# name -> (restype, argtypes); restype int == status unless listed in _RAW
_SIGNATURES = {
    "splahip_device_count": (C.c_int, []),
    "splahip_init": (C.c_int, [C.c_int]),
    "splahip_device_info": (C.c_int, [C.c_char_p, C.c_int]),
    "splahip_last_error": (C.c_char_p, []),
    "splahip_version": (C.c_int, []),
    "splahip_env_refresh": (C.c_int, []),
    "splahip_v_assign_masked_jit": (C.c_int, [C.c_int, C.c_int, C.c_uint32, _vp, _vp, _vp, C.c_uint32, C.c_int, C.c_char_p, C.c_int, C.c_char_p, _vp]),
    "splahip_v_eadd_fdb_dense_jit": (C.c_int, [C.c_int, C.c_uint32, _vp, _vp, _vp, C.c_uint32, C.c_char_p, _vp]),
    "splahip_v_eadd_fdb_sparse_jit": (C.c_int, [C.c_int, C.c_uint32, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_uint32), C.c_char_p, _vp]),
    "splahip_v_eadd_dense_jit": (C.c_int, [C.c_int, C.c_uint32, _vp, _vp, _vp, C.c_char_p, _vp]),
    "splahip_v_eadd_sparse_jit": (C.c_int, [C.c_int, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, C.c_uint32, _vp, _vp, C.POINTER(C.c_uint32), C.c_char_p, _vp]),
    "splahip_reduce_jit": (C.c_int, [C.c_int, C.c_uint32, _vp, C.c_uint32, C.POINTER(C.c_uint32), C.c_char_p, _vp]),
    "splahip_map_jit": (C.c_int, [C.c_int, C.c_uint32, _vp, _vp, C.c_char_p, _vp]),
    "splahip_csr_plan_set_uniform_values": (C.c_int, [_vp, C.c_int, C.c_uint32]),
    "splahip_words_range": (C.c_int, [_vp, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _vp]),
    "splahip_csr_plan_hint_products": (C.c_int, [_vp, C.c_uint64]),
    "splahip_csr_plan_first_hit": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "splahip_set_deterministic": (C.c_int, [C.c_int]),
    "splahip_set_reference_order": (C.c_int, [C.c_int]),
    "splahip_read_stream_gbs": (C.c_int, [C.c_size_t, C.c_int, C.POINTER(C.c_double), C.c_void_p]),
    "splahip_malloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "splahip_free": (C.c_int, [_vp]),
    "splahip_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "splahip_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "splahip_memcpy_d2d": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "splahip_memset32": (C.c_int, [_vp, C.c_uint32, C.c_size_t, _vp]),
    "splahip_stream_sync": (C.c_int, [_vp]),
    "splahip_csr_plan_create": (C.c_int, [C.POINTER(_vp), C.c_uint32, C.c_uint32, _vp, _vp]),
    "splahip_csr_plan_attach_sell": (C.c_int, [_vp, C.c_uint32, _vp, _vp, _vp, _vp]),
    "splahip_csr_plan_sell_info": (C.c_int, [_vp, _u32p, _u32p, _u32p, _u32p]),
    "splahip_csr_plan_sell_values": (C.c_int, [_vp, C.POINTER(C.c_int), _u32p]),
    "splahip_csr_plan_attach_cob": (C.c_int, [_vp, C.c_uint32, _vp, _vp, _vp, C.c_uint32, C.c_uint32, _vp]),
    "splahip_csr_plan_cob_info": (C.c_int, [_vp, _u32p, _u32p, _u32p, _u32p]),
    "splahip_csr_plan_footprint": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "splahip_mxv_prefers_sell": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "splahip_csr_plan_cob_layout": (C.c_int, [_vp, _u32p, _u32p, _u32p, C.POINTER(C.c_uint64)]),
    "splahip_csr_plan_cob_slots": (C.c_int, [_vp, _u32p, _u32p]),
    "splahip_csr_plan_cob_probe": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "splahip_csr_plan_cob_relabel": (C.c_int, [_vp, _u32p]),
    "splahip_jit_available": (C.c_int, []),
    "splahip_mxv_masked_jit": (C.c_int, [C.c_int, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p,
                                         C.c_uint32, C.c_int, _vp]),
    "splahip_vxm_masked_jit": (C.c_int, [_vp, C.c_int, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_int, _vp, _vp, C.POINTER(C.c_uint32), C.c_int, C.c_char_p,
                                         C.c_int, C.c_char_p, C.c_int, C.c_char_p, _vp]),
    "splahip_dist_available": (C.c_int, []),
    "splahip_dist_get_unique_id": (C.c_int, [_vp]),
    "splahip_dist_comm_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, _vp]),
    "splahip_dist_comm_destroy": (C.c_int, [_vp]),
    "splahip_dist_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "splahip_dist_all_gather": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int, _vp]),
    "splahip_shard_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "splahip_shard_destroy": (C.c_int, [_vp]),
    "splahip_shard_info": (C.c_int, [_vp, _u32p, _u32p, C.POINTER(C.c_int)]),
    "splahip_shard_split_serves": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "splahip_shard_prefers_whole": (C.c_int, [_vp, C.c_int, C.c_int]),
    "splahip_shard_mxv_local": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_uint32, _vp]),
    "splahip_shard_mxv_remote": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "splahip_shard_mxv_whole": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, _vp]),
    "splahip_dist_sx_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_uint32, C.c_uint32]),
    "splahip_dist_sx_destroy": (C.c_int, [_vp]),
    "splahip_dist_sx_pack": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _vp]),
    "splahip_dist_sx_buffers": (C.c_int, [_vp, C.c_uint32, C.c_int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "splahip_dist_sx_scatter": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, _vp]),
    "splahip_dist_sx_overflowed": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "splahip_dist_exchange_sparse": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, _vp, C.c_int, _vp]),
    "splahip_dist_mxv_masked": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_int), _vp]),
    "splahip_mxv_variant_of": (C.c_char_p, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "splahip_csr_plan_destroy": (C.c_int, [_vp]),
    "splahip_csr_plan_info": (C.c_int, [_vp, _u32p, _u32p]),
    "splahip_mxv_masked": (C.c_int, [_vp, C.c_int, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp,
                                     C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, _vp]),
    "splahip_mxv_variant": (C.c_char_p, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "splahip_v_assign_masked_dense": (C.c_int, [C.c_int, C.c_uint32, _vp, _vp, C.c_uint32, C.c_int, C.c_int, _vp]),
    "splahip_v_assign_masked_sparse": (C.c_int, [C.c_int, C.c_uint32, _vp, _vp, _vp, C.c_uint32, C.c_int, C.c_int, _vp]),
    "splahip_v_count_mf_dense": (C.c_int, [C.c_int, C.c_uint32, _vp, C.c_uint32, _u32p, _vp]),
    "splahip_v_eadd_fdb_dense": (C.c_int, [C.c_int, C.c_uint32, _vp, _vp, _vp, C.c_uint32, C.c_int, _vp]),
    "splahip_v_eadd_fdb_sparse": (C.c_int, [C.c_int, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _u32p, C.c_int, _vp]),
    "splahip_vec_dense_to_coo": (C.c_int, [C.c_int, C.c_uint32, _vp, C.c_uint32, _vp, _vp, _u32p, _vp]),
    "splahip_vec_dense_to_coo_limited": (C.c_int, [C.c_int, C.c_uint32, _vp, C.c_uint32, _vp, _vp, C.c_uint32, _u32p, _vp]),
    "splahip_vec_coo_to_dense": (C.c_int, [C.c_uint32, C.c_uint32, _vp, _vp, C.c_uint32, _vp, _vp]),
    "splahip_vxm_probe": (C.c_int, [C.c_uint32, _vp, _vp, C.POINTER(C.c_uint64), C.POINTER(C.c_int), _vp]),
    "splahip_vxm_pull": (C.c_int, [_vp, C.c_int, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_int,
                                   _vp, _vp, _u32p, C.c_int, C.c_int, C.c_int, _vp]),
    "splahip_vxm_push": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp,
                                   _vp, _vp, _u32p, C.c_int, C.c_int, C.c_int, _vp]),
    "splahip_vxm_push_n": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp,
                                     _vp, _vp, _u32p, C.c_int, C.c_int, C.c_int, C.c_uint64, _vp]),
    "splahip_vxm_push_small": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, C.c_uint32, _u32p,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, _vp]),
    "splahip_mxm_trim": (C.c_int, []),
    "splahip_mxm_last_info": (C.c_int, [_u32p, _u32p, _u32p]),
    "splahip_mxm_last_dense_ms": (C.c_int, [C.POINTER(C.c_float)]),
    "splahip_mxm_last_dense": (C.c_int, [_u32p]),
    "splahip_mxmT_masked": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32] + [_vp] * 9 + [_vp, C.POINTER(_vp), C.POINTER(_vp), _u32p,
                                      C.c_int, C.c_int, C.c_int, C.c_uint32, _vp]),
    "splahip_mxm": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32] + [_vp] * 6 + [_vp, C.POINTER(_vp), C.POINTER(_vp), _u32p, C.POINTER(C.c_uint64),
                              C.c_int, C.c_int, C.c_uint32, C.c_uint64, _vp]),
    "splahip_csr_transpose": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "splahip_m_transpose": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "splahip_map": (C.c_int, [C.c_int, C.c_uint32, _vp, _vp, C.c_int, _vp]),
}
_RAW = {"splahip_set_deterministic", "splahip_set_reference_order", "splahip_mxv_prefers_sell", "splahip_device_count", "splahip_last_error", "splahip_version", "splahip_mxv_variant", "splahip_mxv_variant_of", "splahip_dist_available",
        "splahip_shard_split_serves", "splahip_shard_prefers_whole", "splahip_jit_available"}

_lib = None


def load() -> C.CDLL:
    """Load libspla_amd.so (built by `python -m spla_amd.build`); raises if it is missing."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: build it with `python -m spla_amd.build` (there is no CPU fallback)")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import
        # torch first so libspla_amd.so binds to the runtime torch already loaded; two runtimes cannot share the GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                if os.environ.get("SPLA_LIB_TAG"):  # (experiments only: a side-by-side build of an older tree may lack a newer entry point)
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def declared_symbols():
    return sorted(_SIGNATURES)


_env_sig = None


def env_refresh():
    """Tell the library that SPLAHIP_* environment knobs have changed (it reads each knob once and caches it).  Call this after changing os.environ."""
    global _env_sig
    _env_sig = None
    load().splahip_env_refresh()


def _auto_sync() -> bool:
    # Watching the environment on every call costs a scan of os.environ per kernel dispatch (as much host time as a small kernel): only the test suite,
    # which flips knobs through os.environ / monkeypatch all the time, asks for it (pytest sets PYTEST_CURRENT_TEST; SPLA_ENV_AUTOSYNC=1 forces it).
    return "PYTEST_CURRENT_TEST" in os.environ or os.environ.get("SPLA_ENV_AUTOSYNC") == "1"


def _sync_env(lib):
    """Under the test suite: tell the library when SPLAHIP_* knobs changed since the last call (everyone else calls env_refresh() after changing one)."""
    global _env_sig
    if not _auto_sync():
        return
    sig = hash(frozenset((k, v) for k, v in os.environ.items() if k.startswith("SPLAHIP_") or k.startswith("SPLA_HIP_")))
    if sig != _env_sig:
        _env_sig = sig
        lib.splahip_env_refresh()


def call(name: str, *args):
    """Call a status-returning entry point and raise on failure."""
    lib = load()
    _sync_env(lib)
    fn = getattr(lib, name)
    res = fn(*args)
    if name in _RAW:
        return res
    if res != 0:
        raise SplaHipError(res, lib.splahip_last_error().decode(errors="replace"))
    return res
