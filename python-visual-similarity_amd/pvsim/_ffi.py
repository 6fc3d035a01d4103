"""ctypes binding of libpvsim_hip.so (C-ABI: include/pvsim.h).

The library is the product: there is no CPU fallback.  Loading fails loudly (ImportError) if the shared
object has not been built, and creating a Context fails loudly (RuntimeError) if no MI355X is visible.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PVSIM_LIB", os.path.join(_HERE, "libpvsim_hip.so"))

# pvs_status -> Python exception (same exception classes as the reference raises)
PVS_OK, PVS_ERR_INVALID, PVS_ERR_NO_DEVICE, PVS_ERR_OOM, PVS_ERR_UNSUPPORTED, PVS_ERR_DIM = range(6)
_EXC = {PVS_ERR_INVALID: ValueError, PVS_ERR_NO_DEVICE: RuntimeError, PVS_ERR_OOM: MemoryError,
        PVS_ERR_UNSUPPORTED: NotImplementedError, PVS_ERR_DIM: RuntimeError}

DESC_F32, DESC_F32_ROOTSIFT, DESC_U8_ROOTSIFT = 0, 1, 2
OPT_ASSIGN_PREFILTER, OPT_VLAD_PATH, OPT_TOPK_SELECT_ONLY, OPT_AGG_VARIANT, OPT_FISHER_SCALE = 0, 1, 2, 3, 4      # pvs_option
VLAD_PATH_AUTO, VLAD_PATH_GATHER, VLAD_PATH_STREAM, VLAD_PATH_FUSED = 0, 1, 2, 3
TIMER_NAMES = ("assign", "aggregate", "cosine_gemm", "topk", "fisher_posterior", "fisher_moments", "misc", "rescore")


class NormParams(C.Structure):
    _fields_ = [("power_norm_weight", C.c_double), ("norm_order", C.c_double), ("epsilon", C.c_double)]


_vp, _i64, _int, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_size_t
_pp = C.POINTER(C.c_void_p)
_np = C.POINTER(NormParams)

# name -> argtypes  (restype is int everywhere except where noted)
SIGNATURES = {
    "pvs_version": [],
    "pvs_last_error": [],
    "pvs_device_count": [C.POINTER(C.c_int)],
    "pvs_init": [_int, _vp, _pp],
    "pvs_destroy": [_vp],
    "pvs_sync": [_vp],
    "pvs_stream": [_vp],
    "pvs_device_name": [_vp, C.c_char_p, _sz],
    "pvs_set_option": [_vp, _int, _int],
    "pvs_get_option": [_vp, _int, C.POINTER(C.c_int)],
    "pvs_malloc": [_vp, _sz, _pp],
    "pvs_free": [_vp, _vp],
    "pvs_memcpy_h2d": [_vp, _vp, _vp, _sz],
    "pvs_memcpy_d2h": [_vp, _vp, _vp, _sz],
    "pvs_memset": [_vp, _vp, _int, _sz],
    "pvs_fill_dev": [_vp, _vp, _i64, _int, C.c_uint64],
    "pvs_stream_wait": [_vp, _vp],
    "pvs_comm_unique_id": [_vp],
    "pvs_comm_init": [_vp, _int, _int, _vp, _pp],
    "pvs_comm_destroy": [_vp],
    "pvs_comm_library": [],
    "pvs_allgather_dev": [_vp, _vp, _vp, _sz],
    "pvs_alltoall_dev": [_vp, _vp, _vp, _sz],
    "pvs_sendrecv_dev": [_vp, _int, _vp, _vp, _vp, _vp, _vp],
    "pvs_allreduce_max_f64": [_vp, _vp, _int],
    "pvs_comm_barrier": [_vp],
    "pvs_codebook_create": [_vp, _vp, _int, _int, _pp],
    "pvs_codebook_destroy": [_vp, _vp],
    "pvs_gmm_create": [_vp, _vp, _vp, _vp, _int, _int, _pp],
    "pvs_gmm_destroy": [_vp, _vp],
    "pvs_pca_create": [_vp, _vp, _vp, _int, _int, _pp],
    "pvs_pca_destroy": [_vp, _vp],
    "pvs_vlad_encode": [_vp, _vp, _vp, _vp, _int, _vp, _i64, _np, _vp, _vp],
    "pvs_vlad_encode_dev": [_vp, _vp, _vp, _vp, _int, _vp, _i64, _i64, _np, _vp, _vp, _vp],
    "pvs_kmeans_predict_dev": [_vp, _vp, _vp, _int, _i64, _vp],
    "pvs_fisher_encode": [_vp, _vp, _vp, _vp, _int, _vp, _i64, _np, _vp],
    "pvs_fisher_encode_dev": [_vp, _vp, _vp, _vp, _int, _vp, _i64, _i64, _np, _vp, _int],
    "pvs_gmm_predict_proba_dev": [_vp, _vp, _vp, _int, _i64, _vp],
    "pvs_pca_transform_dev": [_vp, _vp, _vp, _int, _i64, _vp],
    "pvs_cosine": [_vp, _vp, _i64, _vp, _i64, _i64, _int, _vp],
    "pvs_row_inv_norms_dev": [_vp, _vp, _i64, _i64, _vp],
    "pvs_cosine_dev": [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i64],
    "pvs_cosine_dual_dev": [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _i64],
    "pvs_topk_dev": [_vp, _vp, _i64, _i64, _i64, _int, _i64, _int, _vp, _vp],
    "pvs_cosine_topk_dev": [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _int, _i64, _int, _vp, _vp],
    "pvs_cosine_topk_filtered_dev": [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _int, _vp, _vp, _vp],
    "pvs_f32_to_f16_dev": [_vp, _vp, _i64, _vp],
    "pvs_cosine_f16_dev": [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i64],
    "pvs_cosine_topk_f16_dev": [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _int, _i64, _int, _vp, _vp],
    "pvs_cosine_topk": [_vp, _vp, _i64, _vp, _i64, _i64, _int, _vp, _vp],
    "pvs_cosine_topk_f64": [_vp, _vp, _i64, _vp, _i64, _i64, _int, _vp, _vp],
    "pvs_row_inv_norms_f64_dev": [_vp, _vp, _i64, _i64, _vp],
    "pvs_cosine_f64_dev": [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _i64],
    "pvs_cosine_topk_f64_dev": [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _int, _vp, _vp],
    "pvs_topk_merge_dev": [_vp, _vp, _vp, _int, _i64, _int, _vp, _vp],
    "pvs_materialise_dev": [_vp, _vp, _int, _int, _i64, _vp],
    "pvs_kmeans_step_dev": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "pvs_gmm_em_step_dev": [_vp, _vp, _vp, _i64, _vp],
    "pvs_label_sums_dev": [_vp, _vp, _int, _i64, _vp, _int, _int, _vp],
    "pvs_gram_dev": [_vp, _vp, _int, _i64, _vp],
    "pvs_seed_distances_dev": [_vp, _vp, _int, _i64, _vp, _int, _vp, _vp, _vp, _int],
    "pvs_seed_pick_dev": [_vp, _vp, _int, _i64, _vp, _vp, _vp, _vp, _int, _vp, _vp],
    "pvs_min_update_dev": [_vp, _vp, _vp, _i64, _vp],
    "pvs_kmeanspp_run_dev": [_vp, _vp, _int, _i64, _int, _int, _vp, _vp],
    "pvs_fused_profile": [_vp, _int, _vp],
    "pvs_timers_enable": [_vp, _int],
    "pvs_timers_reset": [_vp],
    "pvs_timers_read": [_vp, _int, C.POINTER(C.c_double), C.POINTER(C.c_int64)],
}

_lib = None
_lock = threading.Lock()


def mapped_hip_runtimes() -> list[str]:
    """Paths of the HIP runtime images (libamdhip64) mapped into this process."""
    found = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1] if "/" in line else ""
                if "libamdhip64" in os.path.basename(path) and path not in found:
                    found.append(path)
    except OSError:
        pass
    return found


def _one_hip_runtime() -> None:
    """One HIP runtime per process, whatever the import order.

    libpvsim_hip.so needs `libamdhip64.so.7`.  A PyTorch-ROCm wheel ships its own copy under torch/lib with that same
    SONAME, and two runtimes in one process each open the device on their own (the second one then sees no GPU).  The
    dynamic linker binds a DT_NEEDED entry to an already mapped object of that SONAME, so all that is needed is that the
    FIRST runtime mapped is the one everybody else will ask for: if none is mapped yet and torch is installed, map
    torch's copy now (a later `import torch` finds the same file); without torch the system runtime under /opt/rocm is
    used.  pvs_init refuses to run when it finds two runtimes mapped."""
    if mapped_hip_runtimes():
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for loc in (spec.submodule_search_locations if spec and spec.submodule_search_locations else []):
        cand = os.path.join(loc, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            return


def lib():
    """The loaded shared library (loads on first use; raises ImportError if it is not built)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise ImportError(
                        f"{LIB_PATH} not found: build the HIP library first "
                        "(python -c 'import __graft_entry__ as g; g.build()' or make -C python-visual-similarity_amd/csrc). "
                        "pvsim has no CPU fallback.")
                _one_hip_runtime()
                l = C.CDLL(LIB_PATH)
                for name, args in SIGNATURES.items():
                    fn = getattr(l, name)          # AttributeError here = header / library mismatch
                    fn.argtypes = args
                    fn.restype = C.c_int
                l.pvs_last_error.restype = C.c_char_p
                l.pvs_comm_library.restype = C.c_char_p
                l.pvs_stream.restype = C.c_void_p
                _lib = l
    return _lib


def check(status: int) -> None:
    if status != PVS_OK:
        msg = lib().pvs_last_error().decode("utf-8", "replace")
        raise _EXC.get(status, RuntimeError)(msg)


def ptr(a) -> C.c_void_p:
    """Host pointer of a C-contiguous ndarray, raw int device pointer, or None."""
    if a is None:
        return C.c_void_p(None)
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(int(a))


def norm_params(power, norm_order, epsilon) -> NormParams:
    return NormParams(float(power), float(norm_order), float(epsilon))
