"""ctypes binding of libvisfd_hip.so (include/visfd_hip.h) for Python hosts.

Two faces, like the C ABI:
  * `Context.<op>(numpy arrays)`      host face  (drop-in: synchronous, copies in and out)
  * `Context.<op>_dev(torch tensors)` device face (asynchronous on the context's stream; tensors
                                      live in HBM; multi-channel fields are channel-planar:
                                      dir (3,nz,ny,nx), tensor (6,nz,ny,nx))

There is NO CPU fallback: loading fails loudly when the HIP library is missing, and creating a
context fails when no GPU is present.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VISFD_HIP_LIB") or os.path.join(_HERE, "libvisfd_hip.so")  # env: tuning variants (tools/build_variant.py)

INCREASING_EIVALS = 0
DECREASING_EIVALS = 1

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_i64 = C.c_int64
_vp = C.c_void_p


class VisfdHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("visfd_hip error %d: %s" % (code, msg))
        self.code = code


class Blob(C.Structure):
    _fields_ = [("ix", C.c_int32), ("iy", C.c_int32), ("iz", C.c_int32), ("scale", C.c_int32),
                ("sigma", C.c_float), ("score", C.c_float)]


_VOL = [_vp, _vp, _vp, _i64, _i64, _i64]  # src, dst, mask, nx, ny, nz  (pointers as void*)

_SIGS = {
    "visfd_hip_create": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "visfd_hip_destroy": (C.c_int, [_vp]),
    "visfd_hip_synchronize": (C.c_int, [_vp]),
    "visfd_hip_trim": (C.c_int, [_vp]),
    "visfd_hip_last_error": (C.c_char_p, []),
    "visfd_hip_abi_version": (C.c_int, []),
    "visfd_hip_workspace_bytes": (_i64, [_vp]),
    "visfd_hip_set_option": (C.c_int, [_vp, C.c_char_p, _i64]),
    "visfd_hip_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "visfd_hip_gauss_taps": (C.c_int, [C.c_float, C.c_int, _fp]),
    "visfd_hip_ratio_from_threshold": (C.c_float, [C.c_float]),
    "visfd_hip_local_fluctuations": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _fp, C.c_float, C.c_float, C.c_int]),
    "visfd_hip_local_fluctuations_dev": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _fp, C.c_float, C.c_float, C.c_int]),
    "visfd_hip_fluctuation_sigmas": (C.c_int, [_fp, C.c_float, C.c_float, C.c_float, _fp, C.POINTER(C.c_float)]),
    "visfd_hip_gauss_halfwidths": (C.c_int, [_fp, C.c_float, _ip]),
    "visfd_hip_separable3d": (C.c_int, [_vp] + _VOL + [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, _fp]),
    "visfd_hip_separable3d_dev": (C.c_int, [_vp] + _VOL + [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, _fp]),
    "visfd_hip_apply_gauss": (C.c_int, [_vp] + _VOL + [_fp, _ip, C.c_int, _fp]),
    "visfd_hip_apply_gauss_dev": (C.c_int, [_vp] + _VOL + [_fp, _ip, C.c_int, _fp]),
    "visfd_hip_apply_gauss_slab_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _fp, _ip, C.c_int, _fp]),
    "visfd_hip_apply_dog": (C.c_int, [_vp] + _VOL + [_fp, _fp, _ip, _fp, _fp]),
    "visfd_hip_apply_dog_dev": (C.c_int, [_vp] + _VOL + [_fp, _fp, _ip, _fp, _fp]),
    "visfd_hip_apply_log": (C.c_int, [_vp] + _VOL + [_fp, C.c_float, C.c_float, _fp, _fp]),
    "visfd_hip_apply_log_dev": (C.c_int, [_vp] + _VOL + [_fp, C.c_float, C.c_float, _fp, _fp]),
    "visfd_hip_blob_dog": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _fp, C.c_int, _fp, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_int, C.POINTER(Blob), _i64, C.POINTER(_i64),
                                     C.POINTER(Blob), _i64, C.POINTER(_i64)]),
    "visfd_hip_blob_dog_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _fp, C.c_int, _fp, C.c_float, C.c_float,
                                         C.c_float, C.c_float, C.c_int, C.POINTER(Blob), _i64, C.POINTER(_i64),
                                         C.POINTER(Blob), _i64, C.POINTER(_i64)]),
    "visfd_hip_blob_dog_begin_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _fp, C.c_int, _fp, C.c_float, C.c_float,
                                               C.c_float, C.c_float, C.c_int, C.POINTER(_vp)]),
    "visfd_hip_blob_dog_end": (C.c_int, [_vp, C.POINTER(Blob), _i64, C.POINTER(_i64), C.POINTER(Blob), _i64, C.POINTER(_i64)]),
    "visfd_hip_blob_dog_abort": (None, [_vp]),
    "visfd_hip_blob_diameters_to_sigmas": (C.c_int, [_fp, C.c_int, _fp]),
    "visfd_hip_blob_sigmas_to_diameters": (C.c_int, [_fp, C.c_int, _fp]),
    "visfd_hip_calc_hessian": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, C.c_float, C.c_float]),
    "visfd_hip_calc_hessian_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, C.c_float, C.c_float]),
    "visfd_hip_diagonalize_flat_sym3": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int]),
    "visfd_hip_diagonalize_flat_sym3_dev": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int]),
    "visfd_hip_hessian_saliency": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, _vp, _vp]),
    "visfd_hip_hessian_saliency_dev": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, _vp, _vp]),
    "visfd_hip_ridge_saliency_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_float, C.c_float, C.c_int,
                                               _vp, _vp]),
    "visfd_hip_ridge_scores_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_float, C.c_float, C.c_int, _vp, _vp]),
    "visfd_hip_ridge_directions_dev": (C.c_int, [_vp, _vp, _i64, _i64, _i64, C.c_float, C.c_int, _vp, _vp]),
    "visfd_hip_threshold_fraction": (C.c_int, [_vp, _vp, _vp, _i64, C.c_float, _fp]),
    "visfd_hip_threshold_fraction_dev": (C.c_int, [_vp, _vp, _vp, _i64, C.c_float, _fp]),
    "visfd_hip_select_histogram_dev": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_uint32,
                                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "visfd_hip_select_histogram_todev": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, C.c_uint32, _vp]),
    "visfd_hip_get_stream": (_vp, [_vp]),
    "visfd_hip_apply_threshold_dev": (C.c_int, [_vp, _vp, _i64, C.c_float]),
    "visfd_hip_tv_dense_stick": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, C.c_float, C.c_int,
                                           C.c_float, C.c_int]),
    "visfd_hip_tv_dense_stick_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, C.c_float, C.c_int,
                                               C.c_float, C.c_int]),
    "visfd_hip_tv_dense_stick_slab_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                                    C.c_float, C.c_int, C.c_float, C.c_int]),
    "visfd_hip_tv_tables": (C.c_int, [C.c_float, C.c_float, _ip, _fp, _fp]),
    "visfd_hip_tv_weight_sum": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, C.c_float, C.c_float]),
    "visfd_hip_membrane_detect": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_float, C.c_float, C.c_int, C.c_float,
                                            C.c_float, C.c_float, C.c_int, C.c_float, _vp, _vp, _vp, _fp]),
    "visfd_hip_membrane_detect_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_float, C.c_float, C.c_int,
                                                C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, _vp, _vp, _vp,
                                                _fp]),
    "visfd_hip_tensor_saliency": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, _vp]),
    "visfd_hip_tensor_saliency_dev": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, _vp]),
    # the peak-height factor (`-membrane-background`)
    "visfd_hip_peak_background_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_float, C.c_float, C.c_int, _vp]),
    "visfd_hip_ridge_scores_bg_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_float, C.c_float, C.c_int, _vp, _vp, _vp]),
    "visfd_hip_tensor_saliency_bg_dev": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, _vp, _vp, _vp]),
    "visfd_hip_membrane_detect_bg": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_float, C.c_float, C.c_int, C.c_float,
                                               C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int, _vp, _vp, _vp, _fp]),
    "visfd_hip_membrane_detect_bg_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_float, C.c_float, C.c_int,
                                                   C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int,
                                                   _vp, _vp, _vp, _fp]),
    "visfd_hip_membrane_detect_slab_bg_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, C.c_float, C.c_float, C.c_int,
                                                        C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, _fp]),
    "visfd_hip_membrane_detect_slab_bg": (C.c_int, [_vp, _vp, _i64, _i64, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float,
                                                    C.c_int, C.c_float, C.c_float, C.c_int, _vp, _vp, _fp]),
    "visfd_hip_bin_array3d": (C.c_int, [_vp, _vp, C.POINTER(_i64), _vp, C.POINTER(_i64), _ip]),
    "visfd_hip_bin_array3d_dev": (C.c_int, [_vp, _vp, C.POINTER(_i64), _vp, C.POINTER(_i64), _ip]),
    "visfd_hip_unbin_array3d": (C.c_int, [_vp, _vp, C.POINTER(_i64), _vp, C.POINTER(_i64), _ip]),
    "visfd_hip_unbin_array3d_dev": (C.c_int, [_vp, _vp, C.POINTER(_i64), _vp, C.POINTER(_i64), _ip]),
    "visfd_hip_label_connected": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_float, _vp, C.c_float, C.c_float, C.c_int,
                                            _vp, C.c_float, C.c_float, C.c_int, C.c_int, _i64, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(_i64), _vp, _vp, _vp, _i64]),
    "visfd_hip_label_connected_ex": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, C.c_float, _vp, C.c_float, C.c_float, C.c_int,
                                               _vp, C.c_float, C.c_float, C.c_int, C.c_int, _i64, C.c_int, C.c_int, C.c_int,
                                               C.POINTER(_i64), _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "visfd_hip_principal_directions_host": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp]),
    "visfd_hip_tensor_saliency_host": (C.c_int, [_vp, _vp, _i64, C.c_int, _vp]),
    "visfd_hip_diagonalize_sym3_f32_host": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "visfd_hip_diagonalize_flat_sym3_host": (C.c_int, [_vp, _vp, _i64, C.c_int]),
    "visfd_hip_convert_flat_sym2_evects3_host": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "visfd_hip_surface_points": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, C.c_int, _fp, C.c_float, C.c_int, C.c_float,
                                           _vp, _vp, _i64, C.POINTER(_i64)]),
    "visfd_hip_sphere_overlap": (C.c_float, [C.c_float, C.c_float, C.c_float]),
    "visfd_hip_sort_blobs": (C.c_int, [_fp, _fp, _fp, _i64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "visfd_hip_discard_masked_blobs": (C.c_int, [_fp, _fp, _fp, C.POINTER(_i64), _vp, _i64, _i64, _i64]),
    "visfd_hip_discard_overlapping_blobs": (C.c_int, [_fp, _fp, _fp, C.POINTER(_i64), C.c_float, C.c_float,
                                                     C.c_float, C.c_int, C.c_int]),
    # Z-slab runs (csrc/slab.hip)
    "visfd_hip_slab_unique_id": (C.c_int, [_vp]),
    "visfd_hip_slab_create_rccl": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _i64, C.c_int, C.POINTER(_vp)]),
    "visfd_hip_slab_create_custom": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _i64, C.c_int, C.POINTER(_vp)]),
    "visfd_hip_slab_destroy": (C.c_int, [_vp]),
    "visfd_hip_slab_layout": (C.c_int, [_vp, C.POINTER(_i64)]),
    "visfd_hip_slab_set_reserve": (C.c_int, [_vp, C.c_int]),
    "visfd_hip_slab_rccl_available": (C.c_int, []),
    "visfd_hip_apply_gauss_slab": (C.c_int, [_vp, _vp, _i64, _i64, _fp, _ip, C.c_int, _vp, _fp]),
    "visfd_hip_blob_dog_slab": (C.c_int, [_vp, _vp, _i64, _i64, _fp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                          _vp, _i64, C.POINTER(_i64), _vp, _i64, C.POINTER(_i64)]),
    "visfd_hip_slab_selftest": (C.c_int, [_vp, _i64]),
    "visfd_hip_slab_exchange_dev": (C.c_int, [_vp, C.POINTER(_vp), C.c_int, _i64, _i64, C.c_int]),
    "visfd_hip_membrane_detect_slab_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, C.c_float, C.c_float, C.c_int,
                                                     C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, _fp]),
    "visfd_hip_membrane_detect_slab": (C.c_int, [_vp, _vp, _i64, _i64, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float,
                                                 C.c_int, C.c_float, _vp, _vp, _fp]),
    "visfd_hip_blob_dog_slab_dev": (C.c_int, [_vp, _vp, _i64, _i64, _fp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                              C.c_int, _vp, _i64, C.POINTER(_i64), _vp, _i64, C.POINTER(_i64)]),
}

_SENDRECV = C.CFUNCTYPE(C.c_int, _vp, C.c_int, _vp, _vp, C.c_size_t, _vp)
_ALLREDUCE = C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_size_t, _vp)
_GROUP = C.CFUNCTYPE(C.c_int, _vp)


class Transport(C.Structure):   # visfd_hip_transport
    _fields_ = [("sendrecv", _SENDRECV), ("allreduce_sum_u64", _ALLREDUCE), ("group_start", _GROUP), ("group_end", _GROUP),
                ("user", _vp)]

_lib = None


def load_library():
    """Load libvisfd_hip.so; raises if it has not been built (python -m visfd_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -m visfd_amd.build` "
                              "(there is no CPU fallback)" % LIB_PATH)
        # PyTorch-ROCm ships its own copy of the HIP runtime; if this library pulled in the system copy first, a
        # later `import torch` in the same process would bring a second runtime that cannot see the GPU ("no
        # ROCm-capable device").  Loading torch's first makes both share one runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def exported_symbols():
    return sorted(_SIGS)


def _np(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], "need C float32"
    return a.ctypes.data


def _dev(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous() and str(t.dtype) == "torch.float32", "need contiguous cuda float32"
    return t.data_ptr()


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


def gauss_taps(sigma, h):
    out = np.empty(2 * h + 1, np.float32)
    rc = load_library().visfd_hip_gauss_taps(float(sigma), int(h), out.ctypes.data_as(_fp))
    if rc:
        raise VisfdHipError(rc, load_library().visfd_hip_last_error().decode())
    return out


def ratio_from_threshold(thr):
    return float(load_library().visfd_hip_ratio_from_threshold(float(thr)))


def gauss_halfwidths(sigma, ratio):
    hw = (C.c_int * 3)()
    load_library().visfd_hip_gauss_halfwidths(_f3(sigma), float(ratio), hw)
    return tuple(hw)


def fluctuation_sigmas(radius, exponent=2.0, truncate_ratio=-1.0, truncate_threshold=0.03):
    """LocalFluctuationsByRadius' parameter arithmetic (filter3d.hpp:1908-1914, filter3d_variants.hpp:663-669):
    -> (sigma[3], ratio)."""
    sg = (C.c_float * 3)()
    r = C.c_float()
    load_library().visfd_hip_fluctuation_sigmas(_f3(radius), float(exponent), float(truncate_ratio),
                                                float(truncate_threshold), sg, C.byref(r))
    return tuple(sg), r.value


def tv_tables(sigma_tv, cutoff):
    L = load_library()
    h = C.c_int()
    L.visfd_hip_tv_tables(float(sigma_tv), float(cutoff), C.byref(h), None, None)
    n = 2 * h.value + 1
    w = np.empty((n, n, n), np.float32)
    r = np.empty((n, n, n, 3), np.float32)
    L.visfd_hip_tv_tables(float(sigma_tv), float(cutoff), C.byref(h), w.ctypes.data_as(_fp), r.ctypes.data_as(_fp))
    return h.value, w, r


def diameters_to_sigmas(d):
    d = np.ascontiguousarray(d, np.float32)
    s = np.empty_like(d)
    load_library().visfd_hip_blob_diameters_to_sigmas(d.ctypes.data_as(_fp), len(d), s.ctypes.data_as(_fp))
    return s


def sigmas_to_diameters(s):
    s = np.ascontiguousarray(s, np.float32)
    d = np.empty_like(s)
    load_library().visfd_hip_blob_sigmas_to_diameters(s.ctypes.data_as(_fp), len(s), d.ctypes.data_as(_fp))
    return d


# ---- voxel clustering (host-side) ---------------------------------------------------------
def label_connected(saliency, threshold_saliency, mask=None, direction=None, tensor=None,
                    threshold_vector_saliency=-np.inf, threshold_vector_neighbor=-np.inf, consider_dot_product_sign=True,
                    threshold_tensor_saliency=-np.inf, threshold_tensor_neighbor=-np.inf,
                    tensor_is_positive_definite_near_target=True, connectivity=1, label_undefined=-1, sort_by_size=True,
                    standardize_directions=False, start_from_saliency_maxima=True, voxel_weights=None, must_link=None,
                    must_link_directions=None):
    """LabelConnected (connect.hpp:168-1427).  saliency [nz,ny,nx]; direction [nz,ny,nx,3] (rewritten in place when
    standardize_directions); tensor [nz,ny,nx,6].  Returns (labels int64 [nz,ny,nx], n_clusters, seed positions
    [n,3] in final order, sizes [n] and seed saliencies [n] in provisional order)."""
    L = load_library()
    nz, ny, nx = saliency.shape
    labels = np.empty((nz, ny, nx), np.int64)
    n = _i64(0)
    cap = int(saliency.size)
    cm, cs, csal = np.zeros((cap, 3), np.float32), np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    for a in (direction, tensor):
        assert a is None or (a.dtype == np.float32 and a.flags["C_CONTIGUOUS"])
    # must_link: list of groups, each a list of (x, y, z) locations in voxels; must_link_directions: the same shape of
    # 0 / 1 / 2 (same / opposite / automatic, connect.hpp DirectionPairType) or None
    ml_c = ml_n = ml_d = None
    ngroups = 0
    if must_link:
        ngroups = len(must_link)
        ml_c = np.ascontiguousarray(np.concatenate([np.asarray(g_, np.float32).reshape(-1, 3) for g_ in must_link], 0))
        ml_n = np.array([len(g_) for g_ in must_link], np.int64)
        if must_link_directions is not None:
            ml_d = np.ascontiguousarray(np.concatenate([np.asarray(d_, np.int32).ravel() for d_ in must_link_directions]))
    ptr = lambda a: None if a is None else a.ctypes.data
    _chk_host(L, L.visfd_hip_label_connected_ex(
        _np(saliency), labels.ctypes.data, _np(mask), nx, ny, nz, threshold_saliency,
        None if direction is None else direction.ctypes.data, threshold_vector_saliency, threshold_vector_neighbor,
        int(bool(consider_dot_product_sign)), None if tensor is None else tensor.ctypes.data, threshold_tensor_saliency,
        threshold_tensor_neighbor, int(bool(tensor_is_positive_definite_near_target)), int(connectivity),
        int(label_undefined), int(bool(sort_by_size)), int(bool(standardize_directions)),
        int(bool(start_from_saliency_maxima)), C.byref(n), cm.ctypes.data, cs.ctypes.data, csal.ctypes.data, cap,
        _np(voxel_weights), ptr(ml_c), ptr(ml_n), ngroups, ptr(ml_d)))
    k = n.value
    return labels, k, cm[:k], cs[:k], csal[:k]


def tensor_saliency_host(tensor, order, sal_inout, mask=None):
    """lambda0 - lambda1 of every [.., 6] tensor, host arithmetic (handlers.cpp:1868-1888), into sal_inout."""
    L = load_library()
    _chk_host(L, L.visfd_hip_tensor_saliency_host(_np(tensor), _np(mask), int(tensor.size // 6), int(order),
                                                  _np(sal_inout)))
    return sal_inout


def diagonalize_sym3_f32_host(m, order):
    """DiagonalizeSym3<float> (eigen3_simple.hpp:137-266) of m [..., 3, 3] -> (eivals [..., 3], eivects [..., 3, 3] rows)."""
    L = load_library()
    m = np.ascontiguousarray(m, np.float32)
    vals = np.empty(m.shape[:-2] + (3,), np.float32)
    vecs = np.empty(m.shape, np.float32)
    mf, vf, ef = m.reshape(-1, 9), vals.reshape(-1, 3), vecs.reshape(-1, 9)
    for i in range(len(mf)):
        _chk_host(L, L.visfd_hip_diagonalize_sym3_f32_host(mf[i].ctypes.data, int(order), vf[i].ctypes.data, ef[i].ctypes.data))
    return vals, vecs


def diagonalize_flat_sym3_host(m6, order):
    """DiagonalizeFlatSym3 (eigen3_simple.hpp:271-342) of [..., 6] flat matrices on the host -> [..., 6]."""
    L = load_library()
    m6 = np.ascontiguousarray(m6, np.float32)
    out = np.empty_like(m6)
    _chk_host(L, L.visfd_hip_diagonalize_flat_sym3_host(m6.ctypes.data, out.ctypes.data, int(m6.size // 6), int(order)))
    return out


def convert_flat_sym2_evects3_host(m6, order):
    """ConvertFlatSym2Evects3<float> of [..., 6] flat matrices -> (eivals [..., 3], eivects [..., 3, 3] rows)."""
    L = load_library()
    m6 = np.ascontiguousarray(m6, np.float32)
    vals = np.empty(m6.shape[:-1] + (3,), np.float32)
    vecs = np.empty(m6.shape[:-1] + (3, 3), np.float32)
    mf, vf, ef = m6.reshape(-1, 6), vals.reshape(-1, 3), vecs.reshape(-1, 9)
    for i in range(len(mf)):
        _chk_host(L, L.visfd_hip_convert_flat_sym2_evects3_host(mf[i].ctypes.data, int(order), vf[i].ctypes.data,
                                                               ef[i].ctypes.data))
    return vals, vecs


def principal_directions_host(tensor, order, mask=None):
    """Eigenvector row 0 of every [.., 6] tensor, host arithmetic (handlers.cpp:1935-1952) -> [.., 3]."""
    L = load_library()
    out = np.zeros(tensor.shape[:-1] + (3,), np.float32)
    _chk_host(L, L.visfd_hip_principal_directions_host(_np(tensor), _np(mask), int(tensor.size // 6), int(order),
                                                       out.ctypes.data))
    return out


# ---- blob list post-processing (host-side).  Blob lists are (crds[n,3], diameters[n], scores[n]).
DO_NOT_SORT, SORT_DECREASING, SORT_INCREASING, SORT_DECREASING_MAGNITUDE, SORT_INCREASING_MAGNITUDE = range(5)


def _chk_host(L, rc):
    if rc:
        raise VisfdHipError(rc, L.visfd_hip_last_error().decode())


def _blob_arrays(crds, diameters, scores):
    c = np.ascontiguousarray(crds, np.float32).reshape(-1, 3).copy()
    d = np.ascontiguousarray(diameters, np.float32).copy()
    s = np.ascontiguousarray(scores, np.float32).copy()
    assert len(c) == len(d) == len(s), "blob lists differ in length"
    return c, d, s


def _fptr(a):
    return a.ctypes.data_as(_fp)


def sphere_overlap(rij, ri, rj):
    """CalcSphereOverlap (visfd_utils.hpp:95-118)."""
    return float(load_library().visfd_hip_sphere_overlap(rij, ri, rj))


def sort_blobs(crds, diameters, scores, criteria=SORT_DECREASING_MAGNITUDE, ascending=True):
    """SortBlobs (feature.hpp:573-616): returns (crds, diameters, scores, permutation)."""
    L = load_library()
    c, d, s = _blob_arrays(crds, diameters, scores)
    perm = np.arange(len(d), dtype=np.uint64)
    _chk_host(L, L.visfd_hip_sort_blobs(_fptr(c), _fptr(d), _fptr(s), len(d), int(criteria), int(bool(ascending)),
                                        perm.ctypes.data_as(C.POINTER(C.c_uint64))))
    return c, d, s, perm


def discard_masked_blobs(crds, diameters, scores, mask):
    """DiscardMaskedBlobs (feature.hpp:924-969); mask: [nz, ny, nx] float32 or None."""
    L = load_library()
    c, d, s = _blob_arrays(crds, diameters, scores)
    n = _i64(len(d))
    if mask is None:
        return c, d, s
    nz, ny, nx = mask.shape
    _chk_host(L, L.visfd_hip_discard_masked_blobs(_fptr(c), _fptr(d), _fptr(s), C.byref(n), _np(mask), nx, ny, nz))
    return c[:n.value], d[:n.value], s[:n.value]


def discard_overlapping_blobs(crds, diameters, scores, min_radial_separation_ratio, max_volume_overlap_large=np.inf,
                              max_volume_overlap_small=np.inf, criteria=SORT_DECREASING_MAGNITUDE, scale=6):
    """DiscardOverlappingBlobs (feature.hpp:720-913): greedy non-max suppression, best blobs first."""
    L = load_library()
    c, d, s = _blob_arrays(crds, diameters, scores)
    n = _i64(len(d))
    _chk_host(L, L.visfd_hip_discard_overlapping_blobs(_fptr(c), _fptr(d), _fptr(s), C.byref(n),
                                                       min_radial_separation_ratio, max_volume_overlap_large,
                                                       max_volume_overlap_small, int(criteria), int(scale)))
    return c[:n.value], d[:n.value], s[:n.value]


_BLOB_DTYPE = np.dtype([("ix", "<i4"), ("iy", "<i4"), ("iz", "<i4"), ("scale", "<i4"), ("sigma", "<f4"),
                        ("score", "<f4")])


def _blobs_to_rows(arr, n):
    """-> float32 rows x,y,z,sigma,score (the reference's list layout) plus the scale indices."""
    rec = np.frombuffer(arr, dtype=_BLOB_DTYPE, count=n)
    rows = np.empty((n, 5), np.float32)
    rows[:, 0] = rec["ix"]
    rows[:, 1] = rec["iy"]
    rows[:, 2] = rec["iz"]
    rows[:, 3] = rec["sigma"]
    rows[:, 4] = rec["score"]
    return rows, rec["scale"].copy()


class Context:
    """One GPU, one HIP stream, one workspace (visfd_hip_ctx)."""

    def __init__(self, device=0, stream=None):
        self._L = load_library()
        h = _vp()
        rc = self._L.visfd_hip_create(int(device), _vp(stream) if stream else None, C.byref(h))
        if rc:
            raise VisfdHipError(rc, self._L.visfd_hip_last_error().decode())
        self._h = h

    def _chk(self, rc):
        if rc:
            raise VisfdHipError(rc, self._L.visfd_hip_last_error().decode())

    def close(self):
        # slab handles hold a pointer to this context: they go first
        for ref in getattr(self, "_slabs", []):
            s = ref()
            if s is not None:
                s.close()
        self._slabs = []
        if getattr(self, "_h", None):
            self._L.visfd_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._chk(self._L.visfd_hip_synchronize(self._h))

    def trim(self):
        self._chk(self._L.visfd_hip_trim(self._h))

    def workspace_bytes(self):
        return int(self._L.visfd_hip_workspace_bytes(self._h))

    def set_option(self, name, value):
        """Tuning / test switch of this context (include/visfd_hip.h: visfd_hip_set_option)."""
        self._chk(self._L.visfd_hip_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int64()
        self._chk(self._L.visfd_hip_get_option(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def options(self, **kw):
        """Context manager: set options for the duration of a with-block, then restore the values they had before it
        (whatever their origin: the environment, an earlier set_option, an enclosing scope)."""
        ctx = self

        class _Scope:
            def __enter__(self_):
                self_.saved = {k: ctx.get_option(k) for k in kw}
                for k, v in kw.items():
                    ctx.set_option(k, v)
                return ctx

            def __exit__(self_, *a):
                for k, v in self_.saved.items():
                    ctx.set_option(k, v)
                return False
        return _Scope()

    # ---------------------------------------------------------------- host face (numpy)
    def separable3d(self, src, taps_xyz, mask=None, normalize=True):
        nz, ny, nx = src.shape
        dst = np.empty_like(src)
        A = C.c_float()
        t = [np.ascontiguousarray(x, np.float32) for x in taps_xyz]
        h = [(len(x) - 1) // 2 for x in t]
        self._chk(self._L.visfd_hip_separable3d(self._h, _np(src), _np(dst), _np(mask), nx, ny, nz,
                                                t[0].ctypes.data_as(_fp), h[0], t[1].ctypes.data_as(_fp), h[1],
                                                t[2].ctypes.data_as(_fp), h[2], int(normalize), C.byref(A)))
        return dst, A.value

    def gauss_hw(self, src, sigma, hw, mask=None, normalize=True):
        nz, ny, nx = src.shape
        dst = np.empty_like(src)
        A = C.c_float()
        self._chk(self._L.visfd_hip_apply_gauss(self._h, _np(src), _np(dst), _np(mask), nx, ny, nz, _f3(sigma),
                                                _i3(hw), int(normalize), C.byref(A)))
        return dst, A.value

    def gauss_ratio(self, src, sigma, ratio, mask=None, normalize=True):
        return self.gauss_hw(src, sigma, gauss_halfwidths(sigma, ratio), mask, normalize)

    def local_fluctuations(self, src, sigma, ratio, mask=None, normalize=True, exponent=2.0):
        """LocalFluctuations (filter3d.hpp:1698-1853), Gaussian weights."""
        nz, ny, nx = src.shape
        dst = np.empty_like(src)
        self._chk(self._L.visfd_hip_local_fluctuations(self._h, _np(src), _np(dst), _np(mask), nx, ny, nz, _f3(sigma),
                                                       float(exponent), float(ratio), int(normalize)))
        return dst

    def dog(self, src, sigma_a, sigma_b, hw, mask=None):
        nz, ny, nx = src.shape
        dst = np.empty_like(src)
        A, B = C.c_float(), C.c_float()
        self._chk(self._L.visfd_hip_apply_dog(self._h, _np(src), _np(dst), _np(mask), nx, ny, nz, _f3(sigma_a),
                                              _f3(sigma_b), _i3(hw), C.byref(A), C.byref(B)))
        return dst, A.value, B.value

    def log(self, src, sigma, delta, ratio, mask=None):
        nz, ny, nx = src.shape
        dst = np.empty_like(src)
        A, B = C.c_float(), C.c_float()
        self._chk(self._L.visfd_hip_apply_log(self._h, _np(src), _np(dst), _np(mask), nx, ny, nz, _f3(sigma),
                                              float(delta), float(ratio), C.byref(A), C.byref(B)))
        return dst, A.value, B.value

    def _blob_call(self, fn, psrc, pmask, shape, sigmas, aspect, delta, ratio, minima_threshold, maxima_threshold,
                   use_ratios, cap):
        nz, ny, nx = shape
        sig = np.ascontiguousarray(sigmas, np.float32)
        asp = _f3(aspect) if aspect is not None else None
        for attempt in (0, 1):
            # the record arrays are kept by the context between calls (a pipeline asks for millions of records of capacity:
            # mapping and unmapping 2 x 100 MB per call cost 1-25 ms of host time, depending on the state of the machine)
            bufs = getattr(self, "_blob_bufs", None)
            if bufs is None or len(bufs[0]) < cap:
                bufs = self._blob_bufs = (np.empty(cap, _BLOB_DTYPE), np.empty(cap, _BLOB_DTYPE))   # visfd_hip_blob records
            amin, amax = bufs   # (at least `cap` records; the call is still told `cap`)
            nmin, nmax = _i64(), _i64()
            rc = fn(self._h, psrc, pmask, nx, ny, nz, sig.ctypes.data_as(_fp), len(sig), asp, float(delta),
                    float(ratio), float(minima_threshold), float(maxima_threshold), int(use_ratios),
                    amin.ctypes.data_as(C.POINTER(Blob)), cap, C.byref(nmin),
                    amax.ctypes.data_as(C.POINTER(Blob)), cap, C.byref(nmax))
            if rc == 4 and attempt == 0:   # VISFD_HIP_ECAPACITY: the exact counts came back; once more with room for them
                cap = max(nmin.value, nmax.value, 1)
                continue
            self._chk(rc)
            break
        return _blobs_to_rows(amin, nmin.value)[0], _blobs_to_rows(amax, nmax.value)[0]

    def blob_dog_begin_dev(self, src, sigmas, mask=None, aspect=None, delta=0.02, ratio=2.5, minima_threshold=np.inf,
                           maxima_threshold=-np.inf, use_ratios=False):
        """First half of blob_dog_dev (visfd_hip_blob_dog_begin_dev): everything queued, most lists fetched; returns a job for
        blob_dog_end().  Other calls of this context may be queued in between; src and mask must stay as they are until then."""
        nz, ny, nx = src.shape
        sig = np.ascontiguousarray(sigmas, np.float32)
        asp = _f3(aspect) if aspect is not None else None
        job = _vp()
        self._chk(self._L.visfd_hip_blob_dog_begin_dev(self._h, _dev(src), _dev(mask), nx, ny, nz, sig.ctypes.data_as(_fp), len(sig),
                                                       asp, float(delta), float(ratio), float(minima_threshold),
                                                       float(maxima_threshold), int(use_ratios), C.byref(job)))
        return [job, src, mask, sig]   # (the tensors and the sigma array live as long as the job)

    def blob_dog_end(self, job, cap=1 << 16):
        """Second half: -> (minima, maxima) rows x,y,z,sigma,score, as blob_dog_dev returns them."""
        h = job[0]
        if h is None:
            raise ValueError("blob job already finished")
        for attempt in (0, 1):
            bufs = getattr(self, "_blob_bufs", None)
            if bufs is None or len(bufs[0]) < cap:
                bufs = self._blob_bufs = (np.empty(cap, _BLOB_DTYPE), np.empty(cap, _BLOB_DTYPE))
            amin, amax = bufs
            nmin, nmax = _i64(), _i64()
            rc = self._L.visfd_hip_blob_dog_end(h, amin.ctypes.data_as(C.POINTER(Blob)), cap, C.byref(nmin),
                                                amax.ctypes.data_as(C.POINTER(Blob)), cap, C.byref(nmax))
            if rc == 4 and attempt == 0:   # VISFD_HIP_ECAPACITY: the job is still alive, the counts came back
                cap = max(nmin.value, nmax.value, 1)
                continue
            job[0] = None                   # every other outcome has freed the job
            if rc == 4:
                self._L.visfd_hip_blob_dog_abort(h)
            self._chk(rc)
            break
        return _blobs_to_rows(amin, nmin.value)[0], _blobs_to_rows(amax, nmax.value)[0]

    def blob_dog_abort(self, job):
        if job[0] is not None:
            self._L.visfd_hip_blob_dog_abort(job[0])
            job[0] = None

    def blob_dog(self, src, sigmas, mask=None, aspect=None, delta=0.02, ratio=2.5, minima_threshold=np.inf,
                 maxima_threshold=-np.inf, use_ratios=False, cap=1 << 16):
        return self._blob_call(self._L.visfd_hip_blob_dog, _np(src), _np(mask), src.shape, sigmas, aspect, delta,
                               ratio, minima_threshold, maxima_threshold, use_ratios, cap)

    def calc_hessian(self, src, sigma, ratio, mask=None, want_grad=True):
        nz, ny, nx = src.shape
        hess = np.zeros((nz, ny, nx, 6), np.float32)
        grad = np.zeros((nz, ny, nx, 3), np.float32) if want_grad else None
        self._chk(self._L.visfd_hip_calc_hessian(self._h, _np(src), _np(grad), _np(hess), _np(mask), nx, ny, nz,
                                                 float(sigma), float(ratio)))
        return grad, hess

    def diagonalize(self, m6, order):
        m6 = np.ascontiguousarray(m6, np.float32)
        out = np.empty_like(m6)
        self._chk(self._L.visfd_hip_diagonalize_flat_sym3(self._h, _np(m6), _np(out), m6.size // 6, int(order)))
        return out

    def hessian_saliency(self, hess, order, mask=None):
        shp = hess.shape[:-1]
        sal = np.empty(shp, np.float32)
        dirs = np.zeros(shp + (3,), np.float32)
        self._chk(self._L.visfd_hip_hessian_saliency(self._h, _np(hess), _np(mask), sal.size, int(order), _np(sal),
                                                     _np(dirs)))
        return sal, dirs

    def threshold_fraction(self, sal, fraction, mask=None):
        thr = C.c_float()
        self._chk(self._L.visfd_hip_threshold_fraction(self._h, _np(sal), _np(mask), sal.size, float(fraction),
                                                       C.byref(thr)))
        return thr.value

    def tv_dense_stick(self, sal, dirs, sigma_tv, exponent=4, cutoff=2.0 ** 0.5, mask_src=None, mask_dst=None,
                       curves=False):
        nz, ny, nx = sal.shape
        tensor = np.zeros((nz, ny, nx, 6), np.float32)
        self._chk(self._L.visfd_hip_tv_dense_stick(self._h, _np(sal), _np(dirs), _np(tensor), _np(mask_src),
                                                   _np(mask_dst), nx, ny, nz, float(sigma_tv), int(exponent),
                                                   float(cutoff), int(curves)))
        return tensor

    def tv_weight_sum(self, sal, sigma_tv, cutoff=2.0 ** 0.5, mask_src=None, mask_dst=None):
        """The normalisation denominators of TVDenseStick(normalize=true) (feature.hpp:1761-1822); zeros where mask_dst == 0."""
        nz, ny, nx = sal.shape
        den = np.zeros_like(sal)
        self._chk(self._L.visfd_hip_tv_weight_sum(self._h, _np(sal), _np(den), _np(mask_src), _np(mask_dst), nx, ny, nz,
                                                  float(sigma_tv), float(cutoff)))
        return den

    def tensor_saliency(self, tensor, order, sal_inout, mask=None):
        self._chk(self._L.visfd_hip_tensor_saliency(self._h, _np(tensor), _np(mask), sal_inout.size, int(order),
                                                    _np(sal_inout)))
        return sal_inout

    # ---- binning (resample.hpp:53-166); numpy shapes are [nz, ny, nx] -------------------------
    @staticmethod
    def _sizes(shape):
        return (_i64 * 3)(int(shape[2]), int(shape[1]), int(shape[0]))

    def bin_array3d(self, src, dst_shape, offset=None):
        dst = np.empty(tuple(dst_shape), np.float32)
        self._chk(self._L.visfd_hip_bin_array3d(self._h, _np(src), self._sizes(src.shape), _np(dst),
                                                self._sizes(dst_shape), _i3(offset) if offset is not None else None))
        return dst

    def unbin_array3d(self, src, dst_shape, offset=None):
        dst = np.empty(tuple(dst_shape), np.float32)
        self._chk(self._L.visfd_hip_unbin_array3d(self._h, _np(src), self._sizes(src.shape), _np(dst),
                                                  self._sizes(dst_shape), _i3(offset) if offset is not None else None))
        return dst

    def bin_array3d_dev(self, src, dst, offset=None):
        self._chk(self._L.visfd_hip_bin_array3d_dev(self._h, _dev(src), self._sizes(src.shape), _dev(dst),
                                                    self._sizes(dst.shape), _i3(offset) if offset is not None else None))

    def unbin_array3d_dev(self, src, dst, offset=None):
        self._chk(self._L.visfd_hip_unbin_array3d_dev(self._h, _dev(src), self._sizes(src.shape), _dev(dst),
                                                      self._sizes(dst.shape), _i3(offset) if offset is not None else None))

    def membrane_detect(self, src, sigma, ratio, order, best_fraction=0.05, threshold_abs=0.0, sigma_tv=0.0,
                        tv_exponent=4, tv_cutoff=2.0 ** 0.5, mask=None, want_tensor=True, want_dir=False,
                        sigma_background=0.0, normalize_background=True):
        """HandleTV's compute section on host arrays -> (saliency, tensor or None, dirs or None, threshold).
        sigma_background > 0: `-membrane-background`, both scores times (image - background)."""
        nz, ny, nx = src.shape
        sal = np.empty_like(src)
        ten = np.zeros((nz, ny, nx, 6), np.float32) if want_tensor else None
        dirs = np.zeros((nz, ny, nx, 3), np.float32) if want_dir else None
        thr = C.c_float()
        self._chk(self._L.visfd_hip_membrane_detect_bg(self._h, _np(src), _np(mask), nx, ny, nz, float(sigma),
                                                       float(ratio), int(order), float(best_fraction),
                                                       float(threshold_abs), float(sigma_tv), int(tv_exponent),
                                                       float(tv_cutoff), float(sigma_background), int(bool(normalize_background)),
                                                       _np(sal), _np(ten), _np(dirs), C.byref(thr)))
        return sal, ten, dirs, thr.value

    # ---------------------------------------------------------------- device face (torch)
    def gauss_dev(self, src, dst, sigma, hw, mask=None, normalize=True):
        nz, ny, nx = src.shape
        A = C.c_float()
        self._chk(self._L.visfd_hip_apply_gauss_dev(self._h, _dev(src), _dev(dst), _dev(mask), nx, ny, nz,
                                                    _f3(sigma), _i3(hw), int(normalize), C.byref(A)))
        return A.value

    def local_fluctuations_dev(self, src, dst, sigma, ratio, mask=None, normalize=True, exponent=2.0):
        nz, ny, nx = src.shape
        self._chk(self._L.visfd_hip_local_fluctuations_dev(self._h, _dev(src), _dev(dst), _dev(mask), nx, ny, nz,
                                                           _f3(sigma), float(exponent), float(ratio), int(normalize)))

    def gauss_slab_dev(self, src, dst, z_lo, nz_global, sigma, hw, normalize=True):
        nz, ny, nx = src.shape
        A = C.c_float()
        self._chk(self._L.visfd_hip_apply_gauss_slab_dev(self._h, _dev(src), _dev(dst), nx, ny, nz, int(z_lo),
                                                         int(nz_global), _f3(sigma), _i3(hw), int(normalize),
                                                         C.byref(A)))
        return A.value

    def log_dev(self, src, dst, sigma, delta, ratio, mask=None):
        nz, ny, nx = src.shape
        self._chk(self._L.visfd_hip_apply_log_dev(self._h, _dev(src), _dev(dst), _dev(mask), nx, ny, nz,
                                                  _f3(sigma), float(delta), float(ratio), None, None))

    def blob_dog_dev(self, src, sigmas, mask=None, aspect=None, delta=0.02, ratio=2.5, minima_threshold=np.inf,
                     maxima_threshold=-np.inf, use_ratios=False, cap=1 << 16):
        return self._blob_call(self._L.visfd_hip_blob_dog_dev, _dev(src), _dev(mask), tuple(src.shape), sigmas,
                               aspect, delta, ratio, minima_threshold, maxima_threshold, use_ratios, cap)

    def calc_hessian_dev(self, src, grad, hess, sigma, ratio, mask=None):
        nz, ny, nx = src.shape
        self._chk(self._L.visfd_hip_calc_hessian_dev(self._h, _dev(src), _dev(grad), _dev(hess), _dev(mask), nx,
                                                     ny, nz, float(sigma), float(ratio)))

    def hessian_saliency_dev(self, hess, sal, dirs, order, mask=None):
        self._chk(self._L.visfd_hip_hessian_saliency_dev(self._h, _dev(hess), _dev(mask), sal.numel(), int(order),
                                                         _dev(sal), _dev(dirs)))

    def ridge_saliency_dev(self, src, sal, dirs, sigma, ratio, order, mask=None):
        nz, ny, nx = src.shape
        self._chk(self._L.visfd_hip_ridge_saliency_dev(self._h, _dev(src), _dev(mask), nx, ny, nz, float(sigma),
                                                       float(ratio), int(order), _dev(sal), _dev(dirs)))

    def ridge_scores_dev(self, src, sal, smoothed, sigma, ratio, order, mask=None, background=None):
        """Smoothing + Hessian + eigenvalues + score for every voxel; `smoothed` receives the smoothed volume.
        background (a volume from peak_background_dev): every score times (src - background)."""
        nz, ny, nx = src.shape
        self._chk(self._L.visfd_hip_ridge_scores_bg_dev(self._h, _dev(src), _dev(mask), nx, ny, nz, float(sigma),
                                                        float(ratio), int(order), _dev(background), _dev(sal), _dev(smoothed)))

    def peak_background_dev(self, src, background, sigma_background, ratio, mask=None, normalize=True):
        """The background of the peak-height factor: ApplyGauss(src, sigma_b, floor(sigma_b * ratio)) (handlers.cpp:1577-1592)."""
        nz, ny, nx = src.shape
        self._chk(self._L.visfd_hip_peak_background_dev(self._h, _dev(src), _dev(mask), nx, ny, nz, float(sigma_background),
                                                        float(ratio), int(bool(normalize)), _dev(background)))

    def ridge_directions_dev(self, smoothed, sal, dirs, sigma, order):
        """Principal directions of the voxels with sal != 0 (the others keep what dirs held)."""
        nz, ny, nx = smoothed.shape
        self._chk(self._L.visfd_hip_ridge_directions_dev(self._h, _dev(smoothed), nx, ny, nz, float(sigma), int(order),
                                                         _dev(sal), _dev(dirs)))

    def threshold_fraction_dev(self, sal, fraction, mask=None):
        thr = C.c_float()
        self._chk(self._L.visfd_hip_threshold_fraction_dev(self._h, _dev(sal), _dev(mask), sal.numel(),
                                                           float(fraction), C.byref(thr)))
        return thr.value

    def select_histogram_dev(self, sal, pass_, prefix, mask=None):
        hist = np.zeros(2048, np.uint64)
        n = C.c_uint64()
        self._chk(self._L.visfd_hip_select_histogram_dev(self._h, _dev(sal), _dev(mask), sal.numel(), int(pass_),
                                                         int(prefix), hist.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                         C.byref(n)))
        return hist, int(n.value)

    def select_histogram_todev(self, sal, pass_, prefix, hist, mask=None):
        """The round's histogram into `hist` (int64 tensor of 2048 on the device), asynchronous."""
        assert hist.is_cuda and hist.numel() == 2048 and hist.element_size() == 8 and hist.is_contiguous()
        self._chk(self._L.visfd_hip_select_histogram_todev(self._h, _dev(sal), _dev(mask), sal.numel(), int(pass_),
                                                           int(prefix), hist.data_ptr()))

    def stream_handle(self):
        """The hipStream_t (as an integer) this context's device face runs on."""
        return int(self._L.visfd_hip_get_stream(self._h) or 0)

    def apply_threshold_dev(self, sal, thr):
        self._chk(self._L.visfd_hip_apply_threshold_dev(self._h, _dev(sal), sal.numel(), float(thr)))

    def tv_dense_stick_dev(self, sal, dirs, tensor, sigma_tv, exponent=4, cutoff=2.0 ** 0.5, mask_src=None,
                           mask_dst=None, curves=False, z_out=None):
        nz, ny, nx = sal.shape
        z0, z1 = (0, nz) if z_out is None else z_out
        self._chk(self._L.visfd_hip_tv_dense_stick_slab_dev(self._h, _dev(sal), _dev(dirs), _dev(tensor),
                                                            _dev(mask_src), _dev(mask_dst), nx, ny, nz, int(z0),
                                                            int(z1), float(sigma_tv), int(exponent), float(cutoff),
                                                            int(curves)))

    def tensor_saliency_dev(self, tensor, sal, order, mask=None, image=None, background=None):
        self._chk(self._L.visfd_hip_tensor_saliency_bg_dev(self._h, _dev(tensor), _dev(mask), sal.numel(), int(order),
                                                           _dev(image), _dev(background), _dev(sal)))


class Slab:
    """One rank's handle of a Z-slab run (include/visfd_hip.h, section e; csrc/slab.hip): the layout, the transport and the
    slab forms of the stages.  `transport`:
      "rccl"   -- the library's own RCCL communicator (librccl.so loaded at run time): grouped send/recv with the two
                  Z-neighbours on the slab's transfer stream, all-reduces on the context's stream.  The 128-byte id is
                  made on rank 0 and broadcast here through torch.distributed (any backend);
      "torch"  -- callbacks into torch.distributed (host-staged copies): for backends without device point-to-point (gloo:
                  CPU tests, ranks sharing one GPU in a rehearsal).  The same C code drives both."""

    def __init__(self, ctx, rank, world, nz_global, ghost, transport="rccl", group=None):
        import torch
        import torch.distributed as dist
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self._L = load_library()
        self._h = _vp()
        self._keep = None
        if transport == "rccl_loopback":   # a one-rank RCCL communicator (selftest on a one-GPU box)
            assert world == 1
            idbuf = (C.c_char * 128)()
            ctx._chk(self._L.visfd_hip_slab_unique_id(C.addressof(idbuf)))
            ctx._chk(self._L.visfd_hip_slab_create_rccl(ctx._h, C.addressof(idbuf), 0, 1, nz_global, ghost, C.byref(self._h)))
        elif transport == "rccl" or world == 1:
            idbuf = (C.c_char * 128)()
            if world > 1:
                # Every rank takes the SAME steps whatever fails where: the broadcast always runs (an empty id says that rank 0
                # could not make one), then the ranks agree (MIN over "I have an id and can load RCCL") before any of them
                # enters ncclCommInitRank, which blocks until all have joined.
                ok_here = bool(self._L.visfd_hip_slab_rccl_available())
                box = [b""]
                if rank == 0 and ok_here and self._L.visfd_hip_slab_unique_id(C.addressof(idbuf)) == 0:
                    box = [bytes(idbuf)]
                dist.broadcast_object_list(box, src=0, group=group)
                ok_here = ok_here and len(box[0]) == 128
                on_gpu = dist.get_backend(group) == "nccl"
                flag = torch.tensor([1 if ok_here else 0], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                if not int(flag.item()):
                    raise VisfdHipError(2, "the library's RCCL communicator cannot come up on every rank (rank %d: %s)" % (
                        rank, "ok" if ok_here else "no id from rank 0 or librccl.so not loadable"))
                idbuf = (C.c_char * 128).from_buffer_copy(box[0])
            ctx._chk(self._L.visfd_hip_slab_create_rccl(ctx._h, C.addressof(idbuf) if world > 1 else None, rank, world, nz_global,
                                                        ghost, C.byref(self._h)))
        elif transport == "torch":
            def sync(stream):
                torch.cuda.ExternalStream(int(stream)).synchronize() if stream else torch.cuda.synchronize()

            def view(ptr, nbytes, dtype):
                itemsize = torch.empty((), dtype=dtype).element_size()
                return _device_view(torch, int(ptr), nbytes // itemsize, dtype)
            pending = []

            def sendrecv(user, peer, sendbuf, recvbuf, nbytes, stream):
                try:
                    sync(stream)
                    snd = view(sendbuf, nbytes, torch.float32).cpu()
                    rcv = torch.empty_like(snd)
                    pending.append((peer, snd, rcv, recvbuf, nbytes))
                    return 0
                except Exception as e:   # an exception must not cross the C frame
                    sys.stderr.write("visfd_amd.Slab transport: %r\n" % (e,))
                    return 1

            def group_end(user):
                try:
                    ops = []
                    for peer, snd, rcv, _, _ in pending:
                        ops.append(dist.P2POp(dist.isend, snd, peer, group))
                        ops.append(dist.P2POp(dist.irecv, rcv, peer, group))
                    for r in dist.batch_isend_irecv(ops):
                        r.wait()
                    for peer, snd, rcv, recvbuf, nbytes in pending:
                        view(recvbuf, nbytes, torch.float32).copy_(rcv)
                    torch.cuda.synchronize()
                    del pending[:]
                    return 0
                except Exception as e:
                    sys.stderr.write("visfd_amd.Slab transport: %r\n" % (e,))
                    return 1

            def allreduce(user, buf, count, stream):
                try:
                    sync(stream)
                    dv = view(buf, count * 8, torch.int64)
                    h = dv.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                    dv.copy_(h)
                    torch.cuda.synchronize()
                    return 0
                except Exception as e:
                    sys.stderr.write("visfd_amd.Slab transport: %r\n" % (e,))
                    return 1
            tr = Transport(_SENDRECV(sendrecv), _ALLREDUCE(allreduce), _GROUP(lambda user: 0), _GROUP(group_end), None)
            self._keep = tr   # the C side copies the struct; the CFUNCTYPE objects inside must outlive the slab
            ctx._chk(self._L.visfd_hip_slab_create_custom(ctx._h, C.addressof(tr), rank, world, nz_global, ghost, C.byref(self._h)))
        else:
            raise ValueError("transport must be 'rccl' or 'torch'")
        import weakref
        if not hasattr(ctx, "_slabs"):
            ctx._slabs = []
        ctx._slabs.append(weakref.ref(self))    # Context.close() destroys its slabs first
        lay = (_i64 * 7)()
        ctx._chk(self._L.visfd_hip_slab_layout(self._h, lay))
        self.z0, self.z1, self.lo, self.hi, self.own0, self.own1, self.nz_local = [int(v) for v in lay]
        self.nz_global, self.ghost = int(nz_global), int(ghost)

    def close(self):
        if self._h:
            self._L.visfd_hip_slab_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def owned(self, t):
        return t[..., self.own0:self.own1, :, :]

    def selftest(self, count=1 << 20):
        """visfd_hip_slab_selftest: self send/receive on the transfer stream + all-reduce, verified (raises on failure)."""
        self.ctx._chk(self._L.visfd_hip_slab_selftest(self._h, int(count)))

    def set_reserve(self, workgroups):
        self.ctx._chk(self._L.visfd_hip_slab_set_reserve(self._h, int(workgroups)))

    def exchange(self, tensors, depth):
        tensors = tensors if isinstance(tensors, (list, tuple)) else [tensors]
        nz, ny, nx = tensors[0].shape
        assert nz == self.nz_local
        arr = (_vp * len(tensors))(*[_dev(t) for t in tensors])
        self.ctx._chk(self._L.visfd_hip_slab_exchange_dev(self._h, arr, len(tensors), nx, ny, int(depth)))

    def membrane_detect(self, src, sal, dirs, tensor, scratch, sigma, ratio, order, best_fraction, sigma_tv, exponent=4,
                        cutoff=2.0 ** 0.5, src_halo_ready=False, sigma_background=0.0, background=None, normalize_background=True):
        nz, ny, nx = src.shape
        assert nz == self.nz_local and dirs.shape[0] == 3 and tensor.shape[0] == 6
        thr = C.c_float()
        self.ctx._chk(self._L.visfd_hip_membrane_detect_slab_bg_dev(
            self._h, _dev(src), _dev(sal), _dev(dirs), _dev(tensor), _dev(scratch), _dev(background), nx, ny, float(sigma),
            float(ratio), int(order), float(best_fraction), float(sigma_tv), int(exponent), float(cutoff), float(sigma_background),
            int(bool(normalize_background)), int(bool(src_halo_ready)), C.byref(thr)))
        return float(thr.value)

    def membrane_detect_host(self, src_owned, sigma, ratio, order, best_fraction, sigma_tv, exponent=4, cutoff=2.0 ** 0.5,
                             want_tensor=False, sigma_background=0.0, normalize_background=True):
        """visfd_hip_membrane_detect_slab: the slab stage on HOST arrays of the rank's owned planes (numpy float32
        [z1-z0][ny][nx]).  Returns (saliency of the owned planes, tensor [..., 6] or None, threshold)."""
        src_owned = np.ascontiguousarray(src_owned, np.float32)
        nzo, ny, nx = src_owned.shape
        assert nzo == self.z1 - self.z0
        sal = np.empty_like(src_owned)
        ten = np.empty(src_owned.shape + (6,), np.float32) if want_tensor else None
        thr = C.c_float()
        self.ctx._chk(self._L.visfd_hip_membrane_detect_slab_bg(
            self._h, _np(src_owned), nx, ny, float(sigma), float(ratio), int(order), float(best_fraction), float(sigma_tv),
            int(exponent), float(cutoff), float(sigma_background), int(bool(normalize_background)), _np(sal), _np(ten),
            C.byref(thr)))
        return sal, ten, float(thr.value)

    def blob_dog(self, src, sigmas, delta=0.02, ratio=2.5, minima_threshold=np.inf, maxima_threshold=-np.inf,
                 src_halo_ready=False, cap=1 << 22):
        """-> (minima, maxima) rows x, y, z (GLOBAL plane index), sigma, score of the OWNED planes."""
        nz, ny, nx = src.shape
        sig = np.ascontiguousarray(sigmas, np.float32)
        while True:
            mn, mx = (Blob * cap)(), (Blob * cap)()
            nmin, nmax = _i64(), _i64()
            rc = self._L.visfd_hip_blob_dog_slab_dev(self._h, _dev(src), nx, ny, sig.ctypes.data_as(_fp), len(sig), float(delta),
                                                     float(ratio), float(minima_threshold), float(maxima_threshold),
                                                     int(bool(src_halo_ready)), C.addressof(mn), cap, C.byref(nmin),
                                                     C.addressof(mx), cap, C.byref(nmax))
            if rc == 4:   # VISFD_HIP_ECAPACITY: a LOCAL retry -- the ghost planes are in place, so no exchange and no other rank
                cap = max(int(nmin.value), int(nmax.value), cap) + 16
                src_halo_ready = True
                continue
            self.ctx._chk(rc)
            break

        def rows(arr, n):
            a = np.frombuffer(arr, dtype=np.dtype([("ix", "<i4"), ("iy", "<i4"), ("iz", "<i4"), ("scale", "<i4"),
                                                   ("sigma", "<f4"), ("score", "<f4")]), count=n)
            out = np.empty((n, 5), np.float32)
            out[:, 0], out[:, 1], out[:, 2], out[:, 3], out[:, 4] = a["ix"], a["iy"], a["iz"], a["sigma"], a["score"]
            return out
        return rows(mn, int(nmin.value)), rows(mx, int(nmax.value))


def _device_view(torch, ptr, count, dtype):
    """A torch tensor over `count` elements of device memory at `ptr` (the library's buffers inside a transport callback)."""
    class _Iface:
        pass
    typestr = {torch.float32: "<f4", torch.int64: "<i8"}[dtype]
    obj = _Iface()
    obj.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(obj, device="cuda")
