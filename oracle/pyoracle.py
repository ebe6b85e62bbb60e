"""ctypes front-end for the CHECKER libraries (test infrastructure only).

`load("oracle")` -> oracle/libvisfd_oracle.so   (from-scratch CPU restatement, prefix vo_)
`load("ref")`    -> oracle/_ref/libvisfd_ref.so (the real reference templates, prefix vr_;
                    only present when it was built in the container that has /root/reference)

Both expose the same calls on numpy float32 volumes indexed [iz][iy][ix].
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_lp = C.POINTER(C.c_int64)

ORDER_INCREASING = 0  # selfadjoint_eigen3::INCREASING_EIVALS (eigen3_simple.hpp:36-43)
ORDER_DECREASING = 1


def _f(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(_fp)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


class CpuLib:
    def __init__(self, path, prefix):
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path)
        L, p = self.lib, prefix
        g = lambda name: getattr(L, p + name)
        sig = {
            "gauss_taps": (None, [C.c_float, C.c_int, _fp]),
            "ratio_from_threshold": (C.c_float, [C.c_float]),
            "apply_gauss_hw": (C.c_float, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _ip, C.c_int]),
            "apply_gauss_ratio": (C.c_float, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, C.c_float, C.c_int]),
            "apply_dog": (None, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _ip, _fp, _fp]),
            "apply_log": (None, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, C.c_float, C.c_float, _fp, _fp]),
            "blob_dog": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_float, C.c_float,
                                   C.c_float, C.c_float, C.c_int, _fp, C.c_int64, _lp, _fp, C.c_int64, _lp]),
            "blob_diameters_to_sigmas": (None, [_fp, C.c_int, _fp]),
            "blob_sigmas_to_diameters": (None, [_fp, C.c_int, _fp]),
            "calc_hessian": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]),
            "diagonalize_flat_sym3": (None, [_fp, _fp, C.c_int64, C.c_int]),
            "flat_sym_to_evects": (None, [_fp, _fp, _fp, C.c_int64, C.c_int]),
            "hessian_saliency": (None, [_fp, _fp, C.c_int64, C.c_int, _fp, _fp]),
            "threshold_fraction": (C.c_float, [_fp, _fp, C.c_int64, C.c_float]),
            "tv_tables": (C.c_int, [C.c_float, C.c_float, _fp, _fp, C.c_int]),
            "tv_dense_stick": (None, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                      C.c_float, C.c_int, C.c_int, C.c_int]),
            "tensor_saliency": (None, [_fp, _fp, C.c_int64, C.c_int, _fp]),
            "local_fluctuations": (None, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, C.c_float, C.c_float, C.c_int]),
            "bin_array3d": (C.c_int, [_fp, _ip, _fp, _ip, _ip]),
            "unbin_array3d": (C.c_int, [_fp, _ip, _fp, _ip, _ip]),
        }
        if prefix == "vr_":   # blob list post-processing: checked against the reference directly
            _up = C.POINTER(C.c_uint64)
            sig.update({
                "label_connected": (C.c_int64, [_fp, C.POINTER(C.c_int64), _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp,
                                                C.c_float, C.c_float, C.c_int, _fp, C.c_float, C.c_float, C.c_int,
                                                C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp,
                                                C.c_int64]),
                "label_connected_ex": (C.c_int64, [_fp, C.POINTER(C.c_int64), _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp,
                                                   C.c_float, C.c_float, C.c_int, _fp, C.c_float, C.c_float, C.c_int,
                                                   C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp,
                                                   C.c_int64, _fp, _fp, C.POINTER(C.c_int64), C.c_int64,
                                                   C.POINTER(C.c_int)]),
                "trace_product_sym3": (C.c_float, [_fp, _fp]),
                "diagonalize_sym3_f32": (None, [_fp, C.c_int, _fp, _fp]),
                "convert_flat_sym2_evects3": (None, [_fp, C.c_int, _fp, _fp]),
                "sphere_overlap": (C.c_float, [C.c_float, C.c_float, C.c_float]),
                "sort_blobs": (None, [_fp, _fp, _fp, C.c_int64, C.c_int, C.c_int, _up]),
                "discard_masked_blobs": (C.c_int64, [_fp, _fp, _fp, C.c_int64, _fp, C.c_int, C.c_int, C.c_int]),
                "discard_overlapping_blobs": (C.c_int64, [_fp, _fp, _fp, C.c_int64, C.c_float, C.c_float, C.c_float,
                                                          C.c_int, C.c_int]),
            })
        self._fn = {}
        for name, (res, args) in sig.items():
            fn = g(name)
            fn.restype = res
            fn.argtypes = args
            self._fn[name] = fn

    # ---- taps -------------------------------------------------------------------------
    def gauss_taps(self, sigma, h):
        out = np.empty(2 * h + 1, np.float32)
        self._fn["gauss_taps"](float(sigma), int(h), _f(out))
        return out

    def ratio_from_threshold(self, thr):
        return float(self._fn["ratio_from_threshold"](float(thr)))

    # ---- filters ----------------------------------------------------------------------
    def gauss_hw(self, src, sigma, hw, mask=None, normalize=True):
        nz, ny, nx = src.shape
        dst = np.empty_like(src)
        A = self._fn["apply_gauss_hw"](_f(src), _f(dst), _f(mask), nx, ny, nz, _f3(sigma), _i3(hw), int(normalize))
        return dst, float(A)

    def local_fluctuations(self, src, sigma, ratio, mask=None, normalize=True, exponent=2.0):
        nz, ny, nx = src.shape
        dst = np.empty_like(src)
        sg = np.asarray(sigma, np.float32)
        self._fn["local_fluctuations"](_f(src), _f(dst), _f(mask), nx, ny, nz, _f(sg), float(exponent), float(ratio),
                                       int(normalize))
        return dst

    def gauss_ratio(self, src, sigma, ratio, mask=None, normalize=True):
        nz, ny, nx = src.shape
        dst = np.empty_like(src)
        A = self._fn["apply_gauss_ratio"](_f(src), _f(dst), _f(mask), nx, ny, nz, _f3(sigma), float(ratio),
                                          int(normalize))
        return dst, float(A)

    def dog(self, src, sigma_a, sigma_b, hw, mask=None):
        nz, ny, nx = src.shape
        dst = np.empty_like(src)
        A = C.c_float()
        B = C.c_float()
        self._fn["apply_dog"](_f(src), _f(dst), _f(mask), nx, ny, nz, _f3(sigma_a), _f3(sigma_b), _i3(hw),
                              C.byref(A), C.byref(B))
        return dst, A.value, B.value

    def log(self, src, sigma, delta, ratio, mask=None):
        nz, ny, nx = src.shape
        dst = np.empty_like(src)
        A = C.c_float()
        B = C.c_float()
        self._fn["apply_log"](_f(src), _f(dst), _f(mask), nx, ny, nz, _f3(sigma), float(delta), float(ratio),
                              C.byref(A), C.byref(B))
        return dst, A.value, B.value

    # ---- blobs ------------------------------------------------------------------------
    def blob_dog(self, src, sigmas, mask=None, aspect=None, delta=0.02, ratio=2.5, minima_threshold=np.inf,
                 maxima_threshold=-np.inf, use_ratios=False, cap=1 << 20):
        nz, ny, nx = src.shape
        sig = np.ascontiguousarray(sigmas, np.float32)
        omin = np.empty((cap, 5), np.float32)
        omax = np.empty((cap, 5), np.float32)
        nmin = C.c_int64()
        nmax = C.c_int64()
        asp = _f3(aspect) if aspect is not None else None
        rc = self._fn["blob_dog"](_f(src), _f(mask), nx, ny, nz, _f(sig), len(sig), asp, float(delta), float(ratio),
                                  float(minima_threshold), float(maxima_threshold), int(use_ratios), _f(omin), cap,
                                  C.byref(nmin), _f(omax), cap, C.byref(nmax))
        if rc != 0:
            raise RuntimeError("blob capacity too small: %d %d" % (nmin.value, nmax.value))
        return omin[: nmin.value].copy(), omax[: nmax.value].copy()

    def diameters_to_sigmas(self, d):
        d = np.ascontiguousarray(d, np.float32)
        s = np.empty_like(d)
        self._fn["blob_diameters_to_sigmas"](_f(d), len(d), _f(s))
        return s

    def sigmas_to_diameters(self, s):
        s = np.ascontiguousarray(s, np.float32)
        d = np.empty_like(s)
        self._fn["blob_sigmas_to_diameters"](_f(s), len(s), _f(d))
        return d

    # ---- ridge detector ---------------------------------------------------------------
    def calc_hessian(self, src, sigma, ratio, mask=None, want_grad=True):
        nz, ny, nx = src.shape
        hess = np.zeros((nz, ny, nx, 6), np.float32)
        grad = np.zeros((nz, ny, nx, 3), np.float32) if want_grad else None
        rc = self._fn["calc_hessian"](_f(src), _f(grad), _f(hess), _f(mask), nx, ny, nz, float(sigma), float(ratio))
        if rc != 0:
            raise ValueError("CalcHessian requires an image at least 3 voxels wide")
        return grad, hess

    def diagonalize(self, m6, order):
        m6 = np.ascontiguousarray(m6, np.float32)
        out = np.empty_like(m6)
        self._fn["diagonalize_flat_sym3"](_f(m6), _f(out), m6.size // 6, int(order))
        return out

    def evects(self, m6, order):
        m6 = np.ascontiguousarray(m6, np.float32)
        n = m6.size // 6
        ev = np.empty((n, 3), np.float32)
        evec = np.empty((n, 3, 3), np.float32)
        self._fn["flat_sym_to_evects"](_f(m6), _f(ev), _f(evec), n, int(order))
        return ev, evec

    def hessian_saliency(self, hess, order, mask=None):
        shp = hess.shape[:-1]
        sal = np.empty(shp, np.float32)
        dirs = np.zeros(shp + (3,), np.float32)
        self._fn["hessian_saliency"](_f(hess), _f(mask), sal.size, int(order), _f(sal), _f(dirs))
        return sal, dirs

    def threshold_fraction(self, sal, fraction, mask=None):
        """In place; returns the threshold."""
        return float(self._fn["threshold_fraction"](_f(sal), _f(mask), sal.size, float(fraction)))

    def tv_tables(self, sigma_tv, cutoff):
        h = self._fn["tv_tables"](float(sigma_tv), float(cutoff), None, None, 1 << 20)
        n = 2 * h + 1
        w = np.empty((n, n, n), np.float32)
        r = np.empty((n, n, n, 3), np.float32)
        self._fn["tv_tables"](float(sigma_tv), float(cutoff), _f(w), _f(r), h)
        return h, w, r

    def tv_dense_stick(self, sal, dirs, sigma_tv, exponent=4, cutoff=2.0 ** 0.5, mask_src=None, mask_dst=None,
                       curves=False, normalize=False, diagonalize=False):
        nz, ny, nx = sal.shape
        tensor = np.zeros((nz, ny, nx, 6), np.float32)
        self._fn["tv_dense_stick"](_f(sal), _f(dirs), _f(tensor), _f(mask_src), _f(mask_dst), nx, ny, nz,
                                   float(sigma_tv), int(exponent), float(cutoff), int(curves), int(normalize),
                                   int(diagonalize))
        return tensor

    def tensor_saliency(self, tensor, order, sal_inout, mask=None):
        self._fn["tensor_saliency"](_f(tensor), _f(mask), sal_inout.size, int(order), _f(sal_inout))
        return sal_inout


    # ---- binning (resample.hpp:53-166); shapes are numpy [nz, ny, nx] ----------------------
    def _resample(self, name, src, dst_shape, offset):
        dst = np.empty(dst_shape, np.float32)
        ssz, dsz = _i3(src.shape[::-1]), _i3(tuple(dst_shape)[::-1])
        off = _i3(offset) if offset is not None else None
        if self._fn[name](_f(src), ssz, _f(dst), dsz, off):
            raise ValueError("bin offset out of range")
        return dst

    def bin_array3d(self, src, dst_shape, offset=None):
        return self._resample("bin_array3d", src, dst_shape, offset)

    def unbin_array3d(self, src, dst_shape, offset=None):
        return self._resample("unbin_array3d", src, dst_shape, offset)

    # ---- voxel clustering (reference harness only): same signature and returns as visfd_amd.api.label_connected
    def label_connected(self, saliency, threshold_saliency, mask=None, direction=None, tensor=None,
                        threshold_vector_saliency=-np.inf, threshold_vector_neighbor=-np.inf,
                        consider_dot_product_sign=True, threshold_tensor_saliency=-np.inf,
                        threshold_tensor_neighbor=-np.inf, tensor_is_positive_definite_near_target=True, connectivity=1,
                        label_undefined=-1, sort_by_size=True, standardize_directions=False,
                        start_from_saliency_maxima=True, voxel_weights=None, must_link=None, must_link_directions=None):
        nz, ny, nx = saliency.shape
        labels = np.empty((nz, ny, nx), np.int64)
        cap = int(saliency.size)
        cm, cs, csal = np.zeros((cap, 3), np.float32), np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        if voxel_weights is not None or must_link:
            ml_c = ml_n = ml_d = None
            ngroups = 0
            if must_link:
                ngroups = len(must_link)
                ml_c = np.ascontiguousarray(np.concatenate([np.asarray(g_, np.float32).reshape(-1, 3) for g_ in must_link], 0))
                ml_n = np.array([len(g_) for g_ in must_link], np.int64)
                if must_link_directions is not None:
                    ml_d = np.ascontiguousarray(np.concatenate([np.asarray(d_, np.int32).ravel() for d_ in must_link_directions]))
            k = self._fn["label_connected_ex"](
                _f(saliency), labels.ctypes.data_as(C.POINTER(C.c_int64)), _f(mask), nx, ny, nz, threshold_saliency,
                _f(direction), threshold_vector_saliency, threshold_vector_neighbor, int(bool(consider_dot_product_sign)),
                _f(tensor), threshold_tensor_saliency, threshold_tensor_neighbor,
                int(bool(tensor_is_positive_definite_near_target)), int(connectivity), int(label_undefined),
                int(bool(sort_by_size)), int(bool(standardize_directions)), int(bool(start_from_saliency_maxima)),
                _f(cm), _f(cs), _f(csal), cap, _f(voxel_weights), _f(ml_c),
                None if ml_n is None else ml_n.ctypes.data_as(C.POINTER(C.c_int64)), ngroups,
                None if ml_d is None else ml_d.ctypes.data_as(C.POINTER(C.c_int)))
            return labels, int(k), cm[:k], cs[:k], csal[:k]
        k = self._fn["label_connected"](
            _f(saliency), labels.ctypes.data_as(C.POINTER(C.c_int64)), _f(mask), nx, ny, nz, threshold_saliency,
            _f(direction), threshold_vector_saliency, threshold_vector_neighbor, int(bool(consider_dot_product_sign)),
            _f(tensor), threshold_tensor_saliency, threshold_tensor_neighbor,
            int(bool(tensor_is_positive_definite_near_target)), int(connectivity), int(label_undefined),
            int(bool(sort_by_size)), int(bool(standardize_directions)), int(bool(start_from_saliency_maxima)),
            _f(cm), _f(cs), _f(csal), cap)
        return labels, int(k), cm[:k], cs[:k], csal[:k]

    def diagonalize_sym3_f32(self, m, order):
        m = np.ascontiguousarray(m, np.float32)
        vals = np.empty(m.shape[:-2] + (3,), np.float32)
        vecs = np.empty(m.shape, np.float32)
        mf, vf, ef = m.reshape(-1, 9), vals.reshape(-1, 3), vecs.reshape(-1, 9)
        for i in range(len(mf)):
            self._fn["diagonalize_sym3_f32"](_f(mf[i]), int(order), _f(vf[i]), _f(ef[i]))
        return vals, vecs

    def convert_flat_sym2_evects3(self, m6, order):
        m6 = np.ascontiguousarray(m6, np.float32)
        vals = np.empty(m6.shape[:-1] + (3,), np.float32)
        vecs = np.empty(m6.shape[:-1] + (3, 3), np.float32)
        mf, vf, ef = m6.reshape(-1, 6), vals.reshape(-1, 3), vecs.reshape(-1, 9)
        for i in range(len(mf)):
            self._fn["convert_flat_sym2_evects3"](_f(mf[i]), int(order), _f(vf[i]), _f(ef[i]))
        return vals, vecs

    def trace_product_sym3(self, a, b):
        return float(self._fn["trace_product_sym3"](_f(np.ascontiguousarray(a, np.float32)),
                                                    _f(np.ascontiguousarray(b, np.float32))))

    # ---- blob list post-processing (reference harness only) -------------------------------
    @staticmethod
    def _blob_arrays(crds, diameters, scores):
        return (np.ascontiguousarray(crds, np.float32).reshape(-1, 3).copy(),
                np.ascontiguousarray(diameters, np.float32).copy(), np.ascontiguousarray(scores, np.float32).copy())

    def sphere_overlap(self, rij, ri, rj):
        return float(self._fn["sphere_overlap"](rij, ri, rj))

    def sort_blobs(self, crds, diameters, scores, criteria, ascending=True):
        c, d, s = self._blob_arrays(crds, diameters, scores)
        perm = np.zeros(len(d), np.uint64)
        self._fn["sort_blobs"](_f(c), _f(d), _f(s), len(d), int(criteria), int(bool(ascending)),
                               perm.ctypes.data_as(C.POINTER(C.c_uint64)))
        return c, d, s, perm

    def discard_masked_blobs(self, crds, diameters, scores, mask):
        c, d, s = self._blob_arrays(crds, diameters, scores)
        nz, ny, nx = mask.shape
        n = self._fn["discard_masked_blobs"](_f(c), _f(d), _f(s), len(d), _f(mask), nx, ny, nz)
        return c[:n], d[:n], s[:n]

    def discard_overlapping_blobs(self, crds, diameters, scores, min_sep, max_large=np.inf, max_small=np.inf,
                                  criteria=3, scale=6):
        c, d, s = self._blob_arrays(crds, diameters, scores)
        n = self._fn["discard_overlapping_blobs"](_f(c), _f(d), _f(s), len(d), min_sep, max_large, max_small,
                                                  int(criteria), int(scale))
        return c[:n], d[:n], s[:n]


_PATHS = {
    "oracle": (os.path.join(_HERE, "libvisfd_oracle.so"), "vo_"),
    "ref": (os.path.join(_HERE, "_ref", "libvisfd_ref.so"), "vr_"),
}
_cache = {}


def available(kind):
    return os.path.exists(_PATHS[kind][0])


def load(kind):
    if kind not in _cache:
        path, prefix = _PATHS[kind]
        if not os.path.exists(path):
            raise FileNotFoundError(path + " (run `make -C oracle`)")
        _cache[kind] = CpuLib(path, prefix)
    return _cache[kind]
