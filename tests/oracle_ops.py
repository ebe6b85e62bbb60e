"""A CPU stand-in for visfd_amd.api.Context's device face, backed by the oracle (tests only).

visfd_amd/slab.py reaches the stage kernels through an `ops` object; on a GPU that is the real
Context.  The world_size-2 gloo tests run on CPU, where the product has (by design) no compute path,
so they plug in this class: same method names and argument order, torch CPU tensors in place of
device tensors, oracle arithmetic.  What those tests exercise is the slab logic itself -- layout,
halo exchange, distributed radix select, list merging -- not the kernels."""
import numpy as np
import torch

from oracle import pyoracle as po


def _np(t):
    return None if t is None else t.numpy()


class OracleOps:
    def __init__(self):
        self.O = po.load("oracle")

    def gauss_dev(self, src, dst, sigma, hw, mask=None, normalize=True):
        out, A = self.O.gauss_hw(np.ascontiguousarray(_np(src)), sigma, hw, _np(mask), normalize)
        dst.copy_(torch.from_numpy(out))
        return A

    def ridge_saliency_dev(self, src, sal, dirs, sigma, ratio, order, mask=None):
        _, hess = self.O.calc_hessian(np.ascontiguousarray(_np(src)), sigma, ratio, _np(mask), want_grad=False)
        s, d = self.O.hessian_saliency(hess, order, _np(mask))
        sal.copy_(torch.from_numpy(s))
        dirs.copy_(torch.from_numpy(np.ascontiguousarray(np.moveaxis(d, -1, 0))))

    def select_histogram_dev(self, sal, rnd, prefix, mask=None):
        a = np.ascontiguousarray(_np(sal)).reshape(-1)
        u = a.view(np.uint32)
        key = np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000)).astype(np.uint32)
        if mask is not None:
            key = key[np.ascontiguousarray(_np(mask)).reshape(-1) != 0]
        n = key.size
        shift, dmask, pshift = ((21, 0x7FF, 32), (10, 0x7FF, 21), (0, 0x3FF, 10))[rnd]
        if rnd > 0:
            key = key[(key >> np.uint32(pshift)) == np.uint32(prefix)]
        digit = (key >> np.uint32(shift)) & np.uint32(dmask)
        return np.bincount(digit, minlength=2048).astype(np.uint64), n

    def apply_threshold_dev(self, sal, thr):
        sal[sal < np.float32(thr)] = 0.0

    def tv_dense_stick_dev(self, sal, dirs, tensor, sigma_tv, exponent=4, cutoff=2.0 ** 0.5, mask_src=None,
                           mask_dst=None, curves=False, z_out=None):
        d = np.ascontiguousarray(np.moveaxis(_np(dirs), 0, -1))
        t = self.O.tv_dense_stick(np.ascontiguousarray(_np(sal)), d, sigma_tv, exponent, cutoff, _np(mask_src),
                                  _np(mask_dst), curves)
        t = torch.from_numpy(np.ascontiguousarray(np.moveaxis(t, -1, 0)))
        z0, z1 = (0, sal.shape[0]) if z_out is None else z_out
        tensor[:, z0:z1] = t[:, z0:z1]

    def tensor_saliency_dev(self, tensor, sal, order, mask=None):
        t = np.ascontiguousarray(np.moveaxis(_np(tensor), 0, -1))
        s = np.ascontiguousarray(_np(sal))
        self.O.tensor_saliency(t, order, s, _np(mask))
        sal.copy_(torch.from_numpy(s))

    def blob_dog_dev(self, src, sigmas, mask=None, aspect=None, delta=0.02, ratio=2.5, minima_threshold=np.inf,
                     maxima_threshold=-np.inf, use_ratios=False, cap=1 << 20):
        return self.O.blob_dog(np.ascontiguousarray(_np(src)), sigmas, _np(mask), aspect, delta, ratio,
                               minima_threshold, maxima_threshold, use_ratios, cap)
