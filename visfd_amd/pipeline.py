"""Host-side orchestration of the hot path on device-resident volumes (torch tensors in HBM).

These functions mirror the reference's filter_mrc handlers for this path:
  gauss()            HandleGauss        bin/filter_mrc/handlers.cpp:218
  blob_detect()      HandleBlobDetector bin/filter_mrc/handlers.cpp:787  (BlobDogD part)
  membrane_detect()  HandleTV           bin/filter_mrc/handlers.cpp:1501 (up to the vote score, :1892)
They only sequence calls into the C ABI (visfd_amd.api.Context.*_dev); all arithmetic is in
libvisfd_hip.so.  PyTorch is used for device memory and streams only.
"""
import math

import numpy as np

from . import api

TRUNCATE_THRESHOLD = 0.03  # CLI default, bin/filter_mrc/settings.cpp:81


def cli_blob_sigmas(width_min, width_max, growth):
    """The sigma ladder of `-blob-s all <file> min max growth` (settings.cpp:1719-1750 with
    blob_width_multiplier = 2*sqrt(3), then feature.hpp:475)."""
    wmin, wmax, g = np.float32(width_min), np.float32(width_max), np.float32(growth)
    N = 1 + int(math.ceil(np.float32(np.log(np.float32(wmax / wmin), dtype=np.float32) / np.log(g, dtype=np.float32))))
    g = np.float32(math.pow(float(np.float32(wmax / wmin)), 1.0 / N))
    d = np.empty(N, np.float32)
    d[0] = np.float32(wmin * np.float32(2.0 * math.sqrt(3.0)))
    for n in range(1, N):
        d[n] = np.float32(d[n - 1] * g)
    return api.diameters_to_sigmas(d)


def gauss(ctx, src, dst, sigma, truncate_threshold=TRUNCATE_THRESHOLD, mask=None, normalize=True):
    """filter_mrc -gauss sigma (voxels)."""
    ratio = api.ratio_from_threshold(truncate_threshold)
    sig = (sigma,) * 3 if np.isscalar(sigma) else tuple(sigma)
    hw = api.gauss_halfwidths(sig, ratio)
    return ctx.gauss_dev(src, dst, sig, hw, mask, normalize)


def blob_detect(ctx, src, sigmas, truncate_threshold=TRUNCATE_THRESHOLD, delta=0.02, mask=None,
                minima_threshold=np.inf, maxima_threshold=-np.inf, use_ratios=False, cap=1 << 22):
    """filter_mrc -blob-s: returns (minima, maxima) rows x,y,z,sigma,score."""
    ratio = api.ratio_from_threshold(truncate_threshold)
    return ctx.blob_dog_dev(src, sigmas, mask, None, delta, ratio, minima_threshold, maxima_threshold,
                            use_ratios, cap)


def blob_detect_begin(ctx, src, sigmas, truncate_threshold=TRUNCATE_THRESHOLD, delta=0.02, mask=None,
                      minima_threshold=np.inf, maxima_threshold=-np.inf, use_ratios=False):
    """blob_detect in two halves: everything is queued here; blob_detect_end() hands the lists over.  Device work queued on
    the same context in between (the membrane stage) runs right behind the last scan, while the host still handles lists."""
    ratio = api.ratio_from_threshold(truncate_threshold)
    return ctx.blob_dog_begin_dev(src, sigmas, mask, None, delta, ratio, minima_threshold, maxima_threshold, use_ratios)


def blob_detect_end(ctx, job, cap=1 << 22):
    return ctx.blob_dog_end(job, cap)


def membrane_detect(ctx, src, sal, dirs, tensor, sigma, tv_sigma_ratio, tv_exponent=4, best_fraction=0.05,
                    truncate_threshold=TRUNCATE_THRESHOLD, tv_truncate_ratio=math.sqrt(2.0), minima=True,
                    mask=None, scratch=None, sigma_background=0.0, background=None):
    """filter_mrc -membrane {minima|maxima} -tv ratio -tv-angle-exponent n: fills `sal` with the
    post-voting saliency (lambda0 - lambda1 of the vote tensor), `tensor` with the 6 vote planes.
    Returns the saliency threshold that was applied before voting."""
    order = api.DECREASING_EIVALS if minima else api.INCREASING_EIVALS  # handlers.cpp:1524-1535
    ratio = api.ratio_from_threshold(truncate_threshold)
    # scores for every voxel, threshold, then directions of the survivors only (`scratch`: a volume-sized tensor
    # for the smoothed image; allocated here when the caller has none to lend).  Voxels below the threshold keep
    # whatever `dirs` held: nothing downstream reads them.
    smoothed = scratch if scratch is not None else src.new_empty(src.shape)
    bg = None
    if sigma_background > 0:   # -membrane-background: both scores times (src - background), handlers.cpp:1577-1605,1698-1702
        bg = background if background is not None else src.new_empty(src.shape)
        ctx.peak_background_dev(src, bg, sigma_background, ratio, mask)
    ctx.ridge_scores_dev(src, sal, smoothed, sigma, ratio, order, mask, bg)
    thr = ctx.threshold_fraction_dev(sal, best_fraction, mask)
    ctx.ridge_directions_dev(smoothed, sal, dirs, sigma, order)
    sigma_tv = float(np.float32(tv_sigma_ratio) * np.float32(sigma))  # settings.cpp:3535-3540
    ctx.tv_dense_stick_dev(sal, dirs, tensor, sigma_tv, tv_exponent, tv_truncate_ratio, mask, mask)
    ctx.tensor_saliency_dev(tensor, sal, order, mask, src if bg is not None else None, bg)
    return thr
