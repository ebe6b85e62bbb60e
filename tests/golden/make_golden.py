#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Run in the build container only (needs oracle/_ref/libvisfd_ref.so, which oracle/Makefile
compiles from the reference headers under /root/reference):

    make -C oracle && python tests/golden/make_golden.py

Outputs (committed): tests/golden/*.npz -- inputs (or the seed/recipe to rebuild them) and
the reference's outputs.  The three *.rec files in this directory are data fixtures copied
from the reference's own tests/ directory (test_blob_detect.rec, test_blob_detect_mask.rec,
test_image_membrane.rec).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyoracle as po  # noqa: E402
import volgen  # noqa: E402  (tests/volgen.py: seeded synthetic volumes + MRC reader)

R = po.load("ref")


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    ratio = R.ratio_from_threshold(0.03)  # CLI default truncate threshold (settings.cpp:81)

    # ---- taps (filter1d.hpp:409) -------------------------------------------------------
    taps = {}
    for s, h in volgen.TAP_CASES:
        taps["s%g_h%d" % (s, h)] = R.gauss_taps(s, h)
    save("taps", ratio=np.float32(ratio), **taps)

    # ---- Gaussian on the reference's blob fixture (BASELINE config 1) -------------------
    img = volgen.read_mrc(os.path.join(HERE, "test_blob_detect.rec"))
    msk = volgen.read_mrc(os.path.join(HERE, "test_blob_detect_mask.rec"))
    g, A = R.gauss_ratio(img, (2, 2, 2), ratio, None, True)
    gm, Am = R.gauss_ratio(img, (2, 2, 2), ratio, msk, True)
    save("gauss_blobrec", out=g, A=np.float32(A), out_masked=gm, A_masked=np.float32(Am))

    # ---- Gaussian / DoG / LoG on a seeded volume ----------------------------------------
    src = volgen.noise_volume(volgen.GAUSS_SHAPE, seed=101)
    mask = volgen.block_mask(volgen.GAUSS_SHAPE, seed=102)
    out = {}
    for tag, m in (("nomask", None), ("mask", mask)):
        for norm in (True, False):
            o, A = R.gauss_hw(src, volgen.ANISO_SIGMA, volgen.ANISO_HW, m, norm)
            out["aniso_%s_norm%d" % (tag, norm)] = o
            out["aniso_%s_norm%d_A" % (tag, norm)] = np.float32(A)
    o, _ = R.gauss_hw(np.ascontiguousarray(src[:4, :5, :3]), (2, 2, 2), (5, 5, 5))
    out["tiny_n_lt_window"] = o
    o, A, B = R.log(src, (2, 2, 2), 0.02, ratio, None)
    out["log_nomask"] = o
    out["log_AB"] = np.array([A, B], np.float32)
    o, _, _ = R.log(src, (2.5, 2, 1.5), 0.02, ratio, mask)
    out["log_mask_aniso"] = o
    o, A, B = R.dog(src, (1.5, 1.5, 1.5), (2.5, 2.5, 2.5), (6, 6, 6), None)
    out["dog_nomask"] = o
    save("gauss_seeded", **out)

    # ---- blob detection: the reference's own test command --------------------------------
    #  filter_mrc -w 19.6 -mask test_blob_detect_mask.rec -in test_blob_detect.rec
    #             -blob minima test_blobs.txt 160.0 280.0 1.01   (tests/test_blob_detection.sh:21)
    diam = volgen.cli_blob_diameters(160.0, 280.0, 1.01, 1.0) / np.float32(19.6)
    sig = R.diameters_to_sigmas(diam)
    mins, maxs = R.blob_dog(img, sig, msk, None, 0.02, ratio, 0.0, -np.inf, False)
    mins = volgen.sort_blobs(mins, ascending=True)
    d = R.sigmas_to_diameters(np.ascontiguousarray(mins[:, 3]))
    save("blob_rec", diam_vox=diam, sigmas=sig, minima=mins, minima_diam_vox=d)
    print("blob test: %d minima; best = %s  (x,y,z,d in physical units: %s)" % (
        len(mins), mins[0], (mins[0, :3] * np.float32(19.6), d[0] * np.float32(19.6))))

    # seeded blobs, all three threshold modes, masked/unmasked
    bsrc = volgen.blob_volume(volgen.BLOB_SHAPE, seed=201)
    bmask = volgen.block_mask(volgen.BLOB_SHAPE, seed=202)
    bsig = R.diameters_to_sigmas(volgen.BLOB_DIAMS)
    out = {"sigmas": bsig}
    for tag, m in (("nomask", None), ("mask", bmask)):
        for mode, kw in volgen.BLOB_MODES.items():
            a, b = R.blob_dog(bsrc, bsig, m, None, 0.02, ratio, **kw)
            out["%s_%s_min" % (tag, mode)] = volgen.sort_blobs(a, True)
            out["%s_%s_max" % (tag, mode)] = volgen.sort_blobs(b, False)
    save("blob_seeded", **out)

    # ---- ridge detector + tensor voting on the reference's membrane fixture -------------
    mem = volgen.read_mrc(os.path.join(HERE, "test_image_membrane.rec"))
    out = {}
    sigma = np.float32(1.5)
    for oname, order in (("dec", po.ORDER_DECREASING), ("inc", po.ORDER_INCREASING)):
        grad, hess = R.calc_hessian(mem, sigma, ratio, None)
        sal, dirs = R.hessian_saliency(hess, order, None)
        out["hess"] = hess
        out["grad"] = grad
        out["sal_" + oname] = sal.copy()
        out["dir_" + oname] = dirs
        thr = R.threshold_fraction(sal, 0.1, None)
        out["thr_" + oname] = np.float32(thr)
        out["salthr_" + oname] = sal.copy()
        ten = R.tv_dense_stick(sal, dirs, 4 * sigma / 2, 4, 2.0 ** 0.5)
        out["tensor_" + oname] = ten
        s2 = sal.copy()
        R.tensor_saliency(ten, order, s2, None)
        out["tvsal_" + oname] = s2
    save("membrane_rec", **out)

    # seeded membrane volume: masks, exponents 2/4/3, curve mode
    msrc = volgen.membrane_volume(volgen.MEM_SHAPE, seed=301)
    mmask = volgen.block_mask(volgen.MEM_SHAPE, seed=302)
    out = {}
    for tag, m in (("nomask", None), ("mask", mmask)):
        grad, hess = R.calc_hessian(msrc, volgen.MEM_SIGMA, ratio, m)
        sal, dirs = R.hessian_saliency(hess, po.ORDER_DECREASING, m)
        out[tag + "_hess"] = hess
        out[tag + "_grad"] = grad
        out[tag + "_sal"] = sal.copy()
        out[tag + "_dir"] = dirs
        out[tag + "_thr"] = np.float32(R.threshold_fraction(sal, volgen.MEM_FRACTION, m))
        out[tag + "_salthr"] = sal.copy()
        for ex in (4, 2, 3):
            ten = R.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, ex, 2.0 ** 0.5, m, m)
            out["%s_tensor_e%d" % (tag, ex)] = ten
        ten = R.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, 4, 2.0 ** 0.5, m, m)
        s2 = sal.copy()
        R.tensor_saliency(ten, po.ORDER_DECREASING, s2, m)
        out[tag + "_tvsal"] = s2
        out[tag + "_tensor_curves"] = R.tv_dense_stick(sal, dirs, volgen.MEM_TV_SIGMA, 4, 2.0 ** 0.5, m, m,
                                                       curves=True)
    h, w, rh = R.tv_tables(8.66, 2.0 ** 0.5)
    out["tvtab_h12_w"] = w
    h, w, rh = R.tv_tables(volgen.MEM_TV_SIGMA, 2.0 ** 0.5)
    out["tvtab_w"] = w
    out["tvtab_rhat"] = rh
    save("membrane_seeded", **out)

    # ---- eigen solver known answers (eigen3_simple.hpp:137,271,392) ----------------------
    mats = volgen.eigen_cases(seed=401)
    out = {"mats": mats}
    for oname, order in (("inc", 0), ("dec", 1)):
        out["diag_" + oname] = R.diagonalize(mats, order)
        ev, evec = R.evects(mats, order)
        out["evals_" + oname] = ev
        out["evecs_" + oname] = evec
    save("eigen", **out)


if __name__ == "__main__":
    main()
