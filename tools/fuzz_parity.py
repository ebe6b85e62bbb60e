"""Randomised parity sweep: GPU (through the C ABI) against the CPU restatement on random shapes, windows,
masks and parameters, bit for bit.  A development aid beyond the fixed cases of tests/:

    python tools/fuzz_parity.py [--seconds 120] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import volgen  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from visfd_amd import api  # noqa: E402


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def close_rel(a, b, rtol=1e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and float(np.max(np.abs(a - b), initial=0.0)) <= rtol * float(np.max(np.abs(b), initial=0.0)) + 1e-300


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    O = po.load("oracle")
    ctx = api.Context(0)
    t0 = time.time()
    counts = {"gauss": 0, "log": 0, "blob": 0, "tv": 0, "bin": 0, "gauss_fma": 0, "tv_fma": 0}
    bad = []
    it = 0
    t_report = t0
    while time.time() - t0 < a.seconds:
        it += 1
        if time.time() - t_report > 60:    # a progress line per minute (a silent GPU job is taken to be hung)
            t_report = time.time()
            print("... %d cases, %d mismatches after %.0f s" % (sum(counts.values()), len(bad), t_report - t0), flush=True)
        kind = ["gauss", "log", "blob", "tv", "bin"][it % 5]
        shape = tuple(int(v) for v in rng.integers(3, 70, 3))
        if rng.random() < 0.4:
            shape = shape[:2] + (int(rng.integers(1, 18)) * 4,)      # nx % 4 == 0: single-sweep kernels
        src = volgen.membrane_volume(shape, seed=int(rng.integers(1 << 30))) if min(shape) >= 8 else \
            rng.normal(1000, 100, shape).astype(np.float32)
        mask = None
        if rng.random() < 0.4:
            mask = (rng.random(shape) > 0.3).astype(np.float32)
            if rng.random() < 0.3:
                mask *= rng.uniform(0.2, 2.0, shape).astype(np.float32)   # weighted mask
        try:
            if kind == "gauss":
                if rng.random() < 0.5:
                    h = int(rng.integers(0, 14))
                    hw, sigma = (h, h, h), (float(rng.uniform(0.3, 4.0)),) * 3
                else:
                    hw = tuple(int(v) for v in rng.integers(0, 9, 3))
                    sigma = tuple(float(v) for v in rng.uniform(0.3, 4.0, 3))
                norm = bool(rng.random() < 0.7)
                want = O.gauss_hw(src, sigma, hw, mask, norm)[0]
                ok = bits_equal(ctx.gauss_hw(src, sigma, hw, mask, norm)[0], want)
                desc = "gauss shape=%s hw=%s sigma=%s mask=%s norm=%s" % (shape, hw, sigma, mask is not None, norm)
                with ctx.options(gauss_fma=1):     # the tolerance mode on the same case (takes effect where it applies)
                    okf = close_rel(ctx.gauss_hw(src, sigma, hw, mask, norm)[0], want)
                counts["gauss_fma"] += 1
                if not okf:
                    bad.append("gauss_fma " + desc)
                    print("MISMATCH (tolerance mode):", desc, flush=True)
            elif kind == "log":
                sigma = (float(rng.uniform(0.5, 3.0)),) * 3
                ok = bits_equal(ctx.log(src, sigma, 0.02, 2.6482, mask)[0], O.log(src, sigma, 0.02, 2.6482, mask)[0])
                desc = "log shape=%s sigma=%s mask=%s" % (shape, sigma, mask is not None)
            elif kind == "blob":
                s0 = float(rng.uniform(0.8, 2.0))
                sig = (s0 * (1.0 + 0.25 * np.arange(int(rng.integers(3, 6))))).astype(np.float32)
                g = ctx.blob_dog(src, sig, mask, None, 0.02, 2.6482)
                w = O.blob_dog(src, sig, mask, None, 0.02, 2.6482)
                ok = all(bits_equal(volgen.sort_blobs(x, asc), volgen.sort_blobs(y, asc)) for x, y, asc in
                         ((g[0], w[0], True), (g[1], w[1], False)))
                desc = "blob shape=%s sigmas=%s mask=%s" % (shape, sig, mask is not None)
            elif kind == "tv":
                sal = (rng.random(shape) < rng.uniform(0.02, 0.5)).astype(np.float32) * rng.uniform(0.5, 3.0, shape).astype(np.float32)
                d = rng.standard_normal(shape + (3,)).astype(np.float32)
                d /= np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-6).astype(np.float32)
                d = np.ascontiguousarray(d, np.float32)
                sigma_tv = float(rng.uniform(0.8, 9.5))
                ex = int(rng.choice([2, 4]))
                opts = {}
                if rng.random() < 0.6:     # few persistent workgroups, short runs: every workgroup claims many units
                    opts = {"tv_max_wg": int(rng.integers(1, 5)), "tv_zrun": int(rng.integers(1, 9))}
                    if rng.random() < 0.3:
                        opts["tv_no_replay"] = 1
                if rng.random() < 0.3:     # the general exact kernel instead of the box kernel's exact form
                    opts["tv_exact_tiled"] = 1
                if rng.random() < 0.3:
                    opts["tv_poison"] = 1
                # a source mask sends the exact run to tv_tiled.hip: half of the masked cases mask the receivers only
                msrc = mask if (mask is None or rng.random() < 0.5) else None
                with ctx.options(**opts):
                    got = ctx.tv_dense_stick(sal, d, sigma_tv, ex, 2.0 ** 0.5, msrc, mask)
                want = O.tv_dense_stick(sal, d, sigma_tv, ex, 2.0 ** 0.5, msrc, mask)
                ok = bits_equal(got, want) if mask is None else bits_equal(got[mask != 0], want[mask != 0])
                desc = "tv shape=%s sigma_tv=%g exponent=%d mask=%s source mask=%s opts=%s" % (shape, sigma_tv, ex, mask is not None,
                                                                                        msrc is not None, opts)
                with ctx.options(tv_fma=1, **opts):     # the tolerance kernel on the same case
                    gotf = ctx.tv_dense_stick(sal, d, sigma_tv, ex, 2.0 ** 0.5, msrc, mask)
                    okf = close_rel(gotf, want) if mask is None else close_rel(gotf[mask != 0], want[mask != 0])
                counts["tv_fma"] += 1
                if not okf:
                    bad.append("tv_fma " + desc)
                    print("MISMATCH (tolerance mode):", desc, flush=True)
            else:
                b = tuple(int(v) for v in rng.integers(1, 5, 3))
                ds = tuple(max(1, shape[i] // b[i]) for i in range(3))
                off = tuple(int(rng.integers(0, max(1, min(shape[2 - i] // ds[2 - i], shape[2 - i] - ds[2 - i] * (shape[2 - i] // ds[2 - i]) + 1))))
                            for i in range(3))
                ok = bits_equal(ctx.bin_array3d(src, ds, off), O.bin_array3d(src, ds, off))
                ok = ok and bits_equal(ctx.unbin_array3d(O.bin_array3d(src, ds, off), shape, off),
                                       O.unbin_array3d(O.bin_array3d(src, ds, off), shape, off))
                desc = "bin shape=%s -> %s offset=%s" % (shape, ds, off)
        except (api.VisfdHipError, ValueError) as e:
            if kind == "bin":
                continue       # an offset that leaves the source: refused by both
            ok, desc = False, "%s raised %s" % (kind, e)
        counts[kind] += 1
        if not ok:
            bad.append(desc)
            print("MISMATCH:", desc, flush=True)
    ctx.close()
    print("fuzz: %d cases in %.0f s %s; mismatches: %d" % (sum(counts.values()), time.time() - t0, counts, len(bad)), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
