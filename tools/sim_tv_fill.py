"""CPU simulation (numpy/scipy only) of how full the vote steps of a wave are in tensor voting, for two schemes:

  sweep     every sender that reaches at least one of the wave's 64 receivers costs one vote step (what tv_tiled.hip does);
  streams   senders are tested in groups of G, every lane then walks ITS OWN hits of the group: a group costs as many
            steps as its busiest lane has hits (per-lane hit masks; round 1's kernel was a variant of this).

Senders: the top 5 % of a planar-ridge saliency (largest Hessian eigenvalue of Gaussian-smoothed noise, sigma 1.73),
optionally with a tilted membrane as in bench.py; window half-width 12; 8 x 8 wave patches; rows culled to the wave's
reach as in the kernel.  Prints lane use / fill and the instruction-cost model of DESIGN.md 4.2.

    python tools/sim_tv_fill.py [mem]
"""
import sys

import numpy as np
from scipy import ndimage


def main():
    rng = np.random.default_rng(3)
    shape = (64, 128, 128)
    src = rng.normal(1000, 100, shape)
    z, y, x = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    if len(sys.argv) > 1 and sys.argv[1] == "mem":
        dist = (0.15 * x - 0.1 * y + 1.0 * z - 0.5 * shape[0]) / np.sqrt(0.15 ** 2 + 0.1 ** 2 + 1)
        src -= 400 * np.exp(-dist ** 2 / (2 * 1.5 ** 2))
    s = ndimage.gaussian_filter(src, 1.7320508, mode="constant")
    g = np.gradient(s)
    H = np.empty(shape + (3, 3))
    for i in range(3):
        gi = np.gradient(g[i])
        for j in range(3):
            H[..., i, j] = gi[j]
    ev = np.linalg.eigvalsh(H)                      # ascending
    sal = ev[..., 2] - np.abs(ev[..., 1])           # dark sheet: one large positive eigenvalue
    salient = sal >= np.quantile(sal, 0.95)
    print("salient fraction %.4f" % salient.mean())
    h = 12
    nz, ny, nx = shape
    lx = np.arange(8)
    tests = sweep = votes = 0
    steps = {32: 0, 64: 0}
    for rz in range(h, nz - h, 6):
        for py in range(h + 8, ny - h - 16, 24):
            for px in range(h + 8, nx - h - 16, 24):
                RX = (px + lx)[None, :].repeat(8, 0).ravel()
                RY = (py + lx)[:, None].repeat(8, 1).ravel()
                for sz in range(rz + h, rz - h - 1, -1):
                    rho2 = h * h - (rz - sz) ** 2
                    rho = int(np.floor(np.sqrt(rho2)))
                    y0, x0 = py - rho, px - h
                    ys, xs = np.nonzero(salient[sz, y0:py + 8 + rho, x0:px + 8 + h])
                    if len(ys) == 0:
                        continue
                    order = np.lexsort((-xs, -ys))          # vote order: descending position
                    ys, xs = ys[order] + y0, xs[order] + x0
                    hit = (RY[None, :] - ys[:, None]) ** 2 + (RX[None, :] - xs[:, None]) ** 2 <= rho2
                    tests += len(ys)
                    sweep += int(hit.any(1).sum())
                    votes += int(hit.sum())
                    for G in steps:
                        steps[G] += sum(int(hit[g0:g0 + G].sum(0).max()) for g0 in range(0, len(ys), G))
    print("tests %d, sweep steps %d, lane votes %d" % (tests, sweep, votes))
    print("lane use of a sweep step %.3f; fill of a stream step: groups of 32 %.3f, of 64 %.3f"
          % (votes / 64 / sweep, votes / 64 / steps[32], votes / 64 / steps[64]))
    old = 33 * sweep + 4.25 * tests      # issue slots: 33 per voted sender, dot4 + cmp per tested one
    for G in steps:
        new = 42 * steps[G] + 4.0 * tests   # + ffbh, bit clear, entry address, exec test per step; dot4 + alignbit per test
        print("issue-slot model, streams in groups of %d / sweep = %.3f" % (G, new / old))


if __name__ == "__main__":
    main()
