"""Seeded synthetic volumes and small helpers shared by the tests, the golden generator and bench.py.

Everything is numpy/float32 and deterministic (numpy Generator PCG64 with fixed seeds), so the
inputs of the committed golden vectors can be rebuilt bit-for-bit anywhere.
Volumes are indexed [iz][iy][ix] (x fastest), like the reference (alloc3d.hpp:16-23).
"""
import math

import numpy as np

# ---- shared test configurations ------------------------------------------------------------
TAP_CASES = [(0.5, 1), (1.0, 2), (2.0, 5), (3.7, 9), (10.0, 26), (12.0, 31), (9.99, 25), (0.0, 2), (2.02, 5),
             (1.732, 4), (1.98, 5)]
GAUSS_SHAPE = (20, 24, 28)  # (nz, ny, nx)
ANISO_SIGMA = (2.0, 1.5, 1.0)  # (sx, sy, sz)
ANISO_HW = (5, 3, 2)
BLOB_SHAPE = (24, 26, 28)
BLOB_DIAMS = np.array([5, 5.8, 6.6, 7.5, 8.6, 9.8, 11], np.float32)
BLOB_MODES = {
    "none": dict(),
    "abs": dict(minima_threshold=-50.0, maxima_threshold=50.0, use_ratios=False),
    "ratio": dict(minima_threshold=0.5, maxima_threshold=0.5, use_ratios=True),
    # one-sided ratios: the infinite side is NOT disabled in the reference (feature.hpp:286-289, 369-372)
    "ratio_min_only": dict(minima_threshold=0.5, maxima_threshold=-np.inf, use_ratios=True),
    "ratio_max_only": dict(minima_threshold=np.inf, maxima_threshold=0.5, use_ratios=True),
}
MEM_SHAPE = (20, 22, 26)
MEM_SIGMA = np.float32(1.732)
MEM_TV_SIGMA = np.float32(3.2)
MEM_FRACTION = 0.2


def noise_volume(shape, seed, mean=1000.0, sd=100.0):
    rng = np.random.default_rng(seed)
    return (mean + sd * rng.standard_normal(shape)).astype(np.float32)


def block_mask(shape, seed):
    """0/1 mask with a few rectangular holes and one whole slab removed; ~75 % ones."""
    rng = np.random.default_rng(seed)
    nz, ny, nx = shape
    m = np.ones(shape, np.float32)
    m[:, :2, :] = 0
    for _ in range(3):
        z0, y0, x0 = (int(rng.integers(0, n - 4)) for n in shape)
        m[z0:z0 + 4, y0:y0 + 5, x0:x0 + 6] = 0
    sprinkle = rng.random(shape) < 0.05
    m[sprinkle] = 0
    return m


def blob_volume(shape, seed, nblobs=5):
    rng = np.random.default_rng(seed)
    v = noise_volume(shape, seed + 1000, sd=20.0).astype(np.float64)
    zz, yy, xx = np.mgrid[0:shape[0], 0:shape[1], 0:shape[2]]
    for i in range(nblobs):
        c = [rng.uniform(5, n - 5) for n in shape]
        s = rng.uniform(2.0, 3.2)
        amp = -300.0 if i % 3 != 2 else 250.0
        v += amp * np.exp(-((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) / (2 * s * s))
    return v.astype(np.float32)


def membrane_volume(shape, seed, sd=100.0, amp=-400.0, thickness=1.5):
    """Noise plus a tilted dark plane and a dark spherical shell."""
    rng = np.random.default_rng(seed)
    v = (1000.0 + sd * rng.standard_normal(shape)).astype(np.float64)
    zz, yy, xx = np.mgrid[0:shape[0], 0:shape[1], 0:shape[2]]
    nz, ny, nx = shape
    d_plane = (zz - nz / 2.0) + 0.3 * (xx - nx / 2.0) - 0.2 * (yy - ny / 2.0)
    d_plane /= math.sqrt(1 + 0.09 + 0.04)
    v += amp * np.exp(-(d_plane ** 2) / (2 * thickness ** 2))
    r = np.sqrt((zz - nz * 0.3) ** 2 + (yy - ny * 0.6) ** 2 + (xx - nx * 0.4) ** 2)
    v += amp * np.exp(-((r - min(shape) * 0.3) ** 2) / (2 * thickness ** 2))
    return v.astype(np.float32)


def three_membranes(n=64, seed=411):
    """Noise (1000 +- 100) with three dark membranes that do not touch: two gently tilted planes near z = 12 and z = 51
    and a spherical shell of radius 9 between them -- three clusters for `-connect`."""
    rng = np.random.default_rng(seed)
    v = rng.normal(1000.0, 100.0, (n, n, n)).astype(np.float32)
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    d1 = (z - 12.0) + 0.1 * (x - 32)
    d2 = (z - 51.0) - 0.08 * (y - 32)
    r = np.sqrt((z - 32.0) ** 2 + (y - 32.0) ** 2 + (x - 31.0) ** 2) - 9.0
    for d in (d1, d2, r):
        v += (-400.0 * np.exp(-0.5 * (d / 1.5) ** 2)).astype(np.float32)
    return v


def eigen_cases(seed, nrand=4096):
    """Flat symmetric matrices (xx,yy,zz,xy,yz,xz): degenerate, diagonal, tiny, huge, random."""
    spec = np.array([
        [1, 1, 1, 0, 0, 0], [3, 2, 1, 0, 0, 0], [1, 2, 3, 0, 0, 0], [2, 2, 1, 0, 0, 0], [1, 2, 2, 0, 0, 0],
        [0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1], [0, 0, 0, 1, 0, 0], [5, 5, 5, 1e-7, 0, 0],
        [-1, -4, 2, 0.5, 0.25, -0.3], [1e-30, 2e-30, 3e-30, 0, 0, 0], [1e20, 2e20, -1e20, 3e19, 0, 1e19],
        [2, 2, 2, 0, 0, 1], [-3, -3, 7, 0, 0, 0],
    ], np.float32)
    rng = np.random.default_rng(seed)
    rnd = rng.standard_normal((nrand, 6)).astype(np.float32)
    rnd[: nrand // 4] *= np.float32(1e3)
    return np.ascontiguousarray(np.concatenate([spec, rnd], 0))


# ---- the CLI's blob-diameter ladder (bin/filter_mrc/settings.cpp:1719-1750) ----------------
def cli_blob_diameters(width_min, width_max, growth, multiplier=1.0):
    wmin, wmax, g = np.float32(width_min), np.float32(width_max), np.float32(growth)
    # log(float)/log(float) are float; ceil -> int
    N = 1 + int(math.ceil(np.float32(np.log(np.float32(wmax / wmin), dtype=np.float32) /
                                     np.log(g, dtype=np.float32))))
    # pow(float, double) is evaluated in double and stored to float
    g = np.float32(math.pow(float(np.float32(wmax / wmin)), 1.0 / N))
    d = np.empty(N, np.float32)
    d[0] = np.float32(wmin * np.float32(multiplier))
    for n in range(1, N):
        d[n] = np.float32(d[n - 1] * g)
    return d


def sort_blobs(rows, ascending=True):
    """Deterministic order for comparing blob lists (rows: x,y,z,sigma,score): by score, ties by
    (sigma,z,y,x).  The reference's in-memory order depends on thread scheduling
    (feature.hpp:310-345); its CLI sorts by score before writing (handlers.cpp:876-909)."""
    if len(rows) == 0:
        return rows.reshape(0, 5)
    key = rows[:, 4] if ascending else -rows[:, 4]
    idx = np.lexsort((rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3], key))
    return np.ascontiguousarray(rows[idx])


# ---- minimal MRC reader/writer (tests only; clean-room from the MRC2014 layout) ------------
def read_mrc(path):
    with open(path, "rb") as f:
        raw = f.read()
    h = np.frombuffer(raw[:1024], dtype="<i4")
    nx, ny, nz, mode = (int(v) for v in h[0:4])
    nsymbt = int(h[23])
    signed = True
    if mode == 0:
        if path.endswith(".rec"):
            signed = False
        if int(h[38]) == 1146047817:  # IMOD stamp; bit 0 of the flags = signed bytes
            signed = bool(int(h[39]) & 1)
    dt = {0: ("i1" if signed else "u1"), 1: "<i2", 2: "<f4", 6: "<u2"}[mode]
    a = np.frombuffer(raw[1024 + nsymbt:], dtype=dt, count=nx * ny * nz)
    return np.ascontiguousarray(a.reshape(nz, ny, nx).astype(np.float32))


def write_mrc(path, vol, voxel_width=1.0, mode=2, imod_flags=None):
    """Writes a minimal MRC2014 file.  mode 2: float32; 0: bytes (signed unless the IMOD stamp + flag bit 0 say
    otherwise or the name ends in .rec -- the reference's rules); 1: int16; 6: uint16."""
    nz, ny, nx = vol.shape
    dt = {0: "i1", 1: "<i2", 2: "<f4", 6: "<u2"}[mode]
    if mode == 0 and (path.endswith(".rec") or (imod_flags is not None and not (imod_flags & 1))):
        dt = "u1"
    data = np.ascontiguousarray(vol).astype(dt)
    h = np.zeros(256, "<i4")
    h[0:3] = (nx, ny, nz)
    h[3] = mode
    h[7:10] = (nx, ny, nz)
    hf = h.view("<f4")
    hf[10:13] = (nx * voxel_width, ny * voxel_width, nz * voxel_width)
    hf[13:16] = 90.0
    h[16:19] = (1, 2, 3)
    hf[19:22] = (float(data.min()), float(data.max()), float(data.astype(np.float64).mean()))
    if imod_flags is not None:
        h[38] = 1146047817   # "IMOD"
        h[39] = imod_flags
    raw = bytearray(h.tobytes())
    raw[208:212] = b"MAP "
    raw[212:216] = bytes([0x44, 0x44, 0, 0])
    with open(path, "wb") as f:
        f.write(bytes(raw))
        f.write(data.tobytes())
