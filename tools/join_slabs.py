"""Stack the per-rank output files of `filter_mrc ... -slab RANK WORLD IDFILE` (mode-2 MRC files of consecutive Z-slabs) into
one MRC file:   python tools/join_slabs.py out.rec slab0.rec slab1.rec ...
The header is the first slab's, with nz, mz, the cell's z extent and the density statistics of the whole volume.

Blob lists (`-blob ... -slab`: every rank writes "<file>.slab<RANK>", rows x y z diameter score with GLOBAL z):
    python tools/join_slabs.py --blobs {minima|maxima} out.txt list.slab0 list.slab1 ...
merges them in the order the program writes a list: by score (ascending for minima, descending for maxima), ties in rank
order (= z order)."""
import sys

import numpy as np


def read(path):
    raw = open(path, "rb").read()
    w = np.frombuffer(raw[:1024], np.int32).copy()
    nx, ny, nz, mode, ext = int(w[0]), int(w[1]), int(w[2]), int(w[3]), int(w[23])
    if mode != 2:
        raise SystemExit("%s: expected a 32-bit float (mode 2) file" % path)
    data = np.frombuffer(raw, np.float32, nx * ny * nz, 1024 + ext).reshape(nz, ny, nx)
    return raw[:1024], data


def join_blobs(kind, out, parts):
    rows = []
    for p in parts:
        for ln in open(p):
            if ln.strip():
                rows.append((float(ln.split()[4]), ln))
    rows.sort(key=lambda r: r[0], reverse=(kind == "maxima"))   # (stable: ties keep rank order)
    with open(out, "w") as f:
        f.writelines(ln for _, ln in rows)


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--blobs":
        if sys.argv[2] not in ("minima", "maxima"):
            raise SystemExit(__doc__)
        return join_blobs(sys.argv[2], sys.argv[3], sys.argv[4:])
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    parts = [read(p) for p in sys.argv[2:]]
    vol = np.concatenate([d for _, d in parts], 0)
    if any(d.shape[1:] != vol.shape[1:] for _, d in parts):
        raise SystemExit("the slabs differ in nx or ny")
    hdr = bytearray(parts[0][0])
    w = np.frombuffer(bytes(hdr[:96]), np.int32).copy()
    f = np.frombuffer(bytes(hdr[:96]), np.float32).copy()
    w[2] = w[9] = vol.shape[0]
    cz = sum(float(np.frombuffer(h[48:52], np.float32)[0]) for h, _ in parts)
    hdr[:96] = w.tobytes()
    hdr[48:52] = np.float32(cz).tobytes()
    hdr[76:88] = np.array([vol.min(), vol.max(), vol.mean(dtype=np.float64)], np.float32).tobytes()
    hdr[92:96] = np.int32(0).tobytes()
    del f
    with open(sys.argv[1], "wb") as out:
        out.write(bytes(hdr))
        out.write(np.ascontiguousarray(vol, np.float32).tobytes())


if __name__ == "__main__":
    main()
