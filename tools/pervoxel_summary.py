"""Summarise gpurun_out/tolerance_pervoxel.jsonl (written by tests/conftest.py: assert_close_rel(pervoxel=...)) into the table
kept as profiles/rNN_tolerance_pervoxel.txt:   python tools/pervoxel_summary.py [jsonl] > profiles/r04_tolerance_pervoxel.txt"""
import json
import sys
from collections import OrderedDict

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tolerance_pervoxel.jsonl"
rows = [json.loads(line) for line in open(path) if line.strip()]
groups = OrderedDict()
for r in rows:
    key = r["what"].split(" {")[0].split(" [")[0]
    g = groups.setdefault(key, {"n": 0, "sig": 0, "bound": r["bound"], "max_rel": 0.0, "max_scale": 0.0, "over": 0.0})
    g["n"] += 1
    g["sig"] = max(g["sig"], r["significant"])
    g["max_rel"] = max(g["max_rel"], r["max_rel"])
    g["max_scale"] = max(g["max_scale"], r["max_err_of_scale"])
    g["over"] = max(g["over"], r["frac_over"])
print("# The per-voxel companion of the 1e-5 tolerance (tests/conftest.py: assert_close_rel(..., pervoxel=bound)), final `pytest -m gpu`")
print("# run of the round: for every tolerance assertion that carries it -- grouped by what is compared, over all option sweeps --")
print("# the number of assertions, the significant voxels (|reference| > 1e-3 of the field's scale) of the largest case, the bound")
print("# on the FRACTION of those voxels with |a - b| > 1e-5 |b| (their own value), the largest such fraction observed, the")
print("# largest |a - b| / |b| among them, and the largest |a - b| in units of the field's scale (the 1e-5 contract itself).")
print("%-64s %5s %10s %7s %9s %11s %12s" % ("what", "runs", "voxels", "bound", "fraction", "max |d|/|b|", "max/scale"))
for k, g in groups.items():
    print("%-64s %5d %10d %7.3g %9.2g %11.3g %12.3g" % (k[:64], g["n"], g["sig"], g["bound"], g["over"], g["max_rel"], g["max_scale"]))
