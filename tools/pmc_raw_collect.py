"""Sum the per-dispatch raw counters of tools/pmc_raw.sh per kernel and print one table (development aid)."""
import csv
import glob
import os
import sys
from collections import defaultdict

KEYS = ("tv_boxx_kernel", "tv_box_kernel", "tvl_row_kernel", "tvl_scan_kernel", "tv_tiled_kernel", "gauss_fused_kernel", "ridge_score_kernel", "ridge_directions_kernel", "ridge_fused_kernel",
        "blob_candidates_kernel", "blob_verify_kernel", "tensor_saliency_kernel", "conv_march_kernel", "conv_row_kernel")


def key(full):
    for k in KEYS:
        if k in full:
            return k
    return None


out = sys.argv[1]
tab = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in sorted(glob.glob(os.path.join(out, "*_p[0-9]"))):
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = key(r["Kernel_Name"])
            if name:
                tab[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "*", "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            name = key(r["Kernel_Name"])
            dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for name in sorted(tab):
        line = "%s  (launch ms under PMC: %s)" % (name, ", ".join("%.3f" % x for x in dur[name][:6]))
        print(line); fo.write(line + "\n")
        for c in sorted(tab[name]):
            v = tab[name][c]
            line = "    %-24s per launch %.6g   (launches %d)" % (c, sum(v) / len(v), len(v))
            print(line); fo.write(line + "\n")
