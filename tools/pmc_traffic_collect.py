"""Assemble a traffic JSON from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/pmc_traffic.py or
tools/pmc_tv.py:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <fetch_dir> -- python3 tools/pmc_traffic.py 1024
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <write_dir> -- python3 tools/pmc_traffic.py 1024
    python tools/pmc_traffic_collect.py <fetch_dir> <write_dir> gauss_fused_kernel <out.json> 8 [n] [label]

The first kernels of either run are calibration launches with known traffic (apply_threshold(-inf) reads 4 B/voxel,
apply_threshold(+inf) reads and writes 4 B/voxel): they fix the unit and the gfx950 correction of FETCH_SIZE (it
reports half of the streamed read bytes)."""
import csv
import glob
import json
import os
import sys


def rows(d):
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    return list(csv.DictReader(open(f)))


def main():
    dfetch, dwrite, key, out_path, alg_per_voxel = sys.argv[1:6]
    n = int(sys.argv[6]) if len(sys.argv) > 6 else 1024
    label = sys.argv[7] if len(sys.argv) > 7 else ""
    cal, vals = [], {"FETCH_SIZE": [], "WRITE_SIZE": []}
    kname = None
    for d in (dfetch, dwrite):
        for r in rows(d):
            name, c, v = r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])
            if c not in vals:
                continue
            if "apply_threshold_kernel" in name:
                cal.append([name[:52], c, v])
            elif key in name:
                vals[c].append(v)
                kname = name[name.index(key):].split("(")[0]
    nbytes = 4.0 * n ** 3
    fetch_cal = [v for _, c, v in cal if c == "FETCH_SIZE"]
    write_cal = [v for _, c, v in cal if c == "WRITE_SIZE" and v > 0]
    fetch_factor = nbytes / (fetch_cal[0] * 1024.0)          # expected / reported for a pure 4 B/voxel read
    write_factor = nbytes / (write_cal[0] * 1024.0)
    read_b = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) * 1024.0 * fetch_factor
    write_b = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) * 1024.0 * write_factor
    alg = float(alg_per_voxel) * n ** 3
    out = {
        "kernel": (kname + " " + label).strip(), "shape": [n, n, n], "unit": "bytes per launch",
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh, "
                  "tools/pmc_traffic_collect.py); both scaled by calibration launches of known traffic in the same run "
                  "(apply_threshold(-inf) reads 4 B/voxel, apply_threshold(+inf) reads and writes 4 B/voxel): FETCH_SIZE x %.4f "
                  "(gfx950 reports half of the streamed read bytes), WRITE_SIZE x %.4f" % (fetch_factor, write_factor),
        "fetch_size_kb_raw": vals["FETCH_SIZE"], "write_size_kb_raw": vals["WRITE_SIZE"], "calibration": cal,
        "read_bytes": read_b, "write_bytes": write_b, "traffic_bytes": read_b + write_b,
        "algorithmic_bytes": int(alg), "traffic_over_algorithmic": (read_b + write_b) / alg,
    }
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("kernel", "read_bytes", "write_bytes", "traffic_bytes", "traffic_over_algorithmic")}))


if __name__ == "__main__":
    main()
