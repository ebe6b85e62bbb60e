"""Assemble profiles/r01_gauss_traffic.json from two rocprofv3 --pmc passes of tools/pmc_traffic.py:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_traffic.py 1024
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/pmc_traffic.py 1024
    python tools/pmc_traffic_collect.py gpurun_out/pmc_fetch gpurun_out/pmc_write

The first two kernels of the run are calibration launches with known traffic (see tools/pmc_traffic.py): they fix
the unit and the gfx950 correction of FETCH_SIZE (it reports half of the streamed read bytes)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows(d):
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    return list(csv.DictReader(open(f)))


def main():
    dfetch, dwrite = sys.argv[1:3]
    n = 1024
    cal, vals = [], {"FETCH_SIZE": [], "WRITE_SIZE": []}
    kname = None
    for d in (dfetch, dwrite):
        for r in rows(d):
            name, c, v = r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])
            if c not in vals:
                continue
            if "apply_threshold_kernel" in name:
                cal.append([name[:52], c, v])
            elif "gauss_fused_kernel" in name:
                vals[c].append(v)
                kname = name[name.index("gauss_fused_kernel"):].split("(")[0]
    nbytes = 4.0 * n ** 3
    fetch_cal = [v for _, c, v in cal if c == "FETCH_SIZE"]
    write_cal = [v for _, c, v in cal if c == "WRITE_SIZE" and v > 0]
    fetch_factor = nbytes / (fetch_cal[0] * 1024.0)          # expected / reported for a pure 4 B/voxel read
    write_factor = nbytes / (write_cal[0] * 1024.0)
    read_b = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) * 1024.0 * fetch_factor
    write_b = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) * 1024.0 * write_factor
    out = {
        "kernel": kname + " sigma=2 h=5", "shape": [n, n, n], "unit": "bytes per launch",
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_traffic.py, "
                  "tools/pmc_traffic_collect.py); both scaled by calibration launches of known traffic in the same run "
                  "(apply_threshold(-inf) reads 4 GiB, apply_threshold(+inf) reads and writes 4 GiB): FETCH_SIZE x %.4f "
                  "(gfx950 reports half of the streamed read bytes), WRITE_SIZE x %.4f" % (fetch_factor, write_factor),
        "fetch_size_kb_raw": vals["FETCH_SIZE"], "write_size_kb_raw": vals["WRITE_SIZE"], "calibration": cal,
        "read_bytes": read_b, "write_bytes": write_b, "traffic_bytes": read_b + write_b,
        "algorithmic_bytes": int(2 * nbytes), "traffic_over_algorithmic": (read_b + write_b) / (2 * nbytes),
    }
    with open(os.path.join(ROOT, "profiles", "r01_gauss_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("read_bytes", "write_bytes", "traffic_bytes", "traffic_over_algorithmic")}))


if __name__ == "__main__":
    main()
