"""bench.py's output contract on a small volume (GPU): one JSON line with the fields the driver reads, the roofline
objects and the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", "128", "--steps", "1", "--warmup", "1",
                        "--no-2048", "--cpu-sample", "48"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["unit"] == "Mvoxels/s" and out["n_gpus"] == 1 and out["steps"] == 1 and out["higher_is_better"] is True
    assert out["scaling"] == "weak" and out["vs_baseline"] is None and out["dtype"] == "f32" and out["data"] == "synthetic"
    assert "workload" in out["config"] and "model" not in out["config"]
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert abs(out["value"] - 128 ** 3 / (out["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * out["value"]
    rf = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    # `roofline` describes the DOMINANT kernel of the step (tensor voting: a VALU-bound stencil priced in TFLOP/s against the
    # FP32 vector peak); the HBM-bound kernel BASELINE.json names is `roofline_gauss`
    assert rf["bound"] == "valu" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "tv_box_kernel" in rf["kernel"] and 0 < rf["share_of_step"] <= 1.0
    rg = out["roofline_gauss"]
    assert rg["bound"] == "hbm" and rg["unit"] == "GB/s" and abs(rg["frac"] - rg["achieved"] / rg["peak"]) < 1e-3
    assert abs(rg["frac_of_copy"] - rg["achieved"] / out["copy_gbs"]) < 1e-3 and rg["tolerance_mode"]["bound"] == "hbm"
    assert out["config"]["mode"] == "tolerance" and set(out["modes"]) == {"tolerance", "exact"}
    assert out["modes"]["exact"]["results"]["minima"] == out["modes"]["tolerance"]["results"]["minima"]   # indices: exact in both
    assert out["modes"]["exact"]["results"]["maxima"] == out["modes"]["tolerance"]["results"]["maxima"]
    cb = out["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "cpu_model", "omp"):
        assert key in cb, key
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0
    # the best of three OpenMP teams -- every host thread, 64, the per-GPU share of 16 -- each bound close, each in
    # a process of its own; all of them listed
    assert cb["omp"]["OMP_PROC_BIND"] == "close" and len(cb["teams"]) >= 1
    assert max(t["cores"] for t in cb["teams"]) == os.cpu_count() and cb["value"] == max(t["value"] for t in cb["teams"])
    assert set(out["roofline_tv"]) == {"exact", "tolerance"} and out["roofline_tv"]["exact"]["bound"] == "valu"
    assert out["roofline_pass"]["bound"] == "hbm" and "ridge_score_kernel" in out["roofline_ridge"]
    rr = out["roofline_ridge"]   # the eigen kernels against the FP64 vector peak
    for k in ("ridge_score_kernel", "ridge_directions_kernel", "tensor_saliency_kernel"):
        assert rr[k]["fp64_flop_per_voxel"] > 0 and 0 < rr[k]["frac_fp64"] < 1
        assert abs(rr[k]["frac_fp64"] - rr[k]["achieved_fp64_tflops"] / rr["peak_tflops_fp64_vector"]) < 2e-3
    rp = out["roofline_pipeline"]   # the BASELINE metric's own "% HBM roofline"
    assert rp["algorithmic_bytes_per_voxel"] == 336.0 and set(rp["stages"]) == {"gauss", "blob_dog", "membrane_tv"}
    assert abs(rp["frac"] - rp["achieved"] / rp["peak"]) < 1e-3
    assert abs(rp["achieved"] - 336.0 * 128 ** 3 / (out["ms_per_step"] * 1e-3) / 1e9) <= 0.02 * rp["achieved"] + 0.1
