#!/usr/bin/env python3
"""bench.py -- throughput of VISFD's dense 3-D filtering hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size S]

A "step" is one pass of the whole hot path (BASELINE.json: "Gauss+DoG+TV pipeline") over one
synthetic float32 volume that is already resident in HBM:

  1. separable Gaussian, sigma = 2 voxels (11 taps)                      [BASELINE config 2]
  2. DoG/LoG scale-space blob detection, 12 scales sigma = 2..4          [BASELINE config 3]
     (`filter_mrc -blob-s all out 2.0 4.0 1.066 -w 1 -bin 1`)
  3. membrane detection with tensor voting                               [BASELINE config 4]
     (`filter_mrc -membrane minima 3 -tv 5 -tv-angle-exponent 4 -bin 1 -w 1`:
      sigma = 1.732, top 5 % salient, sigma_tv = 8.66 -> 25^3 vote window)

N = 1: one S^3 volume (default 1024^3).  N > 1 (launched by torch.distributed.run, one rank per GPU):
weak scaling -- a volume of S x S x (S*N) voxels cut into N Z-slabs with RCCL neighbour halo
exchange and an all-reduced radix select (visfd_amd/slab.py); value = all voxels / max-over-ranks time.

Prints ONE JSON line on rank 0.  `roofline` describes the separable-Gaussian kernel (the kernel
BASELINE.json's HBM-roofline target names), timed alone with HIP events on the stream it runs on;
`stages` gives the per-stage split of the pipeline step; `cpu_baseline` is the real reference (or
the CPU restatement when oracle/_ref is absent) timed on this box's host cores on a small sample.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

os.environ.setdefault("OMP_NUM_THREADS", "16")  # host threads for the CPU baseline (the box share per GPU)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
GAUSS_SIGMA = 2.0
BLOB = (2.0, 4.0, 1.066)
MEMBRANE = dict(sigma=1.7320508, tv_sigma_ratio=5.0, tv_exponent=4, best_fraction=0.05)


def synth_volume(torch, ctx, shape, device, seed, z_offset=0, nz_global=None):
    """normal(1000,100) background + dark blobs + three dark membranes, built on the
    device.  Planes are generated from their GLOBAL z index so that slabs tile one volume."""
    from visfd_amd import pipeline
    nz, ny, nx = shape
    nzg = nz_global or nz
    g = torch.Generator(device=device)
    vol = torch.empty(shape, device=device, dtype=torch.float32)
    for iz in range(nz):  # per-plane seeds: identical planes whatever the decomposition
        g.manual_seed(seed * 1000003 + z_offset + iz)
        vol[iz] = torch.randn((ny, nx), device=device, generator=g, dtype=torch.float32)
    vol.mul_(100.0).add_(1000.0)
    # membranes: two tilted planes and a spherical shell, 3 voxels thick, amplitude -400
    z = (torch.arange(nz, device=device, dtype=torch.float32) + z_offset).view(nz, 1, 1)
    y = torch.arange(ny, device=device, dtype=torch.float32).view(1, ny, 1)
    x = torch.arange(nx, device=device, dtype=torch.float32).view(1, 1, nx)
    for (a, b, c, d0) in ((0.15, -0.1, 1.0, 0.35 * nzg), (1.0, 0.2, 0.1, 0.6 * nx)):
        nrm = math.sqrt(a * a + b * b + c * c)
        for iz0 in range(0, nz, 64):
            sl = slice(iz0, min(nz, iz0 + 64))
            dist = (a * x + b * y + c * z[sl] - d0) / nrm
            vol[sl] -= 400.0 * torch.exp(-(dist * dist) / (2 * 1.5 * 1.5))
    for iz0 in range(0, nz, 64):
        sl = slice(iz0, min(nz, iz0 + 64))
        r = torch.sqrt((x - 0.5 * nx) ** 2 + (y - 0.4 * ny) ** 2 + (z[sl] - 0.5 * nzg) ** 2)
        vol[sl] -= 400.0 * torch.exp(-((r - 0.3 * min(nx, ny)) ** 2) / (2 * 1.5 * 1.5))
    # blobs: sparse impulses blurred with the library's own Gaussian (sigma 3), amplitude ~ -300
    imp = torch.zeros(shape, device=device, dtype=torch.float32)
    nblobs = max(8, int(4096 * (nz * ny * nx) / 1024 ** 3))
    g.manual_seed(seed * 7919 + z_offset)
    idx = torch.randint(0, nz * ny * nx, (nblobs,), device=device, generator=g)
    imp.view(-1)[idx] = -300.0 * (2 * math.pi * 9.0) ** 1.5
    blur = torch.empty_like(imp)
    pipeline.gauss(ctx, imp, blur, 3.0)
    vol += blur
    del imp, blur
    return vol


def cpu_baseline(sample=96):
    """The same three stages on a sample^3 volume with the reference's OpenMP code on the host."""
    import volgen
    from oracle import pyoracle as po
    kind = "reference" if po.available("ref") else "port"
    if kind == "port" and not po.available("oracle"):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libvisfd_oracle.so"])
    L = po.load("ref" if kind == "reference" else "oracle")
    from visfd_amd import pipeline
    src = volgen.membrane_volume((sample,) * 3, seed=7)
    ratio = L.ratio_from_threshold(0.03)
    sig = pipeline.cli_blob_sigmas(*BLOB)
    t0 = time.perf_counter()
    L.gauss_ratio(src, (GAUSS_SIGMA,) * 3, ratio)
    t1 = time.perf_counter()
    L.blob_dog(src, sig, None, None, 0.02, ratio)
    t2 = time.perf_counter()
    s = np.float32(MEMBRANE["sigma"])
    _, hess = L.calc_hessian(src, s, ratio, None, want_grad=True)
    sal, dirs = L.hessian_saliency(hess, po.ORDER_DECREASING)
    L.threshold_fraction(sal, MEMBRANE["best_fraction"])
    ten = L.tv_dense_stick(sal, dirs, float(np.float32(5.0) * s), 4, 2.0 ** 0.5)
    L.tensor_saliency(ten, po.ORDER_DECREASING, sal)
    t3 = time.perf_counter()
    nvox = sample ** 3
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    return {
        "value": round(nvox / (t3 - t0) / 1e6, 4), "unit": "Mvoxels/s", "cores": cores, "kind": kind,
        "sample": "%d^3 synthetic membrane volume, same three stages (gauss %.3fs, blob %.3fs, membrane+TV %.3fs)"
                  % (sample, t1 - t0, t2 - t1, t3 - t2),
    }


STAGE_BYTES_PER_VOXEL = {"gauss": 8.0, "blob_dog": 216.0, "membrane_tv": 112.0}


def offline_traffic(stem, shape):
    """HBM bytes per launch from PMC counters, measured offline (tools/profile_round.sh) and committed under profiles/:
    the newest round's file for this shape, or (None, None)."""
    for rnd in range(9, 0, -1):
        name = "r%02d_%s.json" % (rnd, stem)
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", name)))
            if list(tr["shape"]) == list(shape):
                return tr["traffic_bytes"], "profiles/" + name
        except Exception:
            continue
    return None, None


def pipeline_roofline(stage_ms, ms_per_step, nvox):
    total_b = sum(STAGE_BYTES_PER_VOXEL.values())
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_voxel": total_b,
           "achieved": round(total_b * nvox / (ms_per_step * 1e-3) / 1e9, 1),
           "frac": round(total_b * nvox / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "stages": {}}
    for (name, b), ms in zip(STAGE_BYTES_PER_VOXEL.items(), stage_ms):
        gbs = b * nvox / (max(float(ms), 1e-9) * 1e-3) / 1e9
        out["stages"][name] = {"bytes_per_voxel": b, "ms": round(float(ms), 3), "achieved": round(gbs, 1),
                               "frac": round(gbs / HBM_PEAK_GBS, 4)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024, help="edge of the per-GPU volume")
    ap.add_argument("--nz", type=int, default=0, help="planes per GPU (default: --size); e.g. --size 2048 --nz 512 is one "
                                                       "slab of BASELINE config 5")
    ap.add_argument("--cpu-sample", type=int, default=192)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-2048", action="store_true", help="skip the extra Gaussian timing on a 2048^3 volume")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` run directly: start the N ranks as a FRESH child (torch.distributed.run, one rank per
        # GPU over RCCL) and relay its JSON line and exit code.  This process has not touched the GPU -- nothing that has
        # may replace itself with another program on this pool, hence a child, not an exec.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        sys.stdout.write(r.stdout)
        sys.stdout.flush()
        raise SystemExit(r.returncode)

    import torch
    import torch.distributed as dist
    from visfd_amd import api, pipeline, slab

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # one GPU per rank over RCCL; a rehearsal with more ranks than GPUs (one-GPU box) shares cuda:0 and
    # stages halos through gloo -- that run checks the code path, its number means nothing
    own_gpu = world == 1 or torch.cuda.device_count() >= world
    dev_index = local_rank if own_gpu else 0
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if own_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    stream = torch.cuda.Stream(device=device)  # one real stream shared by torch and the library
    torch.cuda.set_stream(stream)
    ctx = api.Context(dev_index, stream.cuda_stream)

    S = args.size
    NZ = args.nz if args.nz > 0 else S
    nvox_rank = S * S * NZ
    sig = pipeline.cli_blob_sigmas(*BLOB)
    sigma_tv = float(np.float32(MEMBRANE["tv_sigma_ratio"]) * np.float32(MEMBRANE["sigma"]))
    h_tv = int(math.floor(np.float32(sigma_tv) * np.float32(math.sqrt(2.0))))
    layout = slab.SlabLayout(NZ * world, rank, world, ghost=max(h_tv, 12))
    shape = (layout.nz_local, S, S)

    src = torch.empty(shape, device=device, dtype=torch.float32)
    own = synth_volume(torch, ctx, (layout.z1 - layout.z0, S, S), device, seed=12345, z_offset=layout.z0,
                       nz_global=NZ * world)
    layout.owned(src).copy_(own)
    del own
    dst = torch.empty(shape, device=device, dtype=torch.float32)
    sal = torch.empty(shape, device=device, dtype=torch.float32)
    dirs = torch.empty((3,) + shape, device=device, dtype=torch.float32)
    ten = torch.empty((6,) + shape, device=device, dtype=torch.float32)
    torch.cuda.synchronize()

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    stage_ms = np.zeros(3)
    counts = {}

    def step(record):
        if record:
            ev[0].record()
        if world > 1:
            slab.exchange_halos(src, layout, 6)
        pipeline.gauss(ctx, src, dst, GAUSS_SIGMA)
        if record:
            ev[1].record()
        if world > 1:
            mins, maxs = slab.blob_detect_slab(ctx, layout, src, sig)
        else:
            mins, maxs = pipeline.blob_detect(ctx, src, sig)
        if record:
            ev[2].record()
        if world > 1:
            thr = slab.membrane_detect_slab(ctx, layout, src, sal, dirs, ten, scratch=dst, **MEMBRANE)
        else:
            thr = pipeline.membrane_detect(ctx, src, sal, dirs, ten, scratch=dst, **MEMBRANE)
        if record:
            ev[3].record()
            torch.cuda.synchronize()
            for i in range(3):
                stage_ms[i] += ev[i].elapsed_time(ev[i + 1])
        counts.update(minima=len(mins), maxima=len(maxs), threshold=float(thr))

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    value = nvox_rank * world / (ms_per_step * 1e-3) / 1e6  # Mvoxels/s, whole job

    # ---- roofline of the separable-Gaussian kernel, timed alone with HIP events ----------------
    roofline = roofline_pass = roofline_tv = None
    if rank == 0:
        reps = 10
        pipeline.gauss(ctx, src, dst, GAUSS_SIGMA)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            pipeline.gauss(ctx, src, dst, GAUSS_SIGMA)
        e1.record()
        torch.cuda.synchronize()
        g_ms = e0.elapsed_time(e1) / reps
        nv = shape[0] * shape[1] * shape[2]
        achieved = 8.0 * nv / (g_ms * 1e-3) / 1e9  # algorithmic 8 B/voxel
        # HBM bytes per launch from PMC counters: measured offline with rocprofv3 (separate --pmc passes,
        # tools/pmc_traffic.py) and committed under profiles/; only quoted for the shape it was measured on
        traffic, traffic_file = offline_traffic("gauss_traffic", shape)
        roofline = {"bound": "hbm", "kernel": "gauss_fused_kernel<H=5> (separable 3-D Gaussian, sigma=2)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_unit": "bytes per launch, measured OFFLINE (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE in separate passes, "
                                    "%s), not in this run" % traffic_file,
                    "algorithmic_bytes": 8 * nv,
                    "ms_per_launch": round(g_ms, 4), "voxels_per_launch": nv,
                    "note": "exact mul+add arithmetic (no FMA) makes this kernel VALU-bound, see DESIGN.md"}

        # one 1-D pass of the separable Gaussian (the 3-pass path used for masks, ragged widths and wide windows):
        # three launches (Z, Y, X with the normalisation), 8 B/voxel each; average launch time = total / 3
        with ctx.options(gauss_3pass=1):
            pipeline.gauss(ctx, src, dst, GAUSS_SIGMA)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                pipeline.gauss(ctx, src, dst, GAUSS_SIGMA)
            e1.record()
            torch.cuda.synchronize()
        p_ms = e0.elapsed_time(e1) / reps / 3.0
        p_ach = 8.0 * nv / (p_ms * 1e-3) / 1e9
        p_traffic, _ = offline_traffic("gauss_pass_traffic", shape)   # average of the three passes
        roofline_pass = {"bound": "hbm", "kernel": "conv_march_kernel<5> (Z, Y) / conv_row_kernel<5> (X): one 1-D pass of "
                                                   "the separable Gaussian, sigma=2",
                         "achieved": round(p_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(p_ach / HBM_PEAK_GBS, 4), "traffic": p_traffic, "algorithmic_bytes": 8 * nv,
                         "ms_per_launch": round(p_ms, 4), "voxels_per_launch": nv,
                         "note": "average of the three pass launches; a device copy on this box runs at ~5.0 TB/s"}

        # the tensor-voting kernel (80 % of the step): a VALU-bound stencil, priced against the FP32 vector peak as
        # SURVEY.md 8d asks -- 45 flop per evaluated vote (feature.hpp:2312-2377), votes = salient senders x non-zero taps
        # (boundary clipping ignored: < 4 % at this size)
        order = api.DECREASING_EIVALS
        ratio = api.ratio_from_threshold(0.03)
        ctx.ridge_saliency_dev(src, sal, dirs, MEMBRANE["sigma"], ratio, order)
        ctx.threshold_fraction_dev(sal, MEMBRANE["best_fraction"])
        n_salient = int(torch.count_nonzero(sal).item())
        _, w_tab, _ = api.tv_tables(sigma_tv, math.sqrt(2.0))
        n_taps = int(np.count_nonzero(w_tab))
        torch.cuda.synchronize()
        e0.record()
        ctx.tv_dense_stick_dev(sal, dirs, ten, sigma_tv, MEMBRANE["tv_exponent"], math.sqrt(2.0))
        e1.record()
        torch.cuda.synchronize()
        tv_ms = e0.elapsed_time(e1)
        votes = float(n_salient) * n_taps
        tv_traffic, tv_traffic_file = offline_traffic("tv_traffic", shape)
        tv_tflops = 45.0 * votes / (tv_ms * 1e-3) / 1e12
        roofline_tv = {"bound": "valu", "kernel": "tv_tiled_kernel (dense stick tensor voting, sigma_tv=8.66, h=12)",
                       "achieved": round(tv_tflops, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tv_tflops / 157.3, 4),
                       "traffic": tv_traffic, "ms_per_launch": round(tv_ms, 2), "votes_per_launch": votes,
                       "flop_per_vote": 45, "salient_senders": n_salient, "nonzero_taps": n_taps,
                       # the same work against what the VALU can issue without FMA: 32 multiplies/adds per vote, 70 T scalar
                       # lane-operations/s measured on this chip (profiles/r02_microbench_valu.txt)
                       "useful_lane_ops_per_s": round(32.0 * votes / (tv_ms * 1e-3) / 1e12, 2), "valu_issue_peak_lane_ops_per_s": 70.0,
                       "valu_frac_useful": round(32.0 * votes / (tv_ms * 1e-3) / 70e12, 4),
                       "algorithmic_bytes": 40 * nv, "hbm_achieved_gbs": round(40.0 * nv / (tv_ms * 1e-3) / 1e9, 1),
                       "hbm_frac": round(40.0 * nv / (tv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "traffic_unit": "bytes per launch, measured OFFLINE (%s), not in this run" % tv_traffic_file,
                       "note": "peak = nominal FP32 vector rate, which counts an FMA as 2 flop; the reference's operation order "
                               "forbids FMA, so the reachable rate is the 70 T scalar lane-operations/s the VALU issues "
                               "(profiles/r02_microbench_valu.txt): 32 of them per vote, DESIGN.md 4.2"}

    # ---- the north-star target case: the separable Gaussian on a 2048^3 volume (2^33 voxels, 32 GiB) -------
    roofline_2048 = None
    if rank == 0 and world == 1 and not args.no_2048:
        del dst, sal, dirs, ten
        torch.cuda.empty_cache()
        free, _ = torch.cuda.mem_get_info()
        if free > 140 * 2 ** 30:
            n2 = 2048
            big = torch.empty((n2, n2, n2), device=device, dtype=torch.float32)
            gen = torch.Generator(device=device).manual_seed(12346)
            for z in range(0, n2, 128):
                big[z:z + 128] = torch.randn((128, n2, n2), device=device, generator=gen) * 100 + 1000
            bout = torch.empty_like(big)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def timed(reps):
                pipeline.gauss(ctx, big, bout, GAUSS_SIGMA)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    pipeline.gauss(ctx, big, bout, GAUSS_SIGMA)
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / reps
            f_ms = timed(5)
            with ctx.options(gauss_3pass=1):
                p_ms = timed(3) / 3.0
            nv2 = n2 ** 3
            roofline_2048 = {
                "bound": "hbm", "workload": "separable 3-D Gaussian, sigma=2 (h=5), 2048^3 float32 (32 GiB)",
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes": 8 * nv2,
                "single_sweep": {"kernel": "gauss_fused_kernel<H=5>", "ms_per_launch": round(f_ms, 3),
                                 "achieved": round(8.0 * nv2 / (f_ms * 1e-3) / 1e9, 1),
                                 "frac": round(8.0 * nv2 / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                "one_pass": {"kernel": "conv_march_kernel<5> / conv_row_kernel<5> (average of the Z, Y, X launches)",
                             "ms_per_launch": round(p_ms, 3), "achieved": round(8.0 * nv2 / (p_ms * 1e-3) / 1e9, 1),
                             "frac": round(8.0 * nv2 / (p_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
            del big, bout

    if rank == 0:
        out = {
            "metric": "Mvoxels/s (Gauss+DoG+TV pipeline) on %s float32; %% HBM roofline" % (
                "%d^3" % S if NZ == S else "%dx%dx%d per GPU" % (S, S, NZ)),
            "value": round(value, 3), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "gauss(sigma=2) + blob-dog(12 scales, sigma 2..4) + membrane/TV(sigma=1.732, "
                                   "sigma_tv=8.66, top 5%%) on %dx%dx%d float32" % (S, S, NZ * world),
                       "per_gpu_voxels": nvox_rank, "decomposition": "z-slabs, ghost %d" % layout.ghost,
                       "halo_transport": "none" if world == 1 else ("rccl" if own_gpu else "gloo-staged (shared GPU rehearsal)")},
            "stages_ms": {"gauss": round(stage_ms[0] / args.steps, 3), "blob_dog": round(stage_ms[1] / args.steps, 3),
                          "membrane_tv": round(stage_ms[2] / args.steps, 3)},
            "results": counts,
            # the BASELINE metric's own "% HBM roofline": algorithmic bytes of SURVEY.md 8d per voxel (3-D Gaussian 8, blob
            # detection with 12 scales 8*12 + 12*10 = 216, Gauss + Hessian/eigen + select + TV + score 112) over the
            # measured stage times of the timed steps
            "roofline_pipeline": pipeline_roofline(stage_ms / args.steps, ms_per_step, nvox_rank * world),
            "roofline": roofline,
            "roofline_pass": roofline_pass,
            "roofline_tv": roofline_tv,
            "roofline_2048": roofline_2048,
        }
        if not args.no_cpu and world == 1:  # the CPU baseline is an N=1 line only
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
            except Exception as e:  # the checker libraries are optional on the GPU box
                out["cpu_baseline"] = {"value": None, "unit": "Mvoxels/s", "cores": 0, "kind": "unavailable",
                                       "sample": "failed: %s" % e}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
