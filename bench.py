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

MODES.  BASELINE.json's north_star asks for bit-exact indices and for float voxel values within 1e-5 relative.  The
headline step therefore runs the library's TOLERANCE modes where the output is a float field -- tensor voting
(option tv_fma: fused multiply-adds, sub-patches with two sender streams, hit lists: csrc/tv_box.hip) and the plain Gaussian of stage 1 (gauss_fma) -- and
the exact kernels wherever an index depends on the bits (every LoG of the blob stage, the non-max scan, the radix
select).  `--mode exact` times the all-exact pipeline instead; the line always carries both (`modes`), and
tests/test_tolerance_modes.py + tests/test_full_size.py hold the tolerance kernels to 1e-5 of the field's scale.

Prints ONE JSON line on rank 0.  `roofline` describes the DOMINANT kernel of the step (tensor voting, ~70 % of it: a
VALU-bound stencil priced against the FP32 vector peak); `roofline_gauss` the separable-Gaussian kernel
BASELINE.json's HBM target names; both are timed alone with HIP events on the stream they run on.  `copy_gbs` is a
device-to-device copy measured in the same run: every HBM-bound object also states its fraction of that.  `stages_ms`
gives the per-stage split of the step; `cpu_baseline` is the real reference (or the CPU restatement when oracle/_ref is
absent) timed on this box's host cores on a sample of the same synthetic workload.
"""
import argparse
import gc
import json
import math
import os
import sys
import time

import numpy as np

CPU_SHARE_THREADS = 16   # the box's host-thread share per GPU (gpurun: "16 for one GPU")

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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(src, sample):
    """The same three stages with the reference's OpenMP code on the host, on `src`: a sample^3 volume from the bench's own
    generator (same noise statistics, blob density per voxel and membrane geometry as the GPU workload).  Runs in a process
    of its own whose environment fixed the OpenMP team before the runtime started (cpu_baseline_child)."""
    from oracle import pyoracle as po
    kind = "reference" if po.available("ref") else "port"
    if kind == "port" and not po.available("oracle"):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libvisfd_oracle.so"])
    L = po.load("ref" if kind == "reference" else "oracle")
    from visfd_amd import pipeline
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
        "cpu_model": cpu_model(), "host_cpus": os.cpu_count(),
        "omp": {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_PLACES")},
        "sample": "%d^3 volume from the bench generator (noise + blobs + membranes, seed 12345; the host's random stream), the same "
                  "three stages with the %s (gauss %.3fs, blob %.3fs, membrane+TV %.3fs)"
                  % (sample, "reference's own templates (oracle/_ref)" if kind == "reference" else "CPU restatement (oracle/)",
                     t1 - t0, t2 - t1, t3 - t2),
    }


def synth_volume_host(shape, seed):
    """synth_volume on the host for the CPU baseline's sample (the baseline's processes never touch the GPU): the same
    construction -- per-plane seeded normal(1000, 100) noise, the two tilted planes and the shell, 4096 blobs per 1024^3 voxels
    blurred with sigma 3 -- with the host's random stream and the checker's (bit-identical) Gaussian for the blur."""
    from oracle import pyoracle as po
    nz, ny, nx = shape
    rng = np.random.default_rng(seed)
    vol = (rng.standard_normal(shape, dtype=np.float32) * np.float32(100.0) + np.float32(1000.0)).astype(np.float32)
    z = np.arange(nz, dtype=np.float32).reshape(nz, 1, 1)
    y = np.arange(ny, dtype=np.float32).reshape(1, ny, 1)
    x = np.arange(nx, dtype=np.float32).reshape(1, 1, nx)
    for (a, b, c, d0) in ((0.15, -0.1, 1.0, 0.35 * nz), (1.0, 0.2, 0.1, 0.6 * nx)):
        dist = (a * x + b * y + c * z - d0) / math.sqrt(a * a + b * b + c * c)
        vol -= (400.0 * np.exp(-(dist * dist) / (2 * 1.5 * 1.5))).astype(np.float32)
    r = np.sqrt((x - 0.5 * nx) ** 2 + (y - 0.4 * ny) ** 2 + (z - 0.5 * nz) ** 2)
    vol -= (400.0 * np.exp(-((r - 0.3 * min(nx, ny)) ** 2) / (2 * 1.5 * 1.5))).astype(np.float32)
    imp = np.zeros(shape, np.float32)
    nblobs = max(8, int(4096 * (nz * ny * nx) / 1024 ** 3))
    imp.reshape(-1)[rng.integers(0, nz * ny * nx, nblobs)] = -300.0 * (2 * math.pi * 9.0) ** 1.5
    L = po.load("ref" if po.available("ref") else "oracle")
    blur, _ = L.gauss_ratio(imp, (3.0, 3.0, 3.0), L.ratio_from_threshold(0.03))
    return np.ascontiguousarray(vol + blur, np.float32)


def cpu_baseline_child(argv):
    """`bench.py --cpu-child ...`: the CPU baseline off the GPU process's clock.
      --cpu-child orchestrate SAMPLE   make the sample, then one worker per OpenMP team -- every host thread (nproc), 64 and
                                       the per-GPU share of 16 -- and print one JSON line: the best team's result, all teams listed
      --cpu-child run FILE SAMPLE                   time the three stages on FILE with the team the environment names
    A worker's environment (OMP_NUM_THREADS, OMP_PROC_BIND=close, OMP_PLACES=cores) is set by the orchestrator BEFORE the
    worker starts: libgomp reads it once."""
    import subprocess
    import tempfile
    if argv[0] == "run":
        print(json.dumps(cpu_baseline(np.load(argv[1]), int(argv[2]))))
        return 0
    sample = int(argv[1])
    nproc = os.cpu_count() or 1
    # (team, edge of its sample): the reference's OpenMP loops get SLOWER beyond a few dozen threads (measured on the 256-thread
    # GPU box: 0.60 Mvoxels/s with 16, 0.42 with 64, 0.05 with 256), so the large teams get smaller samples to keep the
    # whole leg at 15-20 s
    # (the team of every host thread costs ~25-35 s whatever its sample -- 34 s at 72^3, 24 s at 36^3 on a 256-thread box: thousands
    #  of parallel regions, each a fork and join of 256 threads -- so shrinking its sample further buys nothing)
    teams = sorted({(nproc, max(32, sample // 2)), (min(64, nproc), max(32, sample * 8 // 9)), (min(CPU_SHARE_THREADS, nproc), sample)},
                   reverse=True)
    if nproc <= CPU_SHARE_THREADS:
        teams = [(nproc, sample)]
    runs = []
    with tempfile.TemporaryDirectory(prefix="visfd_bench_") as d:
        for threads, edge in teams:
            f = os.path.join(d, "sample_%d.npy" % edge)
            if not os.path.exists(f):
                np.save(f, synth_volume_host((edge, edge, edge), 12345))
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_PLACES="cores")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child", "run", f, str(edge)], env=env,
                               stdout=subprocess.PIPE, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            runs.append(json.loads(line[-1]) if (r.returncode == 0 and line) else
                        {"value": None, "cores": threads, "kind": "unavailable", "sample": "worker failed (rc %d)" % r.returncode})
    # the headline is the box's BEST team (the reference's OpenMP loops do not scale to every thread of a 256-thread host);
    # every team that ran is listed beside it: all host threads, 64, and the per-GPU share
    good = [r for r in runs if r.get("value")]
    out = dict(max(good, key=lambda r: r["value"])) if good else dict(runs[0])
    out["teams"] = [{"cores": r["cores"], "value": r.get("value"), "omp": r.get("omp"), "sample_edge": e} for r, (_, e) in zip(runs, teams)]
    print(json.dumps(out))
    return 0


STAGE_BYTES_PER_VOXEL = {"gauss": 8.0, "blob_dog": 216.0, "membrane_tv": 112.0}


def bind_to_gpu_numa_node(torch, dev_index):
    """One process per GPU, its host thread on the GPU's own socket: the box has two NUMA nodes and eight GPUs, and a host thread
    that lands on the other socket stages its device-to-host copies and sorts its blob lists through remote memory (the blob
    stage's host tail then took 25 ms instead of 4).  Returns the node bound to, or None (no such information: nothing done)."""
    try:
        p = torch.cuda.get_device_properties(dev_index)
        bus = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


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
    """nvox: voxels of ONE rank (its stage times against one GPU's HBM peak)."""
    total_b = sum(STAGE_BYTES_PER_VOXEL.values())
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "per": "GPU (this rank's voxels and stage times)",
           "algorithmic_bytes_per_voxel": total_b,
           "achieved": round(total_b * nvox / (ms_per_step * 1e-3) / 1e9, 1),
           "frac": round(total_b * nvox / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "stages": {}}
    for (name, b), ms in zip(STAGE_BYTES_PER_VOXEL.items(), stage_ms):
        gbs = b * nvox / (max(float(ms), 1e-9) * 1e-3) / 1e9
        out["stages"][name] = {"bytes_per_voxel": b, "ms": round(float(ms), 3), "achieved": round(gbs, 1),
                               "frac": round(gbs / HBM_PEAK_GBS, 4)}
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-child":
        raise SystemExit(cpu_baseline_child(sys.argv[2:]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=1024, help="edge of the per-GPU volume")
    ap.add_argument("--nz", type=int, default=0, help="planes per GPU (default: --size); e.g. --size 2048 --nz 512 is one "
                                                       "slab of BASELINE config 5")
    ap.add_argument("--cpu-sample", type=int, default=144, help="edge of the CPU baseline's sample volume (about 10-20 s of CPU "
                                                                "work over its three OpenMP teams)")
    ap.add_argument("--mode", choices=("tolerance", "exact"), default="tolerance",
                    help="headline mode: tolerance = FMA tensor voting + FMA plain Gaussian (1e-5 contract), exact = bit-exact kernels only")
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

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # The CPU baseline (N = 1 only) runs in processes of its own, started BEFORE this one initialises the GPU: it needs its
    # OpenMP team fixed in its environment.  It runs beside this process's set-up (imports, context, the synthetic volume)
    # and is WAITED FOR before the first step: a 256-thread OpenMP job beside the timed steps starves the blob stage's
    # host side (measured: 125 -> 600 ms).
    cpu_proc = None
    if rank == 0 and world == 1 and not args.no_cpu:
        import subprocess
        cpu_proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-child", "orchestrate", str(args.cpu_sample)],
                                    stdout=subprocess.PIPE, text=True)

    import torch
    import torch.distributed as dist
    from visfd_amd import api, pipeline, slab

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # one GPU per rank over RCCL; a rehearsal with more ranks than GPUs (one-GPU box) shares cuda:0 and
    # stages halos through gloo -- that run checks the code path, its number means nothing
    own_gpu = world == 1 or torch.cuda.device_count() >= world
    dev_index = local_rank if own_gpu else 0
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    host_numa_node = bind_to_gpu_numa_node(torch, dev_index)
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
    # world > 1: the library's own slab handle (csrc/slab.hip behind the C ABI: RCCL halos on a transfer stream, overlapped
    # interior votes, device-side histogram all-reduce); it doubles as the layout object
    halo_transport = "none"
    if world > 1:
        # every rank must end up on the same path: agree on whether the library's communicator came up everywhere; if not,
        # the slab stages run on visfd_amd/slab.py's orchestration (torch.distributed point-to-point on device memory)
        try:
            layout = slab.make_slab(ctx, rank, world, NZ * world, max(h_tv, 12))
            up = 1
        except Exception as e:   # noqa: BLE001
            sys.stderr.write("bench: library-owned slab transport unavailable on rank %d (%r)\n" % (rank, e))
            layout, up = None, 0
        flag = torch.tensor([up], device=device if own_gpu else "cpu", dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()):
            halo_transport = ("rccl (library-owned communicator, csrc/slab.hip)" if own_gpu
                              else "gloo-staged callbacks (shared GPU rehearsal)")
        else:
            if layout is not None:
                layout.close()
            layout = slab.SlabLayout(NZ * world, rank, world, ghost=max(h_tv, 12))
            halo_transport = "torch.distributed point-to-point (visfd_amd/slab.py)"
    else:
        layout = slab.SlabLayout(NZ, 0, 1, ghost=max(h_tv, 12))
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

    cpu = None
    if cpu_proc is not None:
        try:
            so, _ = cpu_proc.communicate(timeout=600)
            line = [ln for ln in so.splitlines() if ln.startswith("{")]
            cpu = json.loads(line[-1])
        except Exception as e:  # the checker libraries are optional on the GPU box
            cpu_proc.kill()
            cpu = {"value": None, "unit": "Mvoxels/s", "cores": 0, "kind": "unavailable", "sample": "failed: %s" % e}

    MODE_OPTS = {"tolerance": dict(tv_fma=1, gauss_fma=1, eig_f32=1), "exact": dict(tv_fma=0, gauss_fma=0, eig_f32=0)}
    ratio = api.ratio_from_threshold(0.03)
    # ONE source-halo exchange per step, at the deepest depth any stage needs (the widest LoG window + the 3x3x3 scan;
    # the Gaussian and the ridge stage need less): the stage functions are told the ghost planes are current
    src_depth = min(layout.ghost, int(math.floor(ratio * float(np.max(sig)) * 1.01)) + 1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    counts = {}

    def step(record, stage_ms):
        if record:
            ev[0].record()
        if world > 1:
            slab.exchange_halos(src, layout, src_depth)
        pipeline.gauss(ctx, src, dst, GAUSS_SIGMA)
        if record:
            ev[1].record()
        job = None
        if world > 1:
            mins, maxs = slab.blob_detect_slab(ctx, layout, src, sig, src_halo_ready=True)
        else:
            # blob detection in two halves (visfd_hip_blob_dog_begin_dev / _end): its last lists are fetched, merged and converted
            # while the device already runs the membrane stage -- the device goes from the last scan straight into the ridge
            # kernels (an idle MI355X took 5-25 ms to start the next kernel after the 6-9 ms of host work that used to sit here)
            job = pipeline.blob_detect_begin(ctx, src, sig)
        if record:
            ev[2].record()
        if world > 1:
            thr = slab.membrane_detect_slab(ctx, layout, src, sal, dirs, ten, scratch=dst, src_halo_ready=True, **MEMBRANE)
        else:
            thr = pipeline.membrane_detect(ctx, src, sal, dirs, ten, scratch=dst, **MEMBRANE)
        if record:
            ev[3].record()
        if job is not None:
            mins, maxs = pipeline.blob_detect_end(ctx, job)
        if record:
            torch.cuda.synchronize()
            for i in range(3):
                stage_ms[i] += ev[i].elapsed_time(ev[i + 1])
        counts.update(minima=len(mins), maxima=len(maxs), threshold=float(thr))

    def run_mode(mode, steps, warmup):
        """W untimed + exactly K timed steps, barrier + synchronize on both sides, MAX over ranks."""
        stage_ms = np.zeros(3)
        with ctx.options(**MODE_OPTS[mode]):
            for _ in range(warmup):
                step(False, stage_ms)
            torch.cuda.synchronize()
            # the interpreter's cyclic garbage collector stays out of the timed steps (as timeit does it): with torch loaded a
            # full collection takes 20-30 ms and showed up as 117 vs 146 ms blob stages from one run to the next
            # (tools/blob_tail_time.py); everything the steps allocate is freed by reference counts
            gc.collect()
            gc.disable()
            try:
                if world > 1:
                    dist.barrier()
                t0 = time.perf_counter()
                for _ in range(steps):
                    step(True, stage_ms)
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                elapsed = time.perf_counter() - t0
            finally:
                gc.enable()
        if world > 1:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        ms = elapsed * 1e3 / steps
        return {"ms_per_step": round(ms, 3), "value": round(nvox_rank * world / (ms * 1e-3) / 1e6, 3), "steps": steps,
                "stages_ms": {"gauss": round(stage_ms[0] / steps, 3), "blob_dog": round(stage_ms[1] / steps, 3),
                              "membrane_tv": round(stage_ms[2] / steps, 3)},
                "results": dict(counts)}, stage_ms / steps

    headline, stage_avg = run_mode(args.mode, args.steps, args.warmup)
    other_mode = "exact" if args.mode == "tolerance" else "tolerance"
    other, _ = run_mode(other_mode, 2, 1)      # the other mode, briefly, for the record
    ms_per_step, value = headline["ms_per_step"], headline["value"]

    # ---- multi-GPU: a seam check, so that the first run with one GPU per rank validates bits as well as speed ------
    # Every rank > 0 regenerates the top planes of its lower neighbour's slab (the generator is a pure function of the
    # slab's position), joins them to its own bottom planes into a small volume that has the seam in its middle, runs the
    # single-volume ridge + voting stages on it with the GLOBAL threshold the slab run selected, and compares the planes
    # just above the seam with what the slab run left in `ten` (bit for bit in exact mode, 1e-5 in tolerance mode).
    slab_check = None
    if world > 1:
        ok, nchk = 1, 0
        D = 40
        thr = counts["threshold"]
        order = api.DECREASING_EIVALS
        with ctx.options(**MODE_OPTS[args.mode]):
            step(False, np.zeros(3))           # `ten` of the headline mode
            if rank > 0 and layout.z1 - layout.z0 >= D:
                prev = slab.SlabLayout(NZ * world, rank - 1, world, ghost=layout.ghost)
                below = synth_volume(torch, ctx, (prev.z1 - prev.z0, S, S), device, seed=12345, z_offset=prev.z0,
                                     nz_global=NZ * world)[-D:].clone()
                block = torch.cat([below, layout.owned(src)[:D]], 0).contiguous()
                del below
                bsal, bsm = torch.empty_like(block), torch.empty_like(block)
                bdirs = torch.zeros((3,) + tuple(block.shape), device=device)
                bten = torch.empty((6,) + tuple(block.shape), device=device)
                ctx.ridge_scores_dev(block, bsal, bsm, MEMBRANE["sigma"], ratio, order)
                ctx.apply_threshold_dev(bsal, thr)
                ctx.ridge_directions_dev(bsm, bsal, bdirs, MEMBRANE["sigma"], order)
                ctx.tv_dense_stick_dev(bsal, bdirs, bten, sigma_tv, MEMBRANE["tv_exponent"], math.sqrt(2.0))
                torch.cuda.synchronize()
                margin = h_tv + int(math.floor(np.float32(MEMBRANE["sigma"]) * np.float32(ratio))) + 2
                lo, hi = D, 2 * D - margin                       # block planes = owned planes [0, D - margin)
                got = layout.owned(ten)[:, :hi - lo]
                want = bten[:, lo:hi]
                nchk = hi - lo
                if args.mode == "exact":
                    ok = int(torch.equal(got.view(torch.int32), want.view(torch.int32)))
                else:
                    ok = int(float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()))
                ok = ok and int(float(want.abs().max()) > 0)
                del block, bsal, bsm, bdirs, bten
        flags = torch.tensor([int(ok), nchk], device=device, dtype=torch.int64)
        allf = [torch.empty_like(flags) for _ in range(world)]
        dist.all_gather(allf, flags)
        allf = [f.cpu().numpy() for f in allf]
        slab_check = {"ok": bool(all(int(f[0]) for f in allf)), "mode": args.mode,
                      "compared": "vote tensors of the %d planes above every interior slab seam vs a single-volume run on the "
                                  "regenerated planes around the seam (%s)" % (int(max(f[1] for f in allf)),
                                  "bit for bit" if args.mode == "exact" else "1e-5 of the field's scale"),
                      "seams_checked": int(sum(1 for f in allf if int(f[1]) > 0))}

    # ---- roofline objects, rank 0, each kernel timed alone with HIP events on the stream it runs on ---------------------
    roofline = roofline_gauss = roofline_pass = roofline_ridge = copy_gbs = None
    tv_objs = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps, warm=1):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def hbm_obj(kernel, ms, nv, bytes_per_voxel=8.0, **extra):
        gbs = bytes_per_voxel * nv / (ms * 1e-3) / 1e9
        o = {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_copy": round(gbs / copy_gbs, 4) if copy_gbs else None,
             "algorithmic_bytes": int(bytes_per_voxel * nv), "ms_per_launch": round(ms, 4), "voxels_per_launch": nv}
        o.update(extra)
        return o

    if rank == 0:
        nv = shape[0] * shape[1] * shape[2]
        # what this box's HBM gives a plain device-to-device copy of the same volume (read 4 + write 4 B/voxel)
        c_ms = timed(lambda: dst.copy_(src), 5)
        copy_gbs = round(8.0 * nv / (c_ms * 1e-3) / 1e9, 1)

        # the separable Gaussian (BASELINE's HBM target kernel), exact and tolerance form
        with ctx.options(gauss_fma=0):
            g_ms = timed(lambda: pipeline.gauss(ctx, src, dst, GAUSS_SIGMA), 10)
        with ctx.options(gauss_fma=1):
            gf_ms = timed(lambda: pipeline.gauss(ctx, src, dst, GAUSS_SIGMA), 10)
        traffic, traffic_file = offline_traffic("gauss_traffic", shape)
        roofline_gauss = hbm_obj("gauss_fused_kernel<H=5> (separable 3-D Gaussian in one sweep, sigma=2), exact arithmetic", g_ms, nv,
                                 traffic=traffic,
                                 traffic_unit="bytes per launch, measured OFFLINE (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE in "
                                              "separate passes, %s), not in this run" % traffic_file,
                                 note="exact mul+add arithmetic (no FMA) makes this kernel VALU-bound, see DESIGN.md",
                                 tolerance_mode=hbm_obj("gauss_fused_kernel<H=5, FMA> (option gauss_fma: fused multiply-adds, "
                                                        "reciprocal normaliser; 1e-5 contract)", gf_ms, nv,
                                                        traffic=offline_traffic("gauss_fma_traffic", shape)[0]))
        with ctx.options(gauss_3pass=1):
            p_ms = timed(lambda: pipeline.gauss(ctx, src, dst, GAUSS_SIGMA), 10) / 3.0
        p_traffic, _ = offline_traffic("gauss_pass_traffic", shape)
        roofline_pass = hbm_obj("conv_march_kernel<5> (Z, Y) / conv_row_kernel<5> (X): one 1-D pass of the separable Gaussian, "
                                "sigma=2 (average of the three launches)", p_ms, nv, traffic=p_traffic)

        # the ridge stage of the membrane detector: FP64 eigen-decompositions, priced against HBM (20 B/voxel for
        # scores + directions, 28 for the post-vote score) and against the FP64 vector peak
        order = api.DECREASING_EIVALS
        # (exact arithmetic first, then with the tolerance mode's single-precision angle: option eig_f32)
        eig_ms = {}
        for eig in (1, 0):
            with ctx.options(eig_f32=eig):
                rs_ms = timed(lambda: ctx.ridge_scores_dev(src, sal, dst, MEMBRANE["sigma"], ratio, order), 3)
                ctx.threshold_fraction_dev(sal, MEMBRANE["best_fraction"])
                rd_ms = timed(lambda: ctx.ridge_directions_dev(dst, sal, dirs, MEMBRANE["sigma"], order), 3)
            eig_ms[eig] = [rs_ms, rd_ms]
        n_salient = int(torch.count_nonzero(sal).item())
        _, w_tab, _ = api.tv_tables(sigma_tv, math.sqrt(2.0))
        n_taps = int(np.count_nonzero(w_tab))
        votes = float(n_salient) * n_taps

        # tensor voting, the dominant kernel: a VALU-bound stencil priced against the FP32 vector peak as SURVEY.md 8d asks --
        # 45 flop per evaluated vote (feature.hpp:2312-2377), votes = salient senders x non-zero taps (boundary clipping
        # ignored: < 4 % at this size)
        tv_traffic_of = {"exact": offline_traffic("tv_traffic", shape), "tolerance": offline_traffic("tv_box_traffic", shape)}
        for mode, kname, ops_per_vote, note in (
                ("exact", "tv_boxx_kernel (bit-exact: the reference's 32 multiplies/adds per vote in its order, no FMA; the tolerance kernel's "
                 "lists, box-tested hit lists and zero-padded slices, one sender stream per receiver)", 32.0,
                 "peak = nominal FP32 vector rate, which counts an FMA as 2 flop; the reference's operation order forbids FMA here, "
                 "so the reachable rate is the 70 T lane-operations/s the VALU issues (profiles/r02_microbench_valu.txt)"),
                ("tolerance", "tv_box_kernel (option tv_fma: 19 fused instructions per vote, 4x4x2 sub-patches with two sender streams per "
                 "wave, box-tested hit lists; 1e-5 contract)", 19.0, "flop counted as the reference's 45 per vote (the algorithmic work unit of SURVEY 8d)")):
            with ctx.options(tv_fma=1 if mode == "tolerance" else 0):
                tv_ms = timed(lambda: ctx.tv_dense_stick_dev(sal, dirs, ten, sigma_tv, MEMBRANE["tv_exponent"], math.sqrt(2.0)), 2)
            tfl = 45.0 * votes / (tv_ms * 1e-3) / 1e12
            tv_objs[mode] = {
                "bound": "valu", "kernel": kname + ", sigma_tv=8.66, h=12", "achieved": round(tfl, 2), "peak": 157.3,
                "unit": "TFLOP/s", "frac": round(tfl / 157.3, 4), "ms_per_launch": round(tv_ms, 2), "votes_per_launch": votes,
                "flop_per_vote": 45, "salient_senders": n_salient, "nonzero_taps": n_taps,
                "valu_instructions_per_vote": ops_per_vote,
                "useful_lane_ops_per_s": round(ops_per_vote * votes / (tv_ms * 1e-3) / 1e12, 2),
                "valu_issue_peak_lane_ops_per_s": 70.0,
                "valu_frac_useful": round(ops_per_vote * votes / (tv_ms * 1e-3) / 70e12, 4),
                "traffic": tv_traffic_of[mode][0], "algorithmic_bytes": 40 * nv,
                "hbm_achieved_gbs": round(40.0 * nv / (tv_ms * 1e-3) / 1e9, 1),
                "hbm_frac": round(40.0 * nv / (tv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic_unit": "bytes per launch, measured OFFLINE (%s), not in this run" % tv_traffic_of[mode][1], "note": note}
        roofline = dict(tv_objs[args.mode])
        roofline["share_of_step"] = round(roofline["ms_per_launch"] / ms_per_step, 3)
        for eig in (1, 0):
            with ctx.options(eig_f32=eig):
                ts_ms = timed(lambda: ctx.tensor_saliency_dev(ten, sal, order), 3)
            eig_ms[eig].append(ts_ms)
        # FP64 work of the eigen kernels, counted from the compiled code (tools/count_fp64.py): vector FP64 instructions per
        # voxel on the common path x 2 flop for FMA forms; peak 78.6 TFLOP/s FP64 vector (MI355X_MICROARCH.md)
        FP64_PEAK = 78.6

        def fp64(o, flop_per_voxel, voxels, ms):
            tf = flop_per_voxel * voxels / (ms * 1e-3) / 1e12
            o.update(fp64_flop_per_voxel=flop_per_voxel, fp64_voxels=int(voxels), achieved_fp64_tflops=round(tf, 2),
                     frac_fp64=round(tf / FP64_PEAK, 4))
            return o
        n_sal = int(round(nv * MEMBRANE["best_fraction"]))
        roofline_ridge = {
            "bound": "fp64 valu", "peak_tflops_fp64_vector": FP64_PEAK,
            "ridge_score_kernel": fp64(hbm_obj("Gaussian-smoothed Hessian -> eigenvalues -> planar score, every voxel", rs_ms, nv, 8.0),
                                       153, nv, rs_ms),
            "ridge_directions_kernel": fp64(hbm_obj("eigenvectors of the voxels above the threshold (5 %)", rd_ms, nv, 16.0),
                                            987, n_sal, rd_ms),
            "tensor_saliency_kernel": fp64(hbm_obj("eigenvalues of the vote tensor -> lambda0 - lambda1", ts_ms, nv, 28.0),
                                           146, nv, ts_ms),
            "ms_per_launch_with_eig_f32": {"note": "the same three kernels with the tolerance mode's option eig_f32 = 1 (the solver's "
                                           "angle in single precision); the objects above are the exact arithmetic",
                                           "ridge_score_kernel": round(eig_ms[1][0], 4), "ridge_directions_kernel": round(eig_ms[1][1], 4),
                                           "tensor_saliency_kernel": round(eig_ms[1][2], 4)},
            "note": "fp64_flop_per_voxel: FP64 vector instructions of the compiled kernel per voxel it works on, FMA = 2 (tools/"
                    "count_fp64.py); the kernels issue 2-3 other vector instructions per FP64 one (fp32 Hessian stencil, the angle "
                    "in single precision, selects), so frac_fp64 is what the FP64 pipe sees, not the kernels' issue load; "
                    "ms_per_launch of ridge_score_kernel includes the smoothing Gaussian (sigma 1.73, h=4) in front of it"}

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
            nv2 = n2 ** 3
            c2 = timed(lambda: bout.copy_(big), 3)
            copy2 = 8.0 * nv2 / (c2 * 1e-3) / 1e9
            with ctx.options(gauss_fma=0):
                f_ms = timed(lambda: pipeline.gauss(ctx, big, bout, GAUSS_SIGMA), 4)
            with ctx.options(gauss_fma=1):
                ff_ms = timed(lambda: pipeline.gauss(ctx, big, bout, GAUSS_SIGMA), 4)
            with ctx.options(gauss_3pass=1):
                p2_ms = timed(lambda: pipeline.gauss(ctx, big, bout, GAUSS_SIGMA), 2) / 3.0

            def o2(kernel, ms):
                gbs = 8.0 * nv2 / (ms * 1e-3) / 1e9
                return {"kernel": kernel, "ms_per_launch": round(ms, 3), "achieved": round(gbs, 1),
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_copy": round(gbs / copy2, 4)}
            roofline_2048 = {
                "bound": "hbm", "workload": "separable 3-D Gaussian, sigma=2 (h=5), 2048^3 float32 (32 GiB)",
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes": 8 * nv2, "copy_gbs": round(copy2, 1),
                "single_sweep": o2("gauss_fused_kernel<H=5, FMA> (tolerance mode, option gauss_fma: what `filter_mrc -gauss` "
                                   "output needs -- float voxels, 1e-5 contract)", ff_ms),
                "single_sweep_exact": o2("gauss_fused_kernel<H=5> (bit-exact: what LoG -> non-max indices need)", f_ms),
                "one_pass": o2("conv_march_kernel<5> / conv_row_kernel<5> (average of the Z, Y, X launches)", p2_ms)}
            del big, bout

    if rank == 0:
        out = {
            "metric": "Mvoxels/s (Gauss+DoG+TV pipeline) on %s float32; %% HBM roofline" % (
                "%d^3" % S if NZ == S else "%dx%dx%d per GPU" % (S, S, NZ)),
            "value": value, "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "gauss(sigma=2) + blob-dog(12 scales, sigma 2..4) + membrane/TV(sigma=1.732, "
                                   "sigma_tv=8.66, top 5%%) on %dx%dx%d float32" % (S, S, NZ * world),
                       "mode": args.mode,
                       "mode_note": "tolerance: tensor voting and the stage-1 Gaussian use fused multiply-adds (float outputs, "
                                    "1e-5 of the field's scale, tests/test_tolerance_modes.py); every LoG, the non-max scan and "
                                    "the radix select stay bit-exact.  exact: bit-exact kernels everywhere",
                       "host_numa_node": host_numa_node,
                       "per_gpu_voxels": nvox_rank, "decomposition": "z-slabs, ghost %d" % layout.ghost,
                       "halo_transport": halo_transport},
            "stages_ms": headline["stages_ms"],
            "results": headline["results"],
            "modes": {args.mode: headline, other_mode: other},
            "copy_gbs": copy_gbs,
            # the BASELINE metric's own "% HBM roofline": algorithmic bytes of SURVEY.md 8d per voxel (3-D Gaussian 8, blob
            # detection with 12 scales 8*12 + 12*10 = 216, Gauss + Hessian/eigen + select + TV + score 112) over the
            # measured stage times of the timed steps, per GPU
            "roofline_pipeline": pipeline_roofline(stage_avg, ms_per_step, nvox_rank),
            "roofline": roofline,
            "roofline_tv": tv_objs,
            "roofline_gauss": roofline_gauss,
            "roofline_pass": roofline_pass,
            "roofline_ridge": roofline_ridge,
            "roofline_2048": roofline_2048,
        }
        if slab_check is not None:
            out["slab_check"] = slab_check
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        if hasattr(layout, "close"):
            layout.close()  # the slab handle (its RCCL communicator and transfer stream) before the context it points to
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    if slab_check is not None and not slab_check["ok"]:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
