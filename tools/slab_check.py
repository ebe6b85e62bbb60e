"""Z-slab pipeline versus the single-volume pipeline, both on the GPU, bit for bit.

Launch with one process per rank:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29533 tools/slab_check.py
With at least as many GPUs as ranks every rank takes its own GPU and halos travel over RCCL
("nccl"); on a one-GPU box the ranks share cuda:0 and halos are staged through gloo.  Every rank
runs the slab stages; rank 0 additionally runs the whole volume and compares the gathered owned
planes and the merged blob lists.  Prints "SLAB-OK ..." on success, exits non-zero otherwise.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

import volgen  # noqa: E402
from visfd_amd import api, pipeline, slab  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    ngpu = torch.cuda.device_count()
    own_gpu = ngpu >= world
    dev_index = local if own_gpu else 0
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if own_gpu:
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    shape, sigma, tv_ratio, fraction, ghost = (48, 36, 44), 1.2, 2.0, 0.15, 6
    blob_sigmas = np.array([1.0, 1.25, 1.55, 1.9], np.float32)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = api.Context(dev_index, stream.cuda_stream)
    full = torch.from_numpy(volgen.membrane_volume(shape, seed=55))

    # the library's own slab handle: the C entry points of csrc/slab.hip (RCCL with one GPU per rank, torch-backed
    # callbacks when the ranks share a card); SLAB_PY=1 keeps the Python orchestration of visfd_amd/slab.py instead
    if os.environ.get("SLAB_PY"):
        L = slab.SlabLayout(shape[0], rank, world, ghost=ghost)
    else:
        L = slab.make_slab(ctx, rank, world, shape[0], ghost)
    lshape = (L.nz_local,) + tuple(shape[1:])
    src = torch.full(lshape, float("nan"), device=dev)      # ghosts must come from the exchange
    L.owned(src).copy_(full[L.z0:L.z1])
    sal = torch.zeros(lshape, device=dev)
    dirs = torch.zeros((3,) + lshape, device=dev)
    ten = torch.zeros((6,) + lshape, device=dev)
    if os.environ.get("SLAB_PY"):
        # Python orchestration: a context on a stream of its own would race with torch's halo copies -- must be refused
        # (the C path orders its transfer stream against the context's stream itself)
        other = api.Context(dev_index)
        try:
            slab.membrane_detect_slab(other, L, src, sal, dirs, ten, sigma, tv_ratio, 4, fraction)
            refused = False
        except RuntimeError as e:
            refused = "current stream" in str(e)
        other.close()
        if not refused:
            print("SLAB-MISMATCH: a context on its own stream was accepted", flush=True)
            sys.exit(1)
    thr = slab.membrane_detect_slab(ctx, L, src, sal, dirs, ten, sigma, tv_ratio, 4, fraction)
    host_bad = []
    if isinstance(L, api.Slab):
        # the host-memory face of the same stage (what filter_mrc -slab calls): owned planes in, owned planes out
        import math
        sal_h, ten_h, thr_h = L.membrane_detect_host(
            full[L.z0:L.z1].numpy(), sigma, api.ratio_from_threshold(0.03), api.DECREASING_EIVALS, fraction,
            float(np.float32(tv_ratio) * np.float32(sigma)), 4, math.sqrt(2.0), want_tensor=True)
        ctx.synchronize()
        if np.float32(thr_h) != np.float32(thr):
            host_bad.append("host-face threshold differs on rank %d" % rank)
        if not np.array_equal(sal_h.view(np.uint32), L.owned(sal).cpu().numpy().view(np.uint32)):
            host_bad.append("host-face saliency differs on rank %d" % rank)
        if not np.array_equal(ten_h.view(np.uint32), np.moveaxis(L.owned(ten).cpu().numpy(), 0, -1).view(np.uint32)):
            host_bad.append("host-face tensor differs on rank %d" % rank)
    src2 = torch.full(lshape, float("nan"), device=dev)
    L.owned(src2).copy_(full[L.z0:L.z1])
    mins, maxs = slab.blob_detect_slab(ctx, L, src2, blob_sigmas, 0.03, 0.02, -5.0, 5.0, False)
    ctx.synchronize()
    # a list capacity far too small on ONE rank only: that rank's retry is local (the ghost planes are in place), so the
    # other ranks are not held up and the merged lists are the same
    src3 = torch.full(lshape, float("nan"), device=dev)
    L.owned(src3).copy_(full[L.z0:L.z1])
    mins_c, maxs_c = slab.blob_detect_slab(ctx, L, src3, blob_sigmas, 0.03, 0.02, -5.0, 5.0, False, cap=(2 if rank == world - 1 else 1 << 22))
    ctx.synchronize()
    for name, a, b in (("minima", mins, mins_c), ("maxima", maxs, maxs_c)):
        if a.shape != b.shape or not np.array_equal(volgen.sort_blobs(a, True).view(np.uint32), volgen.sort_blobs(b, True).view(np.uint32)):
            host_bad.append("blob %s differ after a capacity retry on rank %d" % (name, rank))

    part = dict(z0=L.z0, z1=L.z1, thr=np.float32(thr), sal=L.owned(sal).cpu().numpy(),
                ten=L.owned(ten).cpu().numpy(), mins=mins, maxs=maxs, host_bad=host_bad)
    parts = [None] * world
    dist.all_gather_object(parts, part)

    bad = []
    if rank == 0:
        vol = full.to(dev)
        fsal = torch.zeros(shape, device=dev)
        fdirs = torch.zeros((3,) + shape, device=dev)
        ften = torch.zeros((6,) + shape, device=dev)
        fthr = pipeline.membrane_detect(ctx, vol, fsal, fdirs, ften, sigma, tv_ratio, 4, fraction)
        fmins, fmaxs = pipeline.blob_detect(ctx, vol, blob_sigmas, 0.03, 0.02, None, -5.0, 5.0, False)
        ctx.synchronize()
        fsal, ften = fsal.cpu().numpy(), ften.cpu().numpy()
        if not np.abs(ften).max() > 0:
            bad.append("vote tensor is all zero")
        for p in parts:
            bad.extend(p["host_bad"])
            z0, z1 = p["z0"], p["z1"]
            if np.float32(p["thr"]) != np.float32(fthr):
                bad.append("threshold differs on planes %d..%d" % (z0, z1))
            if not np.array_equal(p["sal"].view(np.uint32), fsal[z0:z1].view(np.uint32)):
                bad.append("saliency differs on planes %d..%d" % (z0, z1))
            if not np.array_equal(p["ten"].view(np.uint32), ften[:, z0:z1].view(np.uint32)):
                bad.append("vote tensor differs on planes %d..%d" % (z0, z1))
            for name, got, want, asc in (("minima", p["mins"], fmins, True), ("maxima", p["maxs"], fmaxs, False)):
                a, b = volgen.sort_blobs(got, asc), volgen.sort_blobs(want, asc)
                if a.shape != b.shape or not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
                    bad.append("%s differ (%d vs %d)" % (name, len(a), len(b)))
        if not bad:
            print("SLAB-OK world=%d backend=%s path=%s minima=%d maxima=%d thr=%.6g" %
                  (world, dist.get_backend(), "python" if os.environ.get("SLAB_PY") else "c-abi", len(fmins), len(fmaxs), fthr),
                  flush=True)
        else:
            print("SLAB-MISMATCH: " + "; ".join(bad), flush=True)
    flag = [bool(bad)]
    dist.broadcast_object_list(flag, src=0)
    if hasattr(L, "close"):
        L.close()
    ctx.close()
    dist.destroy_process_group()
    sys.exit(1 if flag[0] else 0)


if __name__ == "__main__":
    main()
