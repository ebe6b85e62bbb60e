"""Z-slab decomposition of the hot path across the GPUs of one node.

The reference is single-process (its only parallelism is OpenMP); volumes larger than one GPU --
or simply more throughput -- are handled here by cutting the volume into slabs of whole XY planes
along Z, one process and one GPU per slab:

  * every rank stores its owned planes [z0, z1) plus up to `ghost` ghost planes on each INTERIOR
    face (none at the two true faces of the volume), so that every "outside the image" rule of the
    reference (zero extension, filter1d.hpp:98-99; Hessian clamp, visfd_utils.hpp:597-610; blob
    border rule, feature.hpp:245-252) fires only at the true faces;
  * halos are contiguous blocks of XY planes and travel point-to-point between Z-neighbours only
    (torch.distributed isend/irecv = RCCL send/recv over one xGMI link per neighbour) -- there is
    no bulk collective on voxel data;
  * the only collectives are tiny: three all-reduces of a 2048-bin histogram for the exact global
    top-fraction threshold (handlers.cpp:1751-1797) and an all-gather of blob lists.

Stage kernels are reached through an `ops` object (visfd_amd.api.Context on a GPU); the arithmetic of
an owned plane never depends on the decomposition, so slab results equal single-volume results
bit-for-bit (tests/test_slab_*.py).

STREAMS.  The functions here interleave library kernels (on the Context's HIP stream) with torch operations and
torch.distributed collectives (on torch's CURRENT stream): the two must be the same stream, i.e. build the context as
`api.Context(dev, torch.cuda.current_stream().cuda_stream)` (bench.py does).  Every entry point checks this and raises
otherwise -- a context on its own stream would race with the halo copies silently.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import api


class SlabLayout:
    """Which planes a rank owns and stores."""

    def __init__(self, nz_global, rank, world, ghost):
        self.nz_global, self.rank, self.world, self.ghost = int(nz_global), int(rank), int(world), int(ghost)
        base, rem = divmod(self.nz_global, self.world)
        self.z0 = rank * base + min(rank, rem)
        self.z1 = self.z0 + base + (1 if rank < rem else 0)
        self.lo = max(0, self.z0 - ghost)           # first stored plane (global index)
        self.hi = min(self.nz_global, self.z1 + ghost)
        self.nz_local = self.hi - self.lo
        self.own0 = self.z0 - self.lo               # owned planes inside the local array
        self.own1 = self.z1 - self.lo
        if world > 1 and (self.z1 - self.z0) < ghost:
            raise ValueError("slabs thinner than the ghost depth are not supported")

    def owned(self, t):
        """View of the owned planes of a local [nz_local, ny, nx] (or [C, nz_local, ny, nx]) tensor."""
        return t[..., self.own0:self.own1, :, :]


def make_slab(ctx, rank, world, nz_global, ghost, group=None):
    """The library's own slab handle (api.Slab -> csrc/slab.hip: halo exchange, global threshold and overlapped voting in
    C++ behind the C ABI).  Over an RCCL process group ("nccl") the library opens its own RCCL communicator; over any
    other backend (gloo: CPU-side rehearsals, ranks sharing one GPU) the same C code runs on a torch.distributed-backed
    transport.  The handle doubles as the layout object of the functions below."""
    transport = "rccl" if (world == 1 or dist.get_backend(group) == "nccl") else "torch"
    return api.Slab(ctx, rank, world, nz_global, ghost, transport, group)


def check_stream(ops, t):
    """The library's stream must be torch's current stream on t's device (see the module docstring)."""
    if t.is_cuda and hasattr(ops, "stream_handle"):
        cur = torch.cuda.current_stream(t.device).cuda_stream
        if ops.stream_handle() != cur:
            raise RuntimeError("visfd_amd.slab: the Context runs on HIP stream %#x but torch's current stream is %#x.  Make a "
                               "NON-DEFAULT stream current first (st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)) and "
                               "create the context on it: api.Context(dev, st.cuda_stream).  (The default stream's handle is 0, "
                               "which the library reads as 'create a stream of my own'.)" % (ops.stream_handle(), cur))


class HaloExchange:
    """The ghost planes of several tensors ([nz_local, ny, nx] each, contiguous) within `depth` planes of the owned range,
    from the two Z-neighbours' owned planes: ALL sends and receives are posted as one batch (one RCCL group: one launch
    per neighbour link, not one per tensor) by start() and complete at wait(); work that does not touch the ghost planes
    may be queued in between and overlaps the transfer."""

    def __init__(self, tensors, layout, depth, group=None):
        self.L, self.depth, self.group = layout, int(depth), group
        self.tensors = list(tensors)
        self.reqs, self.copies = [], []

    def start(self):
        L, depth, group = self.L, self.depth, self.group
        if L.world == 1 or depth == 0:
            return self
        assert depth <= L.ghost
        up, down = L.rank + 1, L.rank - 1
        ops = []
        for t in self.tensors:
            # RCCL moves device memory directly.  gloo (CPU tests, or several ranks sharing one GPU in a rehearsal) has
            # no device-memory point-to-point, so device tensors are staged through host copies.
            staged = t.is_cuda and dist.get_backend(group) != "nccl"
            if down >= 0:
                send, recv = t[L.own0:L.own0 + depth], t[L.own0 - depth:L.own0]
                rbuf = torch.empty(recv.shape, dtype=recv.dtype) if staged else recv
                ops.append(dist.P2POp(dist.isend, send.cpu() if staged else send, down, group))
                ops.append(dist.P2POp(dist.irecv, rbuf, down, group))
                if staged:
                    self.copies.append((recv, rbuf))
            if up < L.world:
                send, recv = t[L.own1 - depth:L.own1], t[L.own1:L.own1 + depth]
                rbuf = torch.empty(recv.shape, dtype=recv.dtype) if staged else recv
                ops.append(dist.P2POp(dist.isend, send.cpu() if staged else send, up, group))
                ops.append(dist.P2POp(dist.irecv, rbuf, up, group))
                if staged:
                    self.copies.append((recv, rbuf))
        self.reqs = dist.batch_isend_irecv(ops)
        return self

    def wait(self):
        for r in self.reqs:
            r.wait()      # RCCL: torch's current stream waits for the transfer; gloo: the host does
        for dst, buf in self.copies:
            dst.copy_(buf)
        self.reqs, self.copies = [], []


def exchange_halos(t, layout, depth, group=None):
    """Fill the ghost planes of `t` (or of every tensor of a list) within `depth` planes of the owned range."""
    if isinstance(layout, api.Slab):
        return layout.exchange(t, depth)
    HaloExchange(t if isinstance(t, (list, tuple)) else [t], layout, depth, group).start().wait()


def _pick_descending(hist, k):
    seen = 0
    for b in range(len(hist) - 1, -1, -1):
        c = int(hist[b])
        if seen + c > k:
            return b, k - seen
        seen += c
    raise RuntimeError("inconsistent histogram")


def _key_to_float(key):
    key = np.uint32(key)
    u = (key & np.uint32(0x7FFFFFFF)) if (key & np.uint32(0x80000000)) else ~key
    return float(np.array([u], np.uint32).view(np.float32)[0])


def distributed_threshold_fraction(ops, sal_owned, fraction, layout, mask_owned=None, group=None):
    """Exact global k-th largest saliency over all ranks' owned voxels, then zero everything below
    it (handlers.cpp:1751-1797).  Three radix rounds; each all-reduces 2048 counters."""
    check_stream(ops, sal_owned)
    prefix, k, thr_key = 0, None, 0
    shifts = (21, 10, 0)
    # over RCCL the histogram never leaves the devices until it is summed: the kernel writes it to device memory, the
    # all-reduce runs there, and one 16 KB copy per round brings the sum to the host that picks the digit
    on_dev = (layout.world > 1 and sal_owned.is_cuda and hasattr(ops, "select_histogram_todev")
              and dist.get_backend(group) == "nccl")
    hdev = torch.empty(2048, dtype=torch.int64, device=sal_owned.device) if on_dev else None
    for rnd in range(3):
        if on_dev:
            ops.select_histogram_todev(sal_owned, rnd, prefix, hdev, mask_owned)
            dist.all_reduce(hdev, op=dist.ReduceOp.SUM, group=group)
            h = hdev.cpu().numpy()
        else:
            hist, _ = ops.select_histogram_dev(sal_owned, rnd, prefix, mask_owned)
            h = torch.from_numpy(hist.astype(np.int64))
            if layout.world > 1:
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            h = h.numpy()
        if rnd == 0:
            n = int(h.sum())
            k = int(math.floor(np.float32(n) * np.float32(fraction)))  # size_t -> float product
            if n == 0 or k >= n:
                raise ValueError("threshold fraction selects no voxel")
        digit, k = _pick_descending(h, k)
        thr_key |= digit << shifts[rnd]
        prefix = (prefix << 11) | digit if rnd < 2 else prefix
    return _key_to_float(thr_key)


def membrane_detect_slab(ops, layout, src, sal, dirs, tensor, sigma, tv_sigma_ratio, tv_exponent=4,
                         best_fraction=0.05, truncate_threshold=0.03, tv_truncate_ratio=math.sqrt(2.0),
                         minima=True, group=None, scratch=None, src_halo_ready=False):
    """HandleTV (handlers.cpp:1501-1892) on one slab.  `src` holds the owned planes (ghosts are
    filled here); all tensors have the local shape [nz_local, ny, nx] (dirs: [3, ...], tensor: [6, ...]).
    Valid results are the owned planes of `sal` (post-voting saliency) and `tensor`."""
    L = layout
    order = api.DECREASING_EIVALS if minima else api.INCREASING_EIVALS
    ratio = api.ratio_from_threshold(truncate_threshold)
    h_gauss = int(math.floor(np.float32(sigma) * np.float32(ratio)))
    sigma_tv = float(np.float32(tv_sigma_ratio) * np.float32(sigma))
    if isinstance(L, api.Slab):   # the C-ABI path: everything below happens in csrc/slab.hip
        check_stream(L.ctx, src)   # (the library queues on its context's stream: it must be torch's current one)
        smoothed = scratch if scratch is not None else src.new_empty(src.shape)
        return L.membrane_detect(src, sal, dirs, tensor, smoothed, sigma, ratio, order, best_fraction, sigma_tv, tv_exponent,
                                 tv_truncate_ratio, src_halo_ready)
    h_tv = int(math.floor(np.float32(sigma_tv) * np.float32(tv_truncate_ratio)))
    assert L.world == 1 or (h_gauss + 1 <= L.ghost and h_tv <= L.ghost), "ghost depth too small"
    check_stream(ops, src)
    # 1. source halo deep enough for smoothing + the 19-point stencil (src_halo_ready: the caller has already exchanged
    #    the source's ghost planes at least that deep in this step)
    if not src_halo_ready:
        exchange_halos(src, L, min(L.ghost, h_gauss + 1), group)
    # 2. saliency/direction on every stored plane; planes closer than h_gauss+1 to an interior
    #    array end are garbage, owned planes are exact
    #    (scores first; directions only for the voxels that survive the threshold -- `scratch`: a tensor of the
    #    local shape for the smoothed image, allocated here when the caller has none to lend)
    two_step = hasattr(ops, "ridge_scores_dev")
    if two_step:
        smoothed = scratch if scratch is not None else src.new_empty(src.shape)
        ops.ridge_scores_dev(src, sal, smoothed, sigma, ratio, order)
    else:
        ops.ridge_saliency_dev(src, sal, dirs, sigma, ratio, order)
    # 3. global top-fraction threshold over owned voxels
    thr = distributed_threshold_fraction(ops, L.owned(sal), best_fraction, L, None, group)
    ops.apply_threshold_dev(L.owned(sal), thr)
    if two_step:
        # ghost planes keep their unthresholded scores here: their directions are overwritten by the halo exchange
        # below, and planes beyond the voting halo are zeroed before voting
        ops.ridge_directions_dev(smoothed, sal, dirs, sigma, order)
    # 4. (saliency, direction) halo for the voting window: the four channels in ONE batch.  Receiver planes at least h_tv
    #    away from both ends of the owned range see owned sender planes only, so their votes are queued while the halo is
    #    in flight; the two bands next to the ends follow once it has arrived.
    #    Stored planes beyond the exchanged halo must not vote: zeroed first (they are not part of the transfer).
    if L.own0 - h_tv > 0:
        sal[:L.own0 - h_tv].zero_()
    if L.own1 + h_tv < L.nz_local:
        sal[L.own1 + h_tv:].zero_()
    halo = HaloExchange([sal, dirs[0], dirs[1], dirs[2]], L, min(L.ghost, h_tv), group).start()
    vote = lambda z0, z1: ops.tv_dense_stick_dev(sal, dirs, tensor, sigma_tv, tv_exponent, tv_truncate_ratio, None, None,
                                                 False, (z0, z1))
    lo_band = L.own0 + (h_tv if L.rank > 0 else 0)              # first receiver plane that needs no ghost plane
    hi_band = L.own1 - (h_tv if L.rank < L.world - 1 else 0)
    if L.world > 1 and hi_band > lo_band:
        vote(lo_band, hi_band)                                  # 5a. interior, overlapping the transfer
        halo.wait()
        if lo_band > L.own0:
            vote(L.own0, lo_band)                               # 5b. the bands that read ghost planes
        if hi_band < L.own1:
            vote(hi_band, L.own1)
    else:
        halo.wait()
        vote(L.own0, L.own1)
    # 6. score
    ops.tensor_saliency_dev(tensor, sal, order)
    return thr


def _all_gather_rows(like, arrays, group=None):
    """Concatenate, on every rank, the ranks' float32 row lists (one all-gather of the row counts, one of a
    padded row block; device tensors over RCCL, host tensors over gloo -- no pickling of multi-megabyte lists)."""
    world = dist.get_world_size(group)
    on_dev = like.is_cuda and dist.get_backend(group) == "nccl"
    dev = like.device if on_dev else torch.device("cpu")
    ncol = arrays[0].shape[1]
    counts = torch.tensor([a.shape[0] for a in arrays], dtype=torch.int64, device=dev)
    all_counts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    all_counts = [c.cpu().numpy() for c in all_counts]
    rows_max = max(int(c.sum()) for c in all_counts)
    block = torch.zeros((max(rows_max, 1), ncol), dtype=torch.float32, device=dev)
    mine = np.concatenate([np.ascontiguousarray(a, np.float32).reshape(-1, ncol) for a in arrays], 0)
    if mine.shape[0]:
        block[:mine.shape[0]] = torch.from_numpy(mine).to(dev)
    blocks = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(blocks, block, group=group)
    blocks = [b.cpu().numpy() for b in blocks]
    out = []
    for i in range(len(arrays)):
        parts = []
        for r in range(world):
            start = int(all_counts[r][:i].sum())
            parts.append(blocks[r][start:start + int(all_counts[r][i])])
        out.append(np.concatenate(parts, 0))
    return out


def blob_detect_slab(ops, layout, src, sigmas, truncate_threshold=0.03, delta=0.02, minima_threshold=np.inf,
                     maxima_threshold=-np.inf, use_ratios=False, group=None, cap=1 << 22, src_halo_ready=False):
    """BlobDog (feature.hpp:53-427) on one slab: LoG volumes are computed on the stored planes, the
    4-D non-max scan keeps candidates of owned planes only, lists are merged on every rank."""
    L = layout
    ratio = api.ratio_from_threshold(truncate_threshold)
    smax = float(np.max(sigmas)) * (1.0 + 0.5 * delta)
    depth = int(math.floor(ratio * smax)) + 1
    assert L.world == 1 or depth <= L.ghost, "ghost depth too small for the widest LoG"
    if isinstance(L, api.Slab) and not use_ratios:   # the C-ABI path (absolute thresholds prune inside the scan)
        check_stream(L.ctx, src)
        mins, maxs = L.blob_dog(src, sigmas, delta, ratio, minima_threshold, maxima_threshold, src_halo_ready, cap)
        if L.world > 1:
            mins, maxs = _all_gather_rows(src, (mins, maxs), group)
        return mins, maxs
    check_stream(ops, src)
    if not src_halo_ready:
        exchange_halos(src, L, min(L.ghost, depth), group)
    # absolute thresholds prune inside the scan (strict, feature.hpp:270-291): only survivors are sorted and copied to
    # the host; ratio thresholds need the global best score first, so they are applied after the merge
    if use_ratios:
        mins, maxs = ops.blob_dog_dev(src, sigmas, None, None, delta, ratio, np.inf, -np.inf, False, cap)
    else:
        mins, maxs = ops.blob_dog_dev(src, sigmas, None, None, delta, ratio, minima_threshold, maxima_threshold, False, cap)

    def own(rows):
        keep = (rows[:, 2] >= L.own0) & (rows[:, 2] < L.own1)
        rows = rows[keep].copy()
        rows[:, 2] += np.float32(L.lo)
        return rows

    mins, maxs = own(mins), own(maxs)
    if L.world > 1:
        mins, maxs = _all_gather_rows(src, (mins, maxs), group)
    inf = np.float32(np.inf)
    tmin, tmax = np.float32(minima_threshold), np.float32(maxima_threshold)
    if use_ratios:
        # feature.hpp:286-289: with maxima_threshold = -inf the reference never records a maximum in ratio mode
        if tmax == -inf:
            maxs = maxs[:0]
        if tmin != inf or tmax != -inf:   # feature.hpp:362-417 with the GLOBAL best scores, multiplied unconditionally
            gmin = np.float32(min([1.0] + list(mins[:, 4])))
            gmax = np.float32(max([-1.0] + list(maxs[:, 4])))
            with np.errstate(invalid="ignore"):
                mins = mins[mins[:, 4] <= np.float32(tmin * gmin)]
                maxs = maxs[maxs[:, 4] >= np.float32(tmax * gmax)]
    return mins, maxs
