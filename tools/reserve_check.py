"""Does the workgroup reservation of the slab runs (visfd_hip_slab_set_reserve -> context option tv_reserve_wg) do what it is
there for?  The voting kernels are persistent grids that fill the chip; a kernel of another stream that arrives while such a
grid is resident -- RCCL's send/recv kernels on the slab's transfer stream -- can only start in slots the grid left free.

A vote (exact kernel: 8 waves per SIMD, every wave slot taken; tolerance kernel: 6) is launched on the context's stream and,
as soon as it has started, a small element-wise kernel on a second stream.  From events: when does the small kernel finish,
relative to the vote?  Prints one line per (kernel, reserve) and RESERVE-OK / RESERVE-FAIL (tests/test_00_slab_gpu.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import api  # noqa: E402

dev = torch.device("cuda:0")
main = torch.cuda.Stream()
side = torch.cuda.Stream()
torch.cuda.set_stream(main)
ctx = api.Context(0, main.cuda_stream)
shape = (256, 512, 512)
g = torch.Generator(device=dev).manual_seed(3)
sal = torch.rand(shape, device=dev, generator=g)
sal[torch.rand(shape, device=dev, generator=g) > 0.05] = 0.0
dirs = torch.randn((3,) + shape, device=dev, generator=g)
dirs /= dirs.norm(dim=0, keepdim=True)
ten = torch.empty((6,) + shape, device=dev)
small = torch.zeros(1 << 20, device=dev)
torch.cuda.synchronize()
ok = True
for mode in (0, 1):
    ctx.set_option("tv_fma", mode)
    res = {}
    for reserve in (0, 64):
        ctx.set_option("tv_reserve_wg", reserve)
        for rep in range(2):   # (the first launch of a kernel pays its code upload)
            e0, e1, s1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            started = torch.cuda.Event()
            e0.record(main)
            started.record(main)
            ctx.tv_dense_stick_dev(sal, dirs, ten, 8.660254, 4, 2 ** 0.5)
            e1.record(main)
            with torch.cuda.stream(side):
                side.wait_event(started)
                torch.cuda._sleep(2000000)   # ~1 ms: the vote is resident on the chip by the time the next kernel arrives
                small.add_(1.0)
                s1.record(side)
            torch.cuda.synchronize()
        res[reserve] = (e0.elapsed_time(s1), e0.elapsed_time(e1))
        print("tv_fma=%d tv_reserve_wg=%d: the side kernel finished %.2f ms after the vote was queued; the vote took %.2f ms" % (
            mode, reserve, res[reserve][0], res[reserve][1]), flush=True)
    # with slots reserved the side kernel must not wait for the vote
    ok = ok and res[64][0] < 0.25 * res[64][1] and res[64][0] > 0.3
    # without them the exact kernel, which owns every wave slot, makes it wait for workgroups to exit
    if mode == 0:
        ok = ok and res[0][0] > 0.5 * res[0][1]
ctx.set_option("tv_reserve_wg", 0)
ctx.close()
print("RESERVE-OK" if ok else "RESERVE-FAIL")
