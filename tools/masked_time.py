import sys, os, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from visfd_amd import api
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = api.Context(0, stream.cuda_stream)
n = 1024
g = torch.Generator(device=dev).manual_seed(1)
src = torch.empty((n, n, n), device=dev)
for z in range(0, n, 128):
    src[z:z+128] = torch.randn((128, n, n), device=dev, generator=g) * 100 + 1000
mask = (torch.rand((n, n, n), device=dev, generator=g) < 0.8).float()
dst = torch.empty_like(src)
for h in (5, 8):
    s = (h / 2.6,) * 3
    for m in (None, mask):
        for norm in (True, False):
            ctx.gauss_dev(src, dst, s, (h, h, h), m, norm); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ctx.gauss_dev(src, dst, s, (h, h, h), m, norm)
            e1.record(); torch.cuda.synchronize()
            print("h=%d masked=%s normalize=%s: %.3f ms" % (h, m is not None, norm, e0.elapsed_time(e1) / 5))
