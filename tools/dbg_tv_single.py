"""Development aid: one sender; which receivers differ between two builds of the library (argv[1], argv[2])."""
import sys, os, subprocess, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np

def run():
    from visfd_amd import api
    c = api.Context(0)
    shape = (20, 22, 40)
    sal = np.zeros(shape, np.float32); d = np.zeros(shape + (3,), np.float32); d[..., 2] = 1.0
    sx = int(os.environ.get("SX", "13"))
    sal[10, 11, sx] = 1.0
    d[10, 11, sx] = (0.6, 0.0, 0.8)
    with c.options(tv_fma=1):
        ten = c.tv_dense_stick(sal, d, 2.0, 4, 2.0 ** 0.5, None, None)
    np.save(sys.argv[2], ten)

if sys.argv[1] == "run":
    run()
else:
    outs = []
    for i, lib in enumerate(sys.argv[1:3]):
        env = dict(os.environ)
        if lib != "default":
            env["VISFD_HIP_LIB"] = os.path.abspath(lib)
        f = "/tmp/dbg_%d.npy" % i
        subprocess.check_call([sys.executable, __file__, "run", f], env=env)
        outs.append(np.load(f))
    a, b = outs
    nz_a = np.abs(a).sum(-1) > 0; nz_b = np.abs(b).sum(-1) > 0
    print("receivers voted: A", nz_a.sum(), "B", nz_b.sum())
    diff = np.abs(a - b).max(-1) > 1e-6 * np.abs(a).max()
    print("differing receivers", diff.sum())
    zz, yy, xx = np.nonzero(diff)
    print("x of differing:", sorted(set(xx.tolist())))
    print("y of differing:", sorted(set(yy.tolist())))
    print("z of differing:", sorted(set(zz.tolist())))
    print("x voted in A:", sorted(set(np.nonzero(nz_a)[2].tolist())), " in B:", sorted(set(np.nonzero(nz_b)[2].tolist())))
    if len(sys.argv) > 3:
        sx = int(os.environ.get("SX", "13"))
        for name, m in (("A", nz_a), ("B", nz_b)):
            zz, yy, xx = np.nonzero(m)
            print(name, sorted((int(z) - 10, int(y) - 11, int(x) - sx) for z, y, x in zip(zz, yy, xx)))
