"""Development aid: link a variant of libvisfd_hip.so in which ONE translation unit is recompiled
with extra flags (tuning sweeps).  The variant is written to visfd_amd/_variants/<name>.so and used
when VISFD_HIP_LIB points at it (visfd_amd/api.py).

    python tools/build_variant.py cap512 tv_tiled.hip -DVH_TV_CAP=512
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import build as B  # noqa: E402


def main():
    name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build(verbose=False)
    outdir = os.path.join(ROOT, "visfd_amd", "_variants")
    os.makedirs(outdir, exist_ok=True)
    units = [(s, os.path.splitext(s)[0], []) for s in B.SOURCES] + B.VARIANTS
    objs = []
    for s, stem, flags in units:
        obj = os.path.join(B.OBJDIR, stem + ".o")
        if s == src:
            obj = os.path.join(outdir, "%s_%s.o" % (name, stem))
            _, rc, out, dt = B._compile(s, obj, list(flags) + extra)
            if rc:
                sys.stderr.write(out)
                raise SystemExit(1)
        objs.append(obj)
    lib = os.path.join(outdir, name + ".so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
