"""Development aid: link a variant of libvisfd_hip.so in which ONE translation unit is recompiled
with extra flags (tuning sweeps).  The variant is written to visfd_amd/_variants/<name>.so and used
when VISFD_HIP_LIB points at it (visfd_amd/api.py).

    python tools/build_variant.py cap512 tv_tiled.hip -DVH_TV_CAP=512
    python tools/build_variant.py gx gauss_fused.hip:gauss_fused_h8,gauss_fused_h9 -DVH_FUSED_EXTRA_CFGS
(source[:stem,stem...] restricts the recompilation to some of the units built from that source)
"""
import concurrent.futures
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visfd_amd import build as B  # noqa: E402


def main():
    name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    only = None
    if ":" in src:
        src, stems = src.split(":")
        only = set(stems.split(","))
    B.build(verbose=False)
    outdir = os.path.join(ROOT, "visfd_amd", "_variants")
    os.makedirs(outdir, exist_ok=True)
    units = [(s, os.path.splitext(s)[0], []) for s in B.SOURCES] + B.VARIANTS
    objs, jobs = [], []
    for s, stem, flags in units:
        obj = os.path.join(B.OBJDIR, stem + ".o")
        if s == src and (only is None or stem in only):
            obj = os.path.join(outdir, "%s_%s.o" % (name, stem))
            jobs.append((s, obj, list(flags) + extra))
        objs.append(obj)
    with concurrent.futures.ThreadPoolExecutor(max_workers=5) as ex:
        for _, rc, out, dt in ex.map(lambda a: B._compile(*a), jobs):
            if rc:
                sys.stderr.write(out)
                raise SystemExit(1)
    lib = os.path.join(outdir, name + ".so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
