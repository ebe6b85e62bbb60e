"""Build libvisfd_hip.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m visfd_amd.build [--force] [-j N]

Objects go to build/ (git-ignored); the shared library is written next to this file so that it
travels with the repository snapshot to the GPU box.  hipcc cross-compiles without a GPU.
-ffp-contract=off is part of the numerical contract (see csrc/gauss.hip).
"""
import concurrent.futures
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(ROOT, "build", "visfd_hip")
LIB = os.path.join(HERE, "libvisfd_hip.so")

SOURCES = ["host_math.cpp", "blob_post.cpp", "connect.cpp", "api.hip", "gauss.hip", "blob.hip", "select.hip", "tv.hip", "resample.hip", "slab.hip"]
# (source, object stem, extra flags): the fused Gaussian is compiled once per window half-width
# the vote loop is faster without the SLP vectoriser's packed-f32 shuffles (profiles/r01 notes)
TV_VARIANT = [("tv_tiled.hip", "tv_tiled", ["-fno-slp-vectorize"]), ("tv_box.hip", "tv_box", ["-fno-slp-vectorize"]),
              ("ridge.hip", "ridge", [])]
VARIANTS = TV_VARIANT + [("gauss_fused.hip", "gauss_fused_h%d" % h, ["-DVH_FUSED_H=%d" % h, "-fno-slp-vectorize"])
                         for h in range(1, 9)]
HEADERS = ["common.hpp", "eigen3.hpp", os.path.join("..", "..", "include", "visfd_hip.h")]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _compile(src, obj, extra=()):
    t0 = time.time()
    cmd = [HIPCC] + FLAGS + list(extra) + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return src, r.returncode, r.stdout + r.stderr, time.time() - t0


def build(force=False, jobs=4, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    newest_header = max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)
    todo, objs = [], []
    units = [(s, os.path.splitext(s)[0], []) for s in SOURCES] + VARIANTS
    flagfile = os.path.join(OBJDIR, "flags.txt")
    flagsig = repr((FLAGS, VARIANTS))
    if not os.path.exists(flagfile) or open(flagfile).read() != flagsig:
        force = True
    for s, stem, extra in units:
        obj = os.path.join(OBJDIR, stem + ".o")
        objs.append(obj)
        if force or _mtime(obj) < max(_mtime(os.path.join(CSRC, s)), newest_header):
            todo.append((s, obj, extra))
    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
            for (src, rc, out, dt), (_, obj, _e) in zip(ex.map(lambda a: _compile(*a), todo), todo):
                if verbose:
                    print("[visfd_amd.build] %-22s %5.1fs %s" % (os.path.basename(obj), dt, "ok" if rc == 0 else "FAILED"))
                if rc != 0:
                    sys.stderr.write(out)
                    raise RuntimeError("hipcc failed on " + src)
                elif out.strip() and verbose:
                    sys.stderr.write(out)
    with open(flagfile, "w") as f:
        f.write(flagsig)
    if todo or not os.path.exists(LIB) or _mtime(LIB) < max(_mtime(o) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print("[visfd_amd.build] linked", LIB)
    build_cli(verbose)
    return LIB


CLI_SRC = os.path.join(HERE, "cli", "filter_mrc.cpp")
CLI_BIN = os.path.join(HERE, "cli", "filter_mrc")


def build_cli(verbose=True):
    """The filter_mrc drop-in: plain C++11 host code on top of the C ABI (no HIP in this file)."""
    deps = [CLI_SRC, os.path.join(ROOT, "include", "visfd_hip.hpp"), os.path.join(ROOT, "include", "visfd_hip.h"), LIB]
    if os.path.exists(CLI_BIN) and _mtime(CLI_BIN) >= max(_mtime(d) for d in deps):
        return CLI_BIN
    cmd = ["g++", "-std=c++11", "-O2", "-Wall", "-o", CLI_BIN, CLI_SRC, "-L" + HERE, "-lvisfd_hip",
           "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("building filter_mrc failed")
    if verbose:
        print("[visfd_amd.build] built", CLI_BIN)
    return CLI_BIN


if __name__ == "__main__":
    j = 4
    if "-j" in sys.argv:
        j = int(sys.argv[sys.argv.index("-j") + 1])
    build(force="--force" in sys.argv, jobs=j)
