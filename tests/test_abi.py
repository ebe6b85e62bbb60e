"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/visfd_hip.h declares; the Python binding covers exactly that set; host-side arithmetic
entry points (taps, tables) agree with the oracle bit-for-bit.  No GPU needed."""
import ctypes
import os
import re

import numpy as np
import pytest

import volgen
from conftest import ROOT, assert_bits_equal, golden


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "visfd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(visfd_hip_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from visfd_amd import api
    if not os.path.exists(api.LIB_PATH):
        from visfd_amd import build
        build.build(verbose=False)
    return api


def test_library_exports_every_declared_symbol(lib):
    L = ctypes.CDLL(lib.LIB_PATH)
    names = _header_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(L, n), "libvisfd_hip.so does not export " + n
    assert sorted(lib.exported_symbols()) == names, "visfd_amd/api.py and include/visfd_hip.h disagree"
    assert L.visfd_hip_abi_version() == 9


def test_host_arithmetic_matches_oracle(lib, oracle):
    for s, h in volgen.TAP_CASES:
        assert_bits_equal(lib.gauss_taps(s, h), oracle.gauss_taps(s, h), "taps")
    g = golden("taps")
    for s, h in volgen.TAP_CASES:
        assert_bits_equal(lib.gauss_taps(s, h), g["s%g_h%d" % (s, h)], "taps vs golden")
    assert np.float32(lib.ratio_from_threshold(0.03)) == g["ratio"]
    assert lib.gauss_halfwidths((2, 2, 0.1), 2.6482) == (5, 5, 1)
    for s, c in ((8.66, 2 ** 0.5), (3.2, 2 ** 0.5), (2.0, 2.5)):
        h, w, r = lib.tv_tables(s, c)
        ho, wo, ro = oracle.tv_tables(s, c)
        assert h == ho
        assert_bits_equal(w, wo, "tv w")
        assert_bits_equal(r, ro, "tv rhat")
    d = np.array([5, 5.8, 6.6, 100.0], np.float32)
    assert_bits_equal(lib.diameters_to_sigmas(d), oracle.diameters_to_sigmas(d), "d2s")
    assert_bits_equal(lib.sigmas_to_diameters(d), oracle.sigmas_to_diameters(d), "s2d")


def test_no_cpu_fallback(lib):
    """Without a GPU the product must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.VisfdHipError):
        lib.Context(0)


def test_cpp_shim_compiles_and_runs_host_entry_points(lib, tmp_path):
    """include/visfd_hip.hpp under g++ -std=c++11 -Wall -Werror, linked against the library; the program exercises
    the reference-signature wrappers of the host-side rows (LabelConnected, blob list post-processing)."""
    import subprocess
    exe = str(tmp_path / "shim_host_check")
    libdir = os.path.dirname(lib.LIB_PATH)
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "shim_host_check.cpp"), "-L" + libdir, "-lvisfd_hip", "-Wl,-rpath," + libdir,
           "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "clusters 2 sizes 270 108 undefined 702" in r.stdout and "blobs kept 2" in r.stdout


@pytest.mark.parametrize("src,flags", [
    ("tv_box.hip", ["-DVH_TV_STAMPS", "-DVH_TV_COUNT"]),
    ("gauss_fused.hip", ["-DVH_FUSED_H=5", "-DVH_FUSED_STAMPS"]),
])
def test_development_builds_cross_compile(src, flags, tmp_path):
    """The only build parameters left in the kernels are the instrumentation switches named in their sources (where a
    wave's time goes; what the voting kernel tests and hits).  They are not part of the product build, so nothing else
    would notice if an edit broke them: cross-compile each for gfx950 (hipcc needs no GPU)."""
    import subprocess
    from visfd_amd import build as B
    cmd = [B.HIPCC] + B.FLAGS + ["-fno-slp-vectorize"] + flags + ["-x", "hip", "-c", os.path.join(B.CSRC, src), "-o", str(tmp_path / "o.o")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_graft_entry_expects_the_library_abi(lib):
    """__graft_entry__.build() asserts the ABI version of the library it has just built: the number written there must be
    the library's (it was left behind once when the ABI moved on)."""
    L = ctypes.CDLL(lib.LIB_PATH)
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    m = re.search(r"visfd_hip_abi_version\(\) == (\d+)", src)
    assert m and int(m.group(1)) == L.visfd_hip_abi_version()
