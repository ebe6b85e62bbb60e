"""The filter_mrc drop-in (visfd_amd/cli/filter_mrc): argument handling on CPU, and on a GPU the
reference's own command lines with their known answers."""
import os
import subprocess

import numpy as np
import pytest

import volgen
from conftest import GOLDEN, ROOT, assert_bits_equal, assert_close_rel, golden

CLI = os.path.join(ROOT, "visfd_amd", "cli", "filter_mrc")


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        from visfd_amd import build
        build.build(verbose=False)
    return CLI


def run(cli, *args):
    return subprocess.run([cli] + [str(a) for a in args], capture_output=True, text=True)


def test_cli_rejects_unknown_arguments(cli):
    r = run(cli, "-in", os.path.join(GOLDEN, "test_blob_detect.rec"), "-frobnicate", 3)
    assert r.returncode == 1 and "Unrecognized" in r.stderr
    r = run(cli, "-gauss", 2)
    assert r.returncode == 1 and "input file" in r.stderr
    r = run(cli, "-in", "/nonexistent.rec", "-gauss", 2)
    assert r.returncode == 1 and "Unable to open" in r.stderr


def test_cli_fails_loudly_without_gpu(cli):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = run(cli, "-in", os.path.join(GOLDEN, "test_blob_detect.rec"), "-out", "/tmp/_x.rec", "-gauss", 2, "-w", 1)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_cli_config1_gauss(cli, tmp_path):
    """BASELINE config 1: filter_mrc -gauss 2 -w 1 on tests/test_blob_detect.rec; the reference
    prints A = 0.00907605 and its output has min/max/mean 33.43096 / 41.48531 / 36.50518."""
    out = tmp_path / "g.rec"
    r = run(cli, "-in", os.path.join(GOLDEN, "test_blob_detect.rec"), "-out", out, "-gauss", 2, "-w", 1)
    assert r.returncode == 0, r.stderr
    assert "A = 0.00907605" in r.stderr
    v = volgen.read_mrc(str(out))
    assert_bits_equal(v, golden("gauss_blobrec")["out"], "CLI gauss output")
    r = run(cli, "-in", os.path.join(GOLDEN, "test_blob_detect.rec"), "-mask",
            os.path.join(GOLDEN, "test_blob_detect_mask.rec"), "-out", out, "-gauss", 2, "-w", 1)
    assert r.returncode == 0, r.stderr
    # library result inside the mask; outside it the program writes the "masked" brightness 0 (filter_mrc.cpp:765-776)
    got, want = volgen.read_mrc(str(out)), golden("gauss_blobrec")["out_masked"].copy()
    m = volgen.read_mrc(os.path.join(GOLDEN, "test_blob_detect_mask.rec"))
    want[m == 0] = 0.0
    assert_bits_equal(got, want, "CLI masked gauss")


@pytest.mark.gpu
def test_cli_blob_reference_command(cli, tmp_path):
    """tests/test_blob_detection.sh:21 of the reference: 11 minima, best line
    '235.2 392 313.6 177.915 -140.018'."""
    out = tmp_path / "blobs.txt"
    r = run(cli, "-w", 19.6, "-mask", os.path.join(GOLDEN, "test_blob_detect_mask.rec"), "-in",
            os.path.join(GOLDEN, "test_blob_detect.rec"), "-blob", "minima", out, 160.0, 280.0, 1.01)
    assert r.returncode == 0, r.stderr
    lines = open(out).read().strip().split("\n")
    assert len(lines) == 11
    assert lines[0] == "235.2 392 313.6 177.915 -140.018"
    g = golden("blob_rec")["minima"]
    for line, row in zip(lines, g):
        x, y, z, d, s = (float(t) for t in line.split())
        assert abs(s - row[4]) <= 1e-3 * abs(row[4])
        assert (round(x / 19.6), round(y / 19.6), round(z / 19.6)) == (int(row[0]), int(row[1]), int(row[2]))


@pytest.mark.gpu
def test_cli_membrane_save_progress(cli, tmp_path, oracle):
    """-membrane minima ... -tv ... -save-progress writes the six vote-tensor channels
    (handlers.cpp:1897-1922); compared with the oracle run of the same pipeline."""
    from oracle import pyoracle as po
    inp = os.path.join(GOLDEN, "test_image_membrane.rec")
    out = tmp_path / "m.rec"
    base = tmp_path / "prog"
    w = 19.2
    r = run(cli, "-w", w, "-in", inp, "-out", out, "-membrane", "minima", 55, "-tv", 4, "-tv-angle-exponent", 4,
            "-bin", 1, "-tv-best", 0.1, "-save-progress", base)
    assert r.returncode == 0, r.stderr
    mem = volgen.read_mrc(inp)
    sigma = np.float32(np.float32(55 / np.sqrt(3.0)) / np.float32(w))
    sigma_tv = np.float32(np.float32(np.float32(4) * np.float32(55 / np.sqrt(3.0))) / np.float32(w))
    ratio = oracle.ratio_from_threshold(0.03)
    _, hess = oracle.calc_hessian(mem, sigma, ratio, None, want_grad=False)
    sal, dirs = oracle.hessian_saliency(hess, po.ORDER_DECREASING)
    oracle.threshold_fraction(sal, 0.1)
    ten = oracle.tv_dense_stick(sal, dirs, sigma_tv, 4, 2.0 ** 0.5)
    oracle.tensor_saliency(ten, po.ORDER_DECREASING, sal)
    got = np.stack([volgen.read_mrc("%s_tensor_%d.rec" % (base, c)) for c in range(6)], -1)
    # device eigen-solver differences can move a few voxels across the saliency threshold: compare
    # on the scale of the field (1e-4) rather than bit-for-bit here; bit-exact voting with identical
    # inputs is covered by test_gpu_parity.py
    assert_close_rel(got, ten, 1e-4, "vote tensor files")
    assert_close_rel(volgen.read_mrc(str(out)), sal, 1e-4, "output saliency")


def test_cli_discard_blobs_reference_scenario(cli, tmp_path):
    """tests/test_blob_detection.sh:25-29 of the reference: the 11 minima of the -blob step, filtered with
    `-discard-blobs in out -blob-separation 1.1 -minima-threshold -90`, leave 2 blobs.  Host-side path
    (no GPU): list file in physical units -> voxels -> score cut -> mask -> greedy overlap removal."""
    g = golden("blob_rec")
    w = 19.6
    src = tmp_path / "test_blobs.txt"
    with open(src, "w") as f:
        f.write("# x y z diameter score\n")
        for row, d in zip(g["minima"], g["minima_diam_vox"]):
            f.write("%g %g %g %g %g\n" % (row[0] * w, row[1] * w, row[2] * w, d * w, row[4]))
    out = tmp_path / "test_blobs_sep_1.1_thresh_-90.txt"
    r = run(cli, "-w", w, "-mask", os.path.join(GOLDEN, "test_blob_detect_mask.rec"), "-in",
            os.path.join(GOLDEN, "test_blob_detect.rec"), "-discard-blobs", src, out, "-blob-separation", 1.1,
            "-minima-threshold", -90)
    assert r.returncode == 0, r.stderr
    lines = open(out).read().strip().split("\n")
    assert len(lines) == 2, lines
    assert lines[0] == "235.2 392 313.6 177.915 -140.018"
    assert "2 blobs remaining" in r.stderr
    # same file names twice / missing separation value are usage errors
    assert run(cli, "-in", os.path.join(GOLDEN, "test_blob_detect.rec"), "-discard-blobs", src, src).returncode == 1


@pytest.mark.gpu
def test_cli_explicit_binning(cli, tmp_path, oracle):
    """-bin 2 (filter_mrc.cpp:128-137, handlers.cpp:2361-2425): the image is averaged over 2x2x2 bins, the voxel
    width doubles, and the filter runs on the small image; the output stays small."""
    src = volgen.membrane_volume((30, 34, 44), seed=3)
    inp, out = tmp_path / "in.rec", tmp_path / "out.rec"
    volgen.write_mrc(str(inp), src, voxel_width=1.0)
    r = run(cli, "-in", inp, "-out", out, "-w", 1, "-bin", 2, "-gauss", 3)
    assert r.returncode == 0, r.stderr
    binned = oracle.bin_array3d(src, (15, 17, 22))
    want, _ = oracle.gauss_ratio(binned, (1.5,) * 3, oracle.ratio_from_threshold(0.03))
    got = volgen.read_mrc(str(out))
    assert got.shape == (15, 17, 22)
    assert_bits_equal(got, want, "CLI -bin 2 -gauss")
    hdr = np.frombuffer(open(out, "rb").read(1024), "<f4")
    assert tuple(hdr[10:13]) == (44.0, 34.0, 30.0)     # cell = binned size x doubled voxel width


@pytest.mark.gpu
def test_cli_automatic_binning_for_wide_membranes(cli, tmp_path):
    """A membrane detector whose sigma exceeds 1.8 voxels bins the image on its own (filter_mrc.cpp:140-175) and
    un-bins the result afterwards (handlers.cpp:2315-2355): the automatic run equals the explicit `-bin N` run
    un-binned, and has the input's size."""
    src = volgen.membrane_volume((40, 48, 56), seed=5)
    inp, out_auto, out_exp = tmp_path / "in.rec", tmp_path / "auto.rec", tmp_path / "exp.rec"
    volgen.write_mrc(str(inp), src, voxel_width=1.0)
    args = ["-in", inp, "-w", 1, "-membrane", "minima", 7, "-tv", 3, "-tv-angle-exponent", 4, "-tv-best", 0.15]
    r = run(cli, *args, "-out", out_auto)
    assert r.returncode == 0, r.stderr
    assert "BINNING THE IMAGE BY A FACTOR OF" in r.stderr
    n = int(r.stderr.split("BINNING THE IMAGE BY A FACTOR OF")[1].split()[0])
    assert n >= 2
    r2 = run(cli, *args, "-bin", n, "-out", out_exp)
    assert r2.returncode == 0, r2.stderr
    auto, small = volgen.read_mrc(str(out_auto)), volgen.read_mrc(str(out_exp))
    assert auto.shape == src.shape and small.shape == tuple(d // n for d in src.shape)
    iz, iy, ix = np.minimum(np.arange(40) // n, small.shape[0] - 1), np.minimum(np.arange(48) // n, small.shape[1] - 1), \
        np.minimum(np.arange(56) // n, small.shape[2] - 1)
    assert_bits_equal(auto, small[np.ix_(iz, iy, ix)], "automatic binning == explicit binning, un-binned")
    assert np.abs(small).max() > 0


@pytest.mark.gpu
def test_cli_membrane_clustering_reference_scenario(cli, tmp_path):
    """tests/test_membrane_detection.sh of the reference, both commands: detect + vote with -bin 2 and
    -save-progress, then -load-progress ... -connect 1e+09 -connect-angle 30.  Known answers of the reference
    (SURVEY.md §4): six 8x8x8 tensor files, "Number of clusters found: 1", 69 voxels with label 1."""
    inp = os.path.join(GOLDEN, "test_image_membrane.rec")
    out, base = tmp_path / "surf.rec", tmp_path / "test_image_membrane"
    common = ["-w", 19.2, "-in", inp, "-out", out, "-membrane", "minima", 55, "-tv", 4, "-tv-angle-exponent", 4, "-bin", 2]
    r = run(cli, *common, "-save-progress", base)
    assert r.returncode == 0, r.stderr
    for c in range(6):
        assert volgen.read_mrc("%s_tensor_%d.rec" % (base, c)).shape == (8, 8, 8)
    r = run(cli, *common, "-load-progress", base, "-connect", 1e9, "-connect-angle", 30, "-select-cluster", 1)
    assert r.returncode == 0, r.stderr
    assert "Number of clusters found: 1" in r.stderr
    lab = volgen.read_mrc(str(out))
    assert lab.shape == (8, 8, 8)
    assert int(((lab > 0.99) & (lab < 1.01)).sum()) == 69
    # clustering in the same run as the voting gives the same labels
    out2 = tmp_path / "surf2.rec"
    r = run(cli, *common[:4], "-out", out2, *common[6:], "-connect", 1e9, "-connect-angle", 30)
    assert r.returncode == 0, r.stderr
    assert_bits_equal(volgen.read_mrc(str(out2)), lab, "clustering after -load-progress == clustering in one run")


# ------------------------------------------------------------------------------------------ against the reference's own program
REF_CLI = os.path.join(ROOT, "oracle", "_ref", "filter_mrc_ref")


@pytest.fixture(scope="module")
def ref_cli():
    """The reference's filter_mrc, compiled by `make -C oracle ref_cli` from its sources (CPU, OpenMP)."""
    if not os.path.exists(REF_CLI):
        pytest.skip("oracle/_ref/filter_mrc_ref not built (needs /root/reference)")
    return REF_CLI


def both(cli, ref_cli, tmp_path, args, out_name="out.rec", env=None):
    """Runs the same command line through both programs in separate directories; returns the two directories.
    `env`: extra environment of this repo's program (e.g. the tolerance-mode options)."""
    dirs = []
    for tag, exe in (("mine", cli), ("ref", ref_cli)):
        d = tmp_path / tag
        d.mkdir(exist_ok=True)
        e = dict(os.environ, **env) if (env and tag == "mine") else None
        r = subprocess.run([exe] + [str(a) for a in args] + ["-out", out_name], cwd=str(d), capture_output=True, text=True, env=e)
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        dirs.append(d)
    return dirs


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [
    ["-gauss", 40, "-w", 19.6],
    ["-gauss-aniso", 30, 45, 25, "-w", 19.6],
    ["-gauss", 2, "-w", 1, "-mask", os.path.join(GOLDEN, "test_blob_detect_mask.rec")],
    ["-dog", 30, 48, "-w", 19.6],
    ["-dog-aniso", 30, 40, 25, 48, 60, 44, "-w", 19.6],
    ["-dog-aniso", 2, 3, 2.5, 3, 4, 3.5, "-w", 1, "-mask", os.path.join(GOLDEN, "test_blob_detect_mask.rec")],
    ["-log-d", 120, "-w", 19.6, "-mask", os.path.join(GOLDEN, "test_blob_detect_mask.rec")],
    ["-gauss", 60, "-w", 19.6, "-bin", 2],
    ["-fluct", 80, "-w", 19.6],
    ["-fluct", 4, "-w", 1, "-mask", os.path.join(GOLDEN, "test_blob_detect_mask.rec")],
    ["-fluct-aniso", 60, 90, 50, "-w", 19.6, "-truncate", 2.2],
    ["-fluctuations", 70, "-w", 19.6, "-truncate-threshold", 0.1, "-normalize-filters", "no"],
])
def test_cli_filters_equal_reference_program(cli, ref_cli, tmp_path, flags):
    mine, ref = both(cli, ref_cli, tmp_path, ["-in", os.path.join(GOLDEN, "test_blob_detect.rec")] + flags)
    assert_bits_equal(volgen.read_mrc(str(mine / "out.rec")), volgen.read_mrc(str(ref / "out.rec")), " ".join(map(str, flags)))


@pytest.mark.gpu
def test_cli_blob_files_equal_reference_program(cli, ref_cli, tmp_path):
    """-blob minima/maxima list files and the -discard-blobs result, text for text."""
    common = ["-w", 19.6, "-mask", os.path.join(GOLDEN, "test_blob_detect_mask.rec"), "-in", os.path.join(GOLDEN, "test_blob_detect.rec")]
    outs = {}
    for tag, exe in (("mine", cli), ("ref", ref_cli)):
        d = tmp_path / tag
        d.mkdir()
        for args in (["-blob", "minima", "mins.txt", 160.0, 280.0, 1.01],
                     ["-blob", "all", "both", 150.0, 300.0, 1.05],
                     ["-discard-blobs", "mins.txt", "kept.txt", "-blob-separation", 1.1, "-minima-threshold", -90]):
            r = subprocess.run([exe] + [str(a) for a in common + args], cwd=str(d), capture_output=True, text=True)
            assert r.returncode == 0, (tag, args, r.stderr[-2000:])
        outs[tag] = {f: open(d / f).read() for f in sorted(os.listdir(d)) if f.endswith(".txt")}
    assert sorted(outs["mine"]) == sorted(outs["ref"]), (sorted(outs["mine"]), sorted(outs["ref"]))
    for f in outs["ref"]:
        assert outs["mine"][f] == outs["ref"][f], f
    assert len(outs["ref"]["kept.txt"].strip().split("\n")) == 2


@pytest.mark.gpu
def test_cli_tolerance_modes_from_the_environment(cli, ref_cli, tmp_path):
    """VISFD_HIP_TV_FMA=1 / VISFD_HIP_GAUSS_FMA=1 switch the command line to the tolerance kernels (a context starts from the
    environment): vote tensors and the post-vote saliency within 1e-5 of the reference program's, but not its bits; a plain
    -gauss within 1e-5; a LoG -- whose values feed index comparisons -- still bit for bit."""
    env = {"VISFD_HIP_TV_FMA": "1", "VISFD_HIP_GAUSS_FMA": "1"}
    inp = os.path.join(GOLDEN, "test_image_membrane.rec")
    base = ["-w", 19.2, "-in", inp, "-membrane", "minima", 55, "-tv", 4, "-tv-angle-exponent", 4, "-bin", 1]
    mine, ref = both(cli, ref_cli, tmp_path, base + ["-save-progress", "prog"], "sal.rec", env=env)
    differs = False
    for c in range(6):
        a, b = volgen.read_mrc(str(mine / ("prog_tensor_%d.rec" % c))), volgen.read_mrc(str(ref / ("prog_tensor_%d.rec" % c)))
        assert_close_rel(a, b, 1e-5, "tolerance-mode vote tensor channel %d" % c)
        differs = differs or not np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert differs, "the tolerance kernels did not run"
    assert_close_rel(volgen.read_mrc(str(mine / "sal.rec")), volgen.read_mrc(str(ref / "sal.rec")), 1e-5, "post-vote saliency")
    blob = os.path.join(GOLDEN, "test_blob_detect.rec")
    mine, ref = both(cli, ref_cli, tmp_path, ["-in", blob, "-gauss", 40, "-w", 19.6], "g.rec", env=env)
    assert_close_rel(volgen.read_mrc(str(mine / "g.rec")), volgen.read_mrc(str(ref / "g.rec")), 1e-5, "-gauss in tolerance mode")
    mine, ref = both(cli, ref_cli, tmp_path, ["-in", blob, "-log-d", 120, "-w", 19.6], "l.rec", env=env)
    assert_bits_equal(volgen.read_mrc(str(mine / "l.rec")), volgen.read_mrc(str(ref / "l.rec")), "-log-d under the tolerance options")


@pytest.mark.gpu
def test_cli_labels_end_to_end_in_both_modes(cli, ref_cli, tmp_path):
    """north_star: "bit-exact for non-max/label indices".  A 64^3 volume with three separate membranes through the whole command --
    `-membrane ... -tv ... -connect ...`, the vote tensors from THIS program's GPU kernels, not the reference's files --
    against the reference program's label volume: identical labels in exact mode; in tolerance mode
    (VISFD_HIP_TV_FMA / GAUSS_FMA / EIG_F32 = 1) the number of voxels whose label differs is counted, written to
    gpurun_out/r4_label_census.txt and bounded."""
    src = volgen.three_membranes(64, seed=411)
    inp = str(tmp_path / "vol.rec")
    volgen.write_mrc(inp, src, voxel_width=1.0)
    base = ["-w", 1, "-in", inp, "-membrane", "minima", 3, "-tv", 4, "-tv-angle-exponent", 4, "-bin", 1]
    # the clustering threshold: a quarter of the largest post-vote saliency of the reference's own run
    mine, ref = both(cli, ref_cli, tmp_path, base, "sal.rec")
    sal_ref = volgen.read_mrc(str(ref / "sal.rec"))
    assert_close_rel(volgen.read_mrc(str(mine / "sal.rec")), sal_ref, 1e-5, "post-vote saliency, exact mode")
    thr = 0.25 * float(sal_ref.max())
    tail = ["-connect", thr, "-connect-angle", 30]
    mine, ref = both(cli, ref_cli, tmp_path, base + tail, "labels.rec")
    want = volgen.read_mrc(str(ref / "labels.rec"))
    n_clusters = int(want.max()) - 1                      # (voxels outside every cluster carry the label n_clusters + 1)
    n_lab = int((want < want.max()).sum())
    assert n_clusters >= 3 and n_lab > 10000, "the reference found %d clusters, %d labelled voxels" % (n_clusters, n_lab)
    got = volgen.read_mrc(str(mine / "labels.rec"))
    lines = ["64^3 synthetic membrane volume, -membrane minima 3 -tv 4 -connect %.6g -connect-angle 30: the reference labels %d voxels "
             "in %d clusters" % (thr, n_lab, n_clusters)]
    lines.append("exact mode: %d voxels with a label different from the reference program's" % int((got != want).sum()))
    assert_bits_equal(got, want, "cluster labels, exact mode, tensors from the GPU")
    env = {"VISFD_HIP_TV_FMA": "1", "VISFD_HIP_GAUSS_FMA": "1", "VISFD_HIP_EIG_F32": "1"}
    d = tmp_path / "tol"
    d.mkdir()
    r = subprocess.run([cli] + [str(a) for a in base + tail] + ["-out", "labels.rec"], cwd=str(d), capture_output=True, text=True,
                       env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr[-2000:]
    tol = volgen.read_mrc(str(d / "labels.rec"))
    ndiff = int((tol != want).sum())
    lines.append("tolerance mode (tv_fma, gauss_fma, eig_f32): %d voxels with a different label (%.3g of the labelled ones)" % (
        ndiff, ndiff / n_lab))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r4_label_census.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    assert ndiff <= 0.002 * n_lab, lines[-1]


@pytest.mark.gpu
def test_cli_membrane_scenario_equals_reference_program(cli, ref_cli, tmp_path):
    """Both commands of tests/test_membrane_detection.sh through both programs: the six vote-tensor files and the
    post-vote saliency agree to 1e-5 (the senders' directions come from the device's fp64 eigen solver, equal to
    glibc's to ~1e-7), and the cluster labels -- computed from the SAME tensor files -- are identical."""
    inp = os.path.join(GOLDEN, "test_image_membrane.rec")
    base = ["-w", 19.2, "-in", inp, "-membrane", "minima", 55, "-tv", 4, "-tv-angle-exponent", 4, "-bin", 2]
    mine, ref = both(cli, ref_cli, tmp_path, base + ["-save-progress", "prog"], "sal.rec")
    for c in range(6):
        assert_close_rel(volgen.read_mrc(str(mine / ("prog_tensor_%d.rec" % c))),
                         volgen.read_mrc(str(ref / ("prog_tensor_%d.rec" % c))), 1e-5, "vote tensor channel %d" % c)
    assert_close_rel(volgen.read_mrc(str(mine / "sal.rec")), volgen.read_mrc(str(ref / "sal.rec")), 1e-5, "post-vote saliency")
    for c in range(6):   # cluster from identical inputs: the reference's tensor files
        os.replace(str(ref / ("prog_tensor_%d.rec" % c)), str(mine / ("prog_tensor_%d.rec" % c)))
        import shutil
        shutil.copy(str(mine / ("prog_tensor_%d.rec" % c)), str(ref / ("prog_tensor_%d.rec" % c)))
    mine, ref = both(cli, ref_cli, tmp_path, base + ["-load-progress", "prog", "-connect", 1e9, "-connect-angle", 30], "labels.rec")
    a, b = volgen.read_mrc(str(mine / "labels.rec")), volgen.read_mrc(str(ref / "labels.rec"))
    assert_bits_equal(a, b, "cluster labels")
    assert int(((b > 0.99) & (b < 1.01)).sum()) == 69


@pytest.mark.gpu
@pytest.mark.parametrize("flags,exact", [
    (["-membrane", "minima", 55, "-tv", 4, "-tv-angle-exponent", 4, "-bin", 1], False),
    (["-membrane", "maxima", 55, "-tv", 3, "-tv-angle-exponent", 2, "-bin", 1, "-tv-best", 0.2], False),
    (["-membrane", "minima", 45, "-bin", 1], False),                                   # ridge saliency only, no voting
    (["-membrane", "minima", 55, "-tv", 4, "-bin", 1, "-detection-threshold", 2000.0], False),
    (["-membrane", "minima", 90, "-tv", 3], False),                                    # sigma > 1.8 voxels: automatic binning + un-binning
    (["-membrane", "minima", 55, "-tv", 4, "-bin", 1, "-truncate", 2.0, "-tv-truncate-ratio", 1.2], False),
    # the peak-height factor of both score loops (handlers.cpp:1577-1605,1698-1702,1883-1887; settings.cpp:2802-2825)
    (["-membrane", "minima", 55, "-tv", 4, "-tv-angle-exponent", 4, "-bin", 1, "-membrane-background", 120], False),
    (["-membrane", "maxima", 55, "-tv", 4, "-bin", 1, "-detection-background", 90, "-tv-best", 0.2], False),
    (["-membrane", "minima", 45, "-bin", 1, "-membrane-background", 100], False),     # ridge scores only
    (["-membrane", "minima", 90, "-tv", 3, "-membrane-background", 150], False),      # with automatic binning
    (["-log-d", 60, "-dog-delta", 0.05, "-truncate-threshold", 0.01], True),
    (["-gauss", 30, "-normalize-filters", "no"], True),
])
def test_cli_membrane_and_options_equal_reference_program(cli, ref_cli, tmp_path, flags, exact):
    """More of the command line against the reference's own program on its 16^3 membrane fixture: eigen-derived
    outputs to 1e-5 (relative to the volume's scale), pure filters bit for bit."""
    inp = os.path.join(GOLDEN, "test_image_membrane.rec")
    mine, ref = both(cli, ref_cli, tmp_path, ["-w", 19.2, "-in", inp] + flags)
    a, b = volgen.read_mrc(str(mine / "out.rec")), volgen.read_mrc(str(ref / "out.rec"))
    assert a.shape == b.shape
    if exact:
        assert_bits_equal(a, b, " ".join(map(str, flags)))
    else:
        assert np.abs(b).max() > 0
        assert_close_rel(a, b, 1e-5, " ".join(map(str, flags)))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [
    ["-blob-s", "all", "b", 20.0, 48.0, 1.3, "-minima-threshold", -20, "-maxima-threshold", 20],
    ["-blob-r", "maxima", "b.txt", 40.0, 80.0, 1.25],
    ["-blob-s", "all", "b", 1.0, 2.4, 1.3],                      # scales far below a voxel: both lists empty
    ["-blob", "minima", "b.txt", 60.0, 150.0, 1.2, "-dog-delta", 0.05, "-truncate-threshold", 0.02],
    ["-blob-s", "all", "b", 20.0, 48.0, 1.3, "-blob-aspect-ratio", 1.0, 1.3, 0.8],
    ["-blob", "minima", "b.txt", 60.0, 150.0, 1.2, "-blob-aspect-ratio", 1.5, 1.0, 1.0, "-minima-threshold", -10],
])
def test_cli_blob_variants_equal_reference_program(cli, ref_cli, tmp_path, flags):
    """Blob detector spellings (-blob = diameters, -blob-s = sigmas, -blob-r = radii; minima / maxima / all) and
    score thresholds on the small membrane fixture: list files identical text for text."""
    inp = os.path.join(GOLDEN, "test_image_membrane.rec")
    outs = {}
    for tag, exe in (("mine", cli), ("ref", ref_cli)):
        d = tmp_path / tag
        d.mkdir()
        r = subprocess.run([exe] + [str(a) for a in ["-w", 19.2, "-in", inp] + flags], cwd=str(d), capture_output=True, text=True)
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        outs[tag] = {f: open(d / f).read() for f in sorted(os.listdir(d))}
    assert sorted(outs["mine"]) == sorted(outs["ref"]) and outs["ref"], (sorted(outs["mine"]), sorted(outs["ref"]))
    for f in outs["ref"]:
        assert outs["mine"][f] == outs["ref"][f], f


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [
    ["-connect", 1e9, "-connect-angle", 30, "-undefined-out", -3],
    ["-connect", 5e8, "-connect-vector-saliency", 0.5, "-connect-vector-neighbor", 0.7, "-connect-tensor-saliency", 0.1,
     "-connect-tensor-neighbor", 0.6],
    ["-connect", 2e9],
    ["-connect-dark", 1e9, "-connect-angle", 30],
])
def test_cli_clustering_options_equal_reference_program(cli, ref_cli, tmp_path, flags):
    """Clustering of the SAME vote tensors (written once by the reference's -save-progress) with different
    thresholds: label volumes identical."""
    inp = os.path.join(GOLDEN, "test_image_membrane.rec")
    base = ["-w", 19.2, "-in", inp, "-membrane", "minima", 55, "-tv", 4, "-tv-angle-exponent", 4, "-bin", 2]
    import shutil
    for tag in ("mine", "ref"):
        (tmp_path / tag).mkdir()
    r = subprocess.run([ref_cli] + [str(a) for a in base + ["-save-progress", "prog", "-out", "s.rec"]],
                       cwd=str(tmp_path / "ref"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    for c in range(6):
        shutil.copy(str(tmp_path / "ref" / ("prog_tensor_%d.rec" % c)), str(tmp_path / "mine" / ("prog_tensor_%d.rec" % c)))
    mine, ref = both(cli, ref_cli, tmp_path, base + ["-load-progress", "prog"] + flags, "labels.rec")
    assert_bits_equal(volgen.read_mrc(str(mine / "labels.rec")), volgen.read_mrc(str(ref / "labels.rec")), " ".join(map(str, flags)))


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode,imod_flags,use_w", [
    ("a.mrc", 2, None, False),      # float, voxel width from the header's cell size
    ("a.mrc", 0, None, True),       # bytes in .mrc: signed
    ("a.rec", 0, None, True),       # bytes in .rec: unsigned (mrc_simple.cpp:186-192)
    ("a.mrc", 0, 0, True),          # IMOD stamp, flag bit 0 clear: unsigned
    ("a.mrc", 0, 1, True),          # IMOD stamp, flag bit 0 set: signed
    ("a.mrc", 1, None, True),       # int16
    ("a.mrc", 6, None, True),       # uint16
])
def test_cli_mrc_input_modes_equal_reference_program(cli, ref_cli, tmp_path, name, mode, imod_flags, use_w):
    """The bulk MRC reader against the reference's one-voxel-at-a-time reader: every input mode and the signed-byte
    rules, checked through a Gaussian of the decoded volume; and the output header's size, mode and cell."""
    rng = np.random.default_rng(9)
    lo, hi = {0: (-100, 100), 1: (-3000, 3000), 2: (-5, 5), 6: (0, 60000)}[mode]
    if mode == 0 and (name.endswith(".rec") or imod_flags == 0):
        lo, hi = 0, 250
    vol = rng.uniform(lo, hi, (10, 12, 16))
    vol = vol.astype(np.float32) if mode == 2 else np.round(vol)
    inp = tmp_path / name
    volgen.write_mrc(str(inp), vol, voxel_width=2.5, mode=mode, imod_flags=imod_flags)
    flags = ["-in", inp, "-gauss", 4.0] + (["-w", 2.5] if use_w else [])
    mine, ref = both(cli, ref_cli, tmp_path, flags)
    assert_bits_equal(volgen.read_mrc(str(mine / "out.rec")), volgen.read_mrc(str(ref / "out.rec")), "%s mode %d" % (name, mode))
    ha = np.frombuffer(open(mine / "out.rec", "rb").read(1024), "<i4")
    hb = np.frombuffer(open(ref / "out.rec", "rb").read(1024), "<i4")
    assert list(ha[0:4]) == list(hb[0:4])                                                     # nx, ny, nz, mode 2
    assert np.allclose(ha.view("<f4")[10:13], hb.view("<f4")[10:13])                          # cell size
    assert np.allclose(ha.view("<f4")[19:22], hb.view("<f4")[19:22], rtol=1e-5, atol=1e-4)    # dmin, dmax, dmean


def _cluster_and_normals(cli, ref_cli, tmp_path, bin_flag, extra=()):
    """The reference program writes the vote tensors once; both programs then cluster them and export the oriented
    point cloud of cluster 1.  Returns (labels mine, labels ref, ply mine, ply ref)."""
    import shutil
    inp = os.path.join(GOLDEN, "test_image_membrane.rec")
    base = ["-w", 19.2, "-in", inp, "-membrane", "minima", 55, "-tv", 4, "-tv-angle-exponent", 4, "-bin", bin_flag]
    for tag in ("mine", "ref"):
        (tmp_path / tag).mkdir()
    r = subprocess.run([ref_cli] + [str(a) for a in base + ["-save-progress", "prog", "-out", "s.rec"]],
                       cwd=str(tmp_path / "ref"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    for c in range(6):
        shutil.copy(str(tmp_path / "ref" / ("prog_tensor_%d.rec" % c)), str(tmp_path / "mine" / ("prog_tensor_%d.rec" % c)))
    mine, ref = both(cli, ref_cli, tmp_path, base + ["-load-progress", "prog", "-connect", 1e9, "-connect-angle", 30,
                                                     "-normals-file", "n.ply", "-select-cluster", 1] + list(extra), "labels.rec")
    return (volgen.read_mrc(str(mine / "labels.rec")), volgen.read_mrc(str(ref / "labels.rec")),
            open(mine / "n.ply").read(), open(ref / "n.ply").read())


@pytest.mark.parametrize("extra", [(), ("-max-voxels-to-feature", 0.6), ("-max-distance-to-feature", "inf")])
def test_cli_clustering_and_normals_host_path_equal_reference_program(cli, ref_cli, tmp_path, extra):
    """-load-progress ... -connect ... -normals-file with -bin 1 needs no GPU at all (tensor files in, host-side
    clustering and surface-point export): label volume identical and PLY file identical text for text."""
    a, b, pa, pb = _cluster_and_normals(cli, ref_cli, tmp_path, 1, extra)
    assert_bits_equal(a, b, "cluster labels")
    assert pa == pb
    assert int(pb.split("element vertex ")[1].split()[0]) > 20


@pytest.mark.parametrize("notation", ["physical", "imod"])
def test_cli_must_link_equal_reference_program(cli, ref_cli, tmp_path, notation):
    """-must-link FILE (settings.cpp:3183-3195, file_io.hpp:667-747, connect.hpp:829-1045): two parallel membranes form
    two clusters; a must-link group with one location on each merges them.  Host path only (the vote tensors are
    written once by the reference's -save-progress): label volumes and the oriented point cloud identical to the
    reference program's, for coordinates in physical units (with an explicit direction column) and in IMOD's
    "Pixel (x, y, z) = v" notation (1-based voxels)."""
    import shutil
    rng = np.random.default_rng(3)
    n = 24
    v = rng.normal(100, 2, (n, n, n)).astype(np.float32)
    z = np.arange(n, dtype=np.float32)[:, None, None]
    for z0 in (6.0, 17.0):
        v -= (60 * np.exp(-((z - z0) ** 2) / 2.0)).astype(np.float32)
    for tag in ("mine", "ref"):
        (tmp_path / tag).mkdir()
        volgen.write_mrc(str(tmp_path / tag / "two.rec"), v, voxel_width=1.0)
        with open(tmp_path / tag / "ml.txt", "w") as f:
            f.write("12 12 6   # first membrane\n11 13 17 1\n" if notation == "physical" else
                    "Pixel (13, 13, 7) = 3.2\n(12, 14, 18)\n")
    base = ["-w", 1, "-in", "two.rec", "-membrane", "minima", 2, "-tv", 3, "-tv-angle-exponent", 4, "-bin", 1]
    r = subprocess.run([ref_cli] + [str(a) for a in base + ["-save-progress", "prog", "-out", "s.rec"]],
                       cwd=str(tmp_path / "ref"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    for c in range(6):
        shutil.copy(str(tmp_path / "ref" / ("prog_tensor_%d.rec" % c)), str(tmp_path / "mine" / ("prog_tensor_%d.rec" % c)))
    cluster = base + ["-load-progress", "prog", "-connect", 5, "-connect-angle", 30]
    counts = {}
    for name, extra in (("free", []), ("linked", ["-must-link", "ml.txt", "-normals-file", "n.ply", "-select-cluster", 1])):
        mine, ref = both(cli, ref_cli, tmp_path, cluster + extra, name + ".rec")
        a, b = volgen.read_mrc(str(mine / (name + ".rec"))), volgen.read_mrc(str(ref / (name + ".rec")))
        assert_bits_equal(a, b, "cluster labels, " + name)
        counts[name] = len(np.unique(b))
    assert counts["free"] == 3 and counts["linked"] == 2      # two membranes + "undefined" -> one cluster + "undefined"
    assert open(mine / "n.ply").read() == open(ref / "n.ply").read()


@pytest.mark.gpu
def test_cli_normals_file_reference_scenario(cli, ref_cli, tmp_path):
    """The second command of tests/test_membrane_detection.sh including -normals-file: 58 vertices,
    file identical to the reference program's."""
    a, b, pa, pb = _cluster_and_normals(cli, ref_cli, tmp_path, 2)
    assert_bits_equal(a, b, "cluster labels")
    assert "element vertex 58" in pb
    assert pa == pb


# ---- Z-slab mode of the program (one filter_mrc per GPU; no reference counterpart) ----------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("id_file", ["-", "id"])
def test_cli_membrane_slab_mode_one_rank_equals_plain_run(cli, tmp_path, id_file):
    """`-slab 0 1 IDFILE`: the slab stage of csrc/slab.hip through the program's host-memory entry point -- with IDFILE "-"
    without a communicator, with a file through a one-rank RCCL communicator made from the id the program publishes there
    (what a one-GPU box can run of the multi-process path).  The output file equals the plain run's byte for byte."""
    inp = os.path.join(GOLDEN, "test_image_membrane.rec")
    flags = ("-w", 19.2, "-in", inp, "-membrane", "minima", 55, "-tv", 4, "-tv-angle-exponent", 4, "-bin", 1, "-tv-best", 0.1)
    plain, slab = tmp_path / "plain.rec", tmp_path / "slab.rec"
    r = run(cli, *flags, "-out", plain)
    assert r.returncode == 0, r.stderr
    idf = "-" if id_file == "-" else str(tmp_path / "rccl.id")
    r = run(cli, *flags, "-out", slab, "-slab", 0, 1, idf)
    assert r.returncode == 0, r.stderr
    assert "slab 0 of 1" in r.stderr
    assert open(plain, "rb").read() == open(slab, "rb").read()
    if id_file != "-":
        assert not os.path.exists(idf), "rank 0 removes the id file once the communicator is up (no stale ids for a later run)"


@pytest.mark.gpu
@pytest.mark.parametrize("id_file", ["-", "rccl.id"])
def test_cli_gauss_blob_and_background_slab_mode_one_rank_equal_plain_runs(cli, tmp_path, id_file):
    """BASELINE config 5 is "full Gauss+TV pipeline": `-slab` also runs `-gauss` (visfd_hip_apply_gauss_slab), `-blob`
    (visfd_hip_blob_dog_slab) and the membrane stage with its peak-height factor; with one rank every output equals the
    plain run's byte for byte (volumes) / line for line (blob lists)."""
    blob_in = os.path.join(GOLDEN, "test_blob_detect.rec")
    mem_in = os.path.join(GOLDEN, "test_image_membrane.rec")
    idf = "-" if id_file == "-" else str(tmp_path / "rccl.id")
    for k, flags in enumerate([("-in", blob_in, "-gauss", 40, "-w", 19.6),
                               ("-in", blob_in, "-gauss-aniso", 30, 45, 25, "-w", 19.6, "-normalize-filters", "no"),
                               ("-w", 19.2, "-in", mem_in, "-membrane", "minima", 55, "-tv", 4, "-bin", 1, "-membrane-background", 120)]):
        plain, slab = tmp_path / ("plain%d.rec" % k), tmp_path / ("slab%d.rec" % k)
        r = run(cli, *flags, "-out", plain)
        assert r.returncode == 0, r.stderr
        r = run(cli, *flags, "-out", slab, "-slab", 0, 1, idf)
        assert r.returncode == 0, r.stderr
        assert open(plain, "rb").read() == open(slab, "rb").read(), flags
    bflags = ("-in", blob_in, "-w", 19.6, "-blob-s", "all", None, 20.0, 48.0, 1.3, "-minima-threshold", -20, "-maxima-threshold", 20)
    outs = []
    for tag, extra in (("plain", ()), ("slab", ("-slab", 0, 1, idf))):
        f = [str(tmp_path / (tag + "_b")) if a is None else a for a in bflags]
        r = run(cli, *f, *extra)
        assert r.returncode == 0, r.stderr
        outs.append([open(str(tmp_path / (tag + "_b")) + suffix).read() for suffix in (".minima.txt", ".maxima.txt")])
    assert outs[0] == outs[1] and len(outs[0][0]) > 0


def test_cli_slab_mode_argument_checks(cli, tmp_path):
    inp = os.path.join(GOLDEN, "test_image_membrane.rec")
    r = run(cli, "-in", inp, "-out", tmp_path / "o.rec", "-membrane", "minima", 55, "-tv", 4, "-slab", 2, 2, "x")
    assert r.returncode == 1 and "RANK < WORLD" in r.stderr
    r = run(cli, "-in", inp, "-out", tmp_path / "o.rec", "-membrane", "minima", 55, "-tv", 4, "-slab", 0)
    assert r.returncode == 1


def test_join_slabs_tool(tmp_path):
    """tools/join_slabs.py stacks the per-rank files of a slab run (CPU only)."""
    import subprocess
    import sys
    rng = np.random.default_rng(5)
    vol = rng.standard_normal((9, 5, 7)).astype(np.float32)
    names = []
    for k, (a, b) in enumerate(((0, 4), (4, 9))):
        p = tmp_path / ("s%d.rec" % k)
        volgen.write_mrc(str(p), vol[a:b])                  # voxel width 1: cell z = number of planes
        names.append(str(p))
    out = tmp_path / "joined.rec"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "join_slabs.py"), str(out)] + names, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert_bits_equal(volgen.read_mrc(str(out)), vol, "joined slabs")
    hdr = np.frombuffer(open(out, "rb").read()[:1024], np.float32)
    assert hdr[12] == np.float32(9.0) and hdr[19] == vol.min() and hdr[20] == vol.max()
