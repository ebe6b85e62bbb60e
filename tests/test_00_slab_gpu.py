"""Z-slab pipeline on the GPU: two ranks with the real kernels against the
single-volume run, bit for bit.  The ranks are started through torch.distributed.run exactly as
the driver starts bench.py.  This file sorts first on purpose: the launcher is started before this
pytest process has touched the GPU (a GPU box does not hand the device to programs exec'ed from a
process that already initialised it)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("world,path", [(2, "c-abi"), (3, "c-abi"), (2, "python")])
def test_slab_pipeline_on_gpu_equals_single_volume(world, path):
    """c-abi: the slab entry points of csrc/slab.hip (visfd_hip_membrane_detect_slab_dev, visfd_hip_blob_dog_slab_dev) --
    halo groups on the transfer stream, the global radix select with all-reduced histograms, overlapped interior votes --
    driven through the callback transport (the ranks share this box's one GPU, so RCCL itself cannot run here);
    python: the torch.distributed orchestration of visfd_amd/slab.py over the same kernels."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tools", "slab_check.py")]
    env = dict(os.environ)
    if path == "python":
        env["SLAB_PY"] = "1"
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "SLAB-OK world=%d" % world in r.stdout and "path=%s" % path in r.stdout, \
        (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` run directly (as the driver runs the N=1 line) starts its ranks itself as a fresh child
    and relays ONE JSON line and the exit code; on a one-GPU box the ranks share the card (a rehearsal of the code path:
    the number means nothing)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "96", "--steps", "1",
                        "--warmup", "1", "--no-2048", "--no-cpu"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["per_gpu_voxels"] == 96 ** 3
    assert abs(out["value"] - 2 * 96 ** 3 / (out["ms_per_step"] * 1e-3) / 1e6) <= 1e-3 * out["value"]


@pytest.mark.gpu
def test_rccl_transport_loopback():
    """What a one-GPU box can run of the RCCL path of csrc/slab.hip: librccl.so loaded at run time, a one-rank communicator
    (ncclGetUniqueId / ncclCommInitRank), a grouped ncclSend + ncclRecv of 4 MiB to itself on the slab's transfer stream
    ordered against the context's stream by events, and an ncclAllReduce(uint64, sum) of the 2048 histogram counters --
    both verified on the host.  (Run in a child process: RCCL initialises its own state.)"""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from visfd_amd import api\n"
            "ctx = api.Context(0)\n"
            "s = api.Slab(ctx, 0, 1, 64, 12, transport='rccl_loopback')\n"
            "s.selftest(1 << 20)\n"
            "assert (s.z0, s.z1, s.nz_local) == (0, 64, 64)\n"
            "s.close(); ctx.close(); print('RCCL-LOOPBACK-OK')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-LOOPBACK-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


@pytest.mark.gpu
def test_reserved_workgroup_slots_admit_a_second_stream():
    """The 64 workgroup slots a slab run keeps free while a halo is in flight (csrc/slab.hip -> option tv_reserve_wg) are
    evidence, not a guess: with them a kernel of a second stream, queued right after the vote has started, finishes long
    before the vote does; without them -- the exact kernel owns every wave slot of the chip -- it waits for the vote's
    workgroups to exit (tools/reserve_check.py, timed with events)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "reserve_check.py")], cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "RESERVE-OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
