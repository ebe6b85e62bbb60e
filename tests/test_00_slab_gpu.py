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
@pytest.mark.parametrize("world", [2, 3])
def test_slab_pipeline_on_gpu_equals_single_volume(world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tools", "slab_check.py")]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SLAB-OK world=%d" % world in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
