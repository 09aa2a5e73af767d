"""The multi-GPU step over the real backend, as far as one GPU goes: a one-rank "nccl" (RCCL) process group, the render
into a torch tensor on a side stream, the in-place all-reduce, the overlapped merge bench.py uses for --gpus N
(pathtracer_amd.dist.OverlappedSampleMerge) and the tile partition's all-gather (TileGather), every frame checked. World sizes above 1 are covered on CPU tensors in
test_dist_gloo.py; RCCL refuses two ranks on one device."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_one_rank_rccl_step_and_overlapped_merge():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # a child process: the process group and its communicator end with it, whatever the other tests did to the device
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank.py")], env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "5 steps of render + RCCL all-reduce" in r.stdout
    assert "6 frames with the exchange overlapped" in r.stdout
    assert "5 frames through the RCCL all-gather of the owned rows" in r.stdout
