"""bench.py --gpus N started directly (no torch.distributed.run): the parent starts the ranks itself.

CPU part: both ranks start and report the missing GPU; the parent passes the failure on. GPU part (one-GPU rehearsal: both
ranks share device 0, gloo in place of RCCL): both partitions run end to end; the tile partition's merged image equals a
one-rank run bit for bit, the sample partition's within fp32 rounding of the sequential mean."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, extra_env=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_direct_start_with_two_gpus_spawns_both_ranks():
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU rehearsal below covers this")
    r = _run(["--gpus", "2", "--lean", "--steps", "1", "--warmup", "0"], {"PT_BENCH_BACKEND": "gloo"})
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") == 2, r.stderr     # both ranks were started and got as far as the GPU check
    assert "WORLD_SIZE" not in r.stderr and "exited with" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("partition", ["tile", "sample"])
def test_two_rank_rehearsal_on_one_gpu(tmp_path, partition):
    common = ["--lean", "--steps", "2", "--warmup", "1", "--workload", "monkeyhead2", "--width", "320", "--height", "184"]
    one = str(tmp_path / "one.npy")
    two = str(tmp_path / "two.npy")
    spp_one = "32" if partition == "tile" else "64"       # sample split: 2 ranks x 32 samples = 64 in all
    r1 = _run(common + ["--gpus", "1", "--spp", spp_one, "--dump-image", one])
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = _run(common + ["--gpus", "2", "--spp", "32", "--partition", partition, "--dump-image", two],
              {"PT_BENCH_SHARE_GPU": "1", "PT_BENCH_BACKEND": "gloo"})
    assert r2.returncode == 0, r2.stderr[-2000:]
    line = json.loads(r2.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == ("strong" if partition == "tile" else "weak")
    assert 0 < line["value"] <= line["value_reference_definition"]
    a, b = np.load(one), np.load(two)
    if partition == "tile":
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    else:
        assert np.abs(a[..., :3] - b[..., :3]).max() < 5e-6 and np.all(b[..., 3] == 1.0)
    single = json.loads(r1.stdout.strip().splitlines()[-1])
    assert single["roofline"]["frac"] <= 1.0 and single["roofline"]["frac"] > 0.0
