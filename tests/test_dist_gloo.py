"""Multi-process partitions (pathtracer_amd/dist.py) on the CPU: world_size 2 over gloo, with the oracle standing
in for the renderer (the partition/merge code takes the renderer as a callable, so the same code runs on the GPU box
with the HIP tracer behind it)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathtracer_amd import dist as ptdist

HERE = os.path.dirname(os.path.abspath(__file__))


def test_sample_indices_cover_every_sample_once():
    for world in (1, 2, 3, 8):
        for total in (0, 1, 5, 16, 1024, 1025):
            seen = []
            for r in range(world):
                first, stride, count = ptdist.sample_indices(r, world, total)
                assert stride == world and first == r + 1
                seen += [first + k * stride for k in range(count)]
            assert sorted(seen) == list(range(1, total + 1))
    with pytest.raises(ValueError):
        ptdist.sample_indices(2, 2, 10)


def test_row_blocks_partition_the_image():
    for world in (1, 2, 4, 8):
        for height in (1, 15, 16, 17, 1080, 2160):
            rows = np.zeros(height, np.int32)
            for r in range(world):
                for y0, h in ptdist.row_blocks(r, world, height):
                    rows[y0:y0 + h] += 1
            assert np.all(rows == 1)
    a = ptdist.row_blocks(0, 4, 2160)
    b = ptdist.row_blocks(3, 4, 2160)
    assert abs(sum(h for _, h in a) - sum(h for _, h in b)) <= 16


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    sys.path.insert(0, HERE)
    import pt_oracle as po
    from helpers import load_case
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx = load_case("cornell_geo_128")
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    W = H = 128
    u = po.uniforms(fx["camera"], W, H, len(fx["indices"]), 10, 1, 1)
    total = 6
    if mode == "sample":
        first, stride, count = ptdist.sample_indices(rank, world, total)
        acc, _ = po.render_sum(scene, u, first, stride, count, threads=2)
        img = torch.from_numpy(acc)
        ptdist.merge_sample_split(img, total)
    elif mode == "sample_overlapped":
        # two frames (samples 1..6 and 7..12) through the merger that exchanges one frame while the next renders
        merger = ptdist.OverlappedSampleMerge(torch.zeros(H, W, 4), total)
        assert merger.asynchronous
        for frame in range(2):
            first, stride, count = ptdist.sample_indices(rank, world, total)
            acc, _ = po.render_sum(scene, u, first + frame * total, stride, count, threads=2)
            sums = torch.from_numpy(acc)
            merger.submit(sums)
            sums.zero_()   # the renderer's image is free again at once
        img = merger.finish().clone()   # frame 1; frame 0 was returned by the finish() inside the second submit
    else:
        rgba = np.zeros((H, W, 4), np.float32)
        rgba[..., 3] = 1.0
        for y0, h in ptdist.row_blocks(rank, world, H, block=16):
            po.render(scene, u, 1, total, rgba, window=(0, y0, W, h), threads=2)
        img = torch.from_numpy(rgba)
        ptdist.zero_unowned_rows(img, rank, world, block=16)
        ptdist.merge_tile_split(img, H, block=16)
    np.save(os.path.join(out_dir, f"{mode}_{rank}.npy"), img.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["sample", "sample_overlapped", "tile"])
def test_two_ranks_reproduce_single_process(tmp_path, mode):
    import pt_oracle as po
    from helpers import load_case
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    fx = load_case("cornell_geo_128")
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    u = po.uniforms(fx["camera"], 128, 128, len(fx["indices"]), 10, 1, 1)
    if mode == "sample_overlapped":   # its result is the second frame: samples 7..12 on their own
        acc, _ = po.render_sum(scene, u, 7, 1, 6, threads=8)
        ref = acc * np.float32(1.0 / 6.0)
    else:
        ref, _ = po.render(scene, u, 1, 6, threads=8)
    outs = [np.load(tmp_path / f"{mode}_{r}.npy") for r in range(world)]
    assert np.array_equal(outs[0], outs[1]), "every rank holds the merged image"
    if mode == "tile":
        # tile split keeps the reference recurrence per pixel: bit-identical to one process
        assert np.array_equal(outs[0].view(np.uint32), ref.view(np.uint32))
    else:
        # sample split sums then scales: equal to the sequential running mean up to fp32 rounding
        assert np.abs(outs[0][..., :3] - ref[..., :3]).max() < 2e-6
        assert np.all(outs[0][..., 3] == 1.0)
