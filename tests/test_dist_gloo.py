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


def _worker(rank, world, port, mode, out_dir, H=128):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    sys.path.insert(0, HERE)
    import pt_oracle as po
    from helpers import load_case
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fx = load_case("cornell_geo_128")
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    W = 128
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
    elif mode == "tile":
        rgba = np.zeros((H, W, 4), np.float32)
        rgba[..., 3] = 1.0
        for y0, h in ptdist.row_blocks(rank, world, H, block=16):
            po.render(scene, u, 1, total, rgba, window=(0, y0, W, h), threads=2)
        img = torch.from_numpy(rgba)
        img[~_owned(rank, world, H)] = float("nan")   # whatever the other ranks' rows hold must not matter
        ptdist.merge_tile_split(img, H, block=16)
    else:
        # "tile_overlapped": two frames (passes 1..6, then 7..12 on top) through the gather that runs beside the next frame
        gather = ptdist.TileGather(torch.zeros(H, W, 4), rank, world, 16)
        assert gather.asynchronous
        rgba = np.zeros((H, W, 4), np.float32)
        rgba[..., 3] = 1.0
        for frame in range(2):
            for y0, h in ptdist.row_blocks(rank, world, H, block=16):
                po.render(scene, u, 1 + frame * total, total, rgba, window=(0, y0, W, h), threads=2)
            mine = torch.from_numpy(rgba.copy())
            gather.submit(mine)
            mine.fill_(float("nan"))   # the renderer's image is free again at once
        img = gather.finish().clone()
    np.save(os.path.join(out_dir, f"{mode}_{rank}.npy"), img.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _owned(rank, world, H):
    own = torch.zeros(H, dtype=torch.bool)
    for y0, h in ptdist.row_blocks(rank, world, H, 16):
        own[y0:y0 + h] = True
    return own


@pytest.mark.parametrize("mode,world,H", [("sample", 2, 128), ("sample_overlapped", 2, 128), ("tile", 2, 128), ("tile", 3, 100),
                                         ("tile_overlapped", 2, 72), ("tile_overlapped", 3, 100)])
def test_ranks_reproduce_single_process(tmp_path, mode, world, H):
    import pt_oracle as po
    from helpers import load_case
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path), H), nprocs=world, join=True)
    fx = load_case("cornell_geo_128")
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    u = po.uniforms(fx["camera"], 128, H, len(fx["indices"]), 10, 1, 1)
    if mode == "sample_overlapped":   # its result is the second frame: samples 7..12 on their own
        acc, _ = po.render_sum(scene, u, 7, 1, 6, threads=8)
        ref = acc * np.float32(1.0 / 6.0)
    elif mode == "tile_overlapped":   # passes 1..12 folded into one image
        ref, _ = po.render(scene, u, 1, 12, threads=8)
    else:
        ref, _ = po.render(scene, u, 1, 6, threads=8)
    outs = [np.load(tmp_path / f"{mode}_{r}.npy") for r in range(world)]
    for o in outs[1:]:
        assert np.array_equal(outs[0].view(np.uint32), o.view(np.uint32)), "every rank holds the merged image"
    if mode.startswith("tile"):
        # tile split keeps the reference recurrence per pixel, and the merge only moves bytes: bit-identical to one process,
        # also when the height is no multiple of block x world (ranks own different numbers of rows)
        assert np.array_equal(outs[0].view(np.uint32), ref.view(np.uint32))
    else:
        # sample split sums then scales: equal to the sequential running mean up to fp32 rounding
        assert np.abs(outs[0][..., :3] - ref[..., :3]).max() < 2e-6
        assert np.all(outs[0][..., 3] == 1.0)


def test_tile_gather_moves_half_of_what_the_zero_padded_sum_moved():
    like = torch.zeros(2160, 64, 4)
    for world in (2, 4, 8):
        g = ptdist.TileGather(like, 0, world, 16)
        image_bytes = like.numel() * 4
        # ring all-reduce of the whole image: 2 (world - 1) / world x image per rank; the gather: (world - 1) x one share
        assert g.bytes_sent_per_rank <= 0.52 * (2 * (world - 1) / world * image_bytes) * 1.02
