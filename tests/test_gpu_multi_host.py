"""The C++ host's multi-GPU render loop (pthost::RunRenderLoopMulti, host/MultiGpu.cpp) as far as one GPU allows: N logical ranks
that share device 0 (their buffers meet in host memory; the packing, the partition, the strided sample ranges and the per-frame
overlap are the ones the RCCL path uses) against the single-context loop, and one rank through RCCL itself (ncclCommInitAll,
ncclAllReduce / ncclAllGather on the exchange stream). Tile partition: bit for bit. Sample partition: the fp32 sum of the ranks'
running sums against the sequential running mean, |difference| < 2e-6 on radiance in [0, 1]."""
import numpy as np
import pytest

from helpers import n_bit_mismatch

pytestmark = pytest.mark.gpu


def _single(workload, W, H, passes, batch, shading="reference"):
    from pathtracer_amd import host_api
    return host_api.render_workload(workload, W, H, passes, batch, shading)


@pytest.mark.parametrize("workload,size,ranks,block_rows,shading", [
    ("monkeyhead2", (320, 180), 3, 16, "reference"),      # 12 row blocks, the last one ragged (4 rows) and owned by rank 2
    ("torus_distorted", (480, 270), 4, 16, "reference"),  # the benchmark's fourth configuration, scaled down
    ("cornell", (200, 120), 8, 5, "materials"),           # more ranks than some of them have full blocks; materials mode
    ("suzanne", (256, 144), 2, 16, "materials"),
])
def test_tile_partition_over_logical_ranks_is_bit_exact(workload, size, ranks, block_rows, shading):
    from pathtracer_amd import host_api
    W, H = size
    passes, batch = 12, 5                                  # frames of 5, 5 and 2 samples
    _tile_partition_case(host_api, workload, W, H, passes, batch, shading, ranks, block_rows)


def test_tile_partition_with_gathered_pixels():
    """Frames of ten samples: from a view's second frame on every rank renders the hit pixels of ITS rows gathered into tiles of 64
    (the list is per row partition), the single context those of the whole frame."""
    from pathtracer_amd import host_api
    _tile_partition_case(host_api, "monkeyhead2", 320, 180, 44, 10, "reference", 3, 16)
    _tile_partition_case(host_api, "suzanne", 256, 144, 36, 12, "materials", 2, 8)


def _tile_partition_case(host_api, workload, W, H, passes, batch, shading, ranks, block_rows):
    want, want8, st = _single(workload, W, H, passes, batch, shading)
    for every_frame in (True, False):
        got, got8, mst = host_api.render_workload(workload, W, H, passes, batch, shading, devices=[0] * ranks, partition="tile",
                                                  exchange="host", block_rows=block_rows, every_frame=every_frame)
        assert n_bit_mismatch(got, want) == 0
        assert np.array_equal(got8, want8)
        assert mst["samples"] == st["samples"] == W * H * passes and mst["rays"] == st["rays"] and not mst["rccl"]


@pytest.mark.parametrize("workload,size,ranks,shading,passes,batch", [("monkeyhead2", (320, 180), 4, "reference", 50, 6), ("world", (256, 144), 3, "reference", 50, 6),
                                                                       ("instanced", (480, 270), 2, "materials", 50, 6),
                                                                       ("instanced", (480, 270), 3, "reference", 130, 36)])   # frames long enough for gathered pixels
def test_sample_partition_over_logical_ranks(workload, size, ranks, shading, passes, batch):
    from pathtracer_amd import host_api
    W, H = size
    # (50 samples over the ranks: residue classes of unequal length)
    want, _, st = _single(workload, W, H, passes, batch, shading)
    got, _, mst = host_api.render_workload(workload, W, H, passes, batch, shading, devices=[0] * ranks, partition="sample", exchange="host")
    scale = 1.0 / passes
    # (the sphere scene has a few pixels that are NaN in every renderer: normalize() of a zero cross product in the shader's
    # tangent frame, pathtracing.fs:44-55; they must be NaN here too)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(want).mean() < 1e-3
    assert np.nanmax(np.abs(got * scale - want * scale)) < 2e-6 * max(1.0, float(np.nanmax(want)) * scale)
    assert mst["samples"] == st["samples"] == W * H * passes and mst["rays"] == st["rays"]
    assert (got != 0).any()


@pytest.mark.parametrize("partition", ["tile", "sample"])
def test_one_rank_through_rccl(partition):
    """RCCL refuses two ranks on one device, so the real backend runs with a communicator of one: ncclCommInitAll, the collective
    on the exchange stream behind the render stream's event, the unpacking -- every frame, beside the next frame's rendering."""
    from pathtracer_amd import host_api
    W, H, passes, batch = 320, 200, 24, 4
    want, want8, st = _single("monkeyhead2", W, H, passes, batch)
    got, got8, mst = host_api.render_workload("monkeyhead2", W, H, passes, batch, devices=[0], partition=partition, exchange="rccl")
    assert mst["rccl"] and mst["rays"] == st["rays"]
    if partition == "tile":
        assert n_bit_mismatch(got, want) == 0 and np.array_equal(got8, want8)
    else:
        assert np.abs(got - want).max() / passes < 2e-6


def test_multi_gpu_loop_reports_its_errors():
    from pathtracer_amd import host_api
    with pytest.raises(RuntimeError, match="device of its own"):
        host_api.render_workload("monkeyhead2", 64, 48, 4, 2, devices=[0, 0], exchange="rccl")
    with pytest.raises(RuntimeError, match="unknown workload"):
        host_api.render_workload("teapot", 64, 48, 4, 2, devices=[0, 0])
    with pytest.raises(RuntimeError):
        host_api.render_workload("monkeyhead2", 64, 48, 4, 2, devices=[0, 99], exchange="host")   # no such device: one rank fails, the other stops
