"""Sparse views: the pixels whose camera ray hits the mesh gathered into tiles of 64 (PT_PIXEL_LIST=1: whenever the view has a
direction grid; =0: never; unset: the library's own rule) against the window's 8x8 blocks. Pure re-mapping of lanes to pixels
plus a separate pass for the sky pixels: images, ray counts in the reference's sense and sample counts must be identical bit for
bit -- reference shading through the grid, the materials mode through the hierarchy (camera rays through the grid), running mean
and running sum, cold dispatches, list rebuilds, split tiles, a window, a row partition, a camera move, single-sample frames."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tracer(mode):
    from pathtracer_amd import Tracer
    old = os.environ.get("PT_PIXEL_LIST")
    if mode is None:
        os.environ.pop("PT_PIXEL_LIST", None)
    else:
        os.environ["PT_PIXEL_LIST"] = mode
    try:
        return Tracer(0)
    finally:
        if old is None:
            os.environ.pop("PT_PIXEL_LIST", None)
        else:
            os.environ["PT_PIXEL_LIST"] = old


@pytest.mark.parametrize("workload,size,spp,shading,accum", [
    ("suzanne", (640, 360), 40, "reference", "mean"), ("monkeyhead2", (480, 270), 24, "reference", "sum"),
    ("instanced", (960, 540), 16, "reference", "mean"), ("cornell", (333, 201), 33, "reference", "mean"),
    ("suzanne", (400, 225), 16, "materials", "mean"), ("instanced", (640, 360), 12, "materials", "sum"),
    ("torus", (400, 300), 20, "reference", "mean")])
def test_gathered_hit_pixels_render_the_same(workload, size, spp, shading, accum):
    from pathtracer_amd import capi
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(workload, *size)
    out = {}
    for mode in ("0", "1", None):
        with _tracer(mode) as t:
            scene.upload(t, shading=shading)
            t.target_create(scene.width, scene.height)
            t.set_accum_mode(capi.ACCUM_SUM if accum == "sum" else capi.ACCUM_MEAN)
            images, gathered = [], []
            for k in range(4):
                t.dispatch(1 + k * spp, spp, 1)
                gathered.append(t.stats()["gathered_pixels"] > 0)
            images.append(t.readback().copy())
            t.set_window(17, 9, scene.width - 40, scene.height - 30)
            t.dispatch(1 + 4 * spp, spp, 1)
            t.dispatch(1 + 5 * spp, spp, 1)
            t.set_window(0, 0, 0, 0)
            images.append(t.readback().copy())
            t.set_row_interleave(8, 1, 3)
            t.dispatch(1 + 6 * spp, spp, 1)
            t.dispatch(1 + 7 * spp, spp, 1)
            t.set_row_interleave(16, 0, 1)
            images.append(t.readback().copy())
            cam = np.array(scene.camera, np.float32).copy()
            cam[0:3] += np.float32(0.01)        # the camera moves: new grid, new list
            t.set_uniforms(cam, scene.num_triangles, 10)
            t.dispatch(1 + 8 * spp, spp, 1)
            t.dispatch(1 + 9 * spp, spp, 1)
            t.dispatch(1 + 10 * spp, 1, 1)      # a single-sample frame
            gathered.append(t.stats()["gathered_pixels"] > 0)
            t.dispatch(2 + 10 * spp, 300, 1)    # a long dispatch (probe + scheduled rest, split tiles)
            t.finish()
            images.append(t.readback().copy())
            st = t.stats()
            assert st["gathered_tiles"] == (st["gathered_pixels"] + 63) // 64
            out[mode] = (images, st["rays"], st["samples"], gathered)
    for mode in ("1", None):
        for a, b in zip(out["0"][0], out[mode][0]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), mode
        assert out["0"][1:3] == out[mode][1:3], mode
    assert not any(out["0"][3])
    # forced: every dispatch of a view that has a direction grid (the materials mode builds it with a view's second dispatch)
    assert out["1"][3] == ([True] * 5 if shading == "reference" else [False] + [True] * 4)
    # the library's own rule: from a view's second dispatch on, dispatches of 8 samples or more, views whose occupied 8x8 blocks would leave 1/32 or more of their lanes to sky pixels
    if workload in ("suzanne", "monkeyhead2", "instanced"):
        assert out[None][3] == [False, True, True, True, False]



def test_a_camera_that_turns_every_frame_gets_no_list():
    """Every list renumbers the tiles and so drops the scheduler's costs: the library builds one only when the previous dispatch
    rendered the same camera (orientation included), window and partition."""
    from pathtracer_amd.scenes import SceneData
    scene = SceneData("monkeyhead2", 480, 270)
    with _tracer(None) as t:
        scene.upload(t)
        t.target_create(scene.width, scene.height)
        cam = np.array(scene.camera, np.float32).copy()
        used = []
        for k in range(5):
            cam[3:6] += np.float32(0.002)       # the lower left corner of the image plane moves: the camera turns, its position stays
            t.set_uniforms(cam, scene.num_triangles, 10)
            t.dispatch(1, 16, 1)
            used.append(t.stats()["gathered_pixels"] > 0)
        for k in range(3):                      # ... and then stays
            t.dispatch(17 + 16 * k, 16, 1)
            used.append(t.stats()["gathered_pixels"] > 0)
        t.finish()
    assert used == [False] * 5 + [True] * 3
