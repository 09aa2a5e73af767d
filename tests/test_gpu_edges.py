"""Edge cases of the HIP path against the CPU oracle (bit for bit): tiny and ragged images, windows that cut tiles, bounce
limits 0 and 1, no triangles traced, degenerate and duplicated triangles, long sample ranges with strides, argument errors."""
import numpy as np
import pytest

import pt_oracle as po
from helpers import load_case, n_bit_mismatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tracer():
    from pathtracer_amd import Tracer
    t = Tracer(0)
    yield t
    t.close()


def upload(t, fx, W, H, depth=10, ntri=None, cam=None, kind=1):
    prog = t.program_create(kind)
    t.use_program(prog)
    t.upload_spheres(fx["spheres"])
    t.upload_materials(fx["materials"])
    t.upload_mesh(fx["vertices"], fx["indices"])
    t.set_normal_mode(int(fx["normal_mode"]))
    t.set_shading_mode(0)
    t.set_accel_mode(0)
    t.set_accum_mode(0)
    t.target_create(W, H)
    cam = po.camera(60.0, W, H, (0, 0, 0), (0, 0, 1), (0, 1, 0)) if cam is None else cam
    n = len(fx["indices"]) if ntri is None else ntri
    t.set_uniforms(cam, n, depth)
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    u = po.uniforms(cam, W, H, n, depth, kind, int(fx["normal_mode"]))
    return prog, scene, u


@pytest.mark.parametrize("size", [(1, 1), (7, 3), (9, 17), (64, 1), (1, 70), (130, 66)])
@pytest.mark.parametrize("kind", [0, 1])
def test_small_and_ragged_images(tracer, size, kind):
    fx = load_case("monkey_mesh_128")
    W, H = size
    prog, scene, u = upload(tracer, fx, W, H, kind=kind)
    ids, ts = tracer.readback_primary()
    tracer.dispatch(1, 5, 1)
    tracer.finish()
    oids, ots, _ = po.primary(scene, u)
    ref, rays = po.render(scene, u, 1, 5)
    assert np.array_equal(ids, oids) and n_bit_mismatch(ts, ots) == 0
    assert n_bit_mismatch(tracer.readback(), ref) == 0
    assert tracer.stats()["rays"] >= W * H * 5
    tracer.reset_stats()
    tracer.program_destroy(prog)


@pytest.mark.parametrize("accel", [1, 2, 4])
@pytest.mark.parametrize("window", [(3, 5, 1, 1), (5, 9, 11, 7), (0, 0, 128, 1), (127, 0, 1, 128), (17, 33, 90, 41)])
def test_windows_that_cut_tiles(tracer, window, accel):
    """Only the window is rendered; what lies outside keeps the clear value. Twice, so that the second dispatch runs from a
    work list with split tiles."""
    fx = load_case("monkey_mesh_128")
    prog, scene, u = upload(tracer, fx, 128, 128)
    tracer.build_accel()
    tracer.set_accel_mode(accel)
    x0, y0, w, h = window
    tracer.set_window(*window)
    ref, _ = po.render(scene, u, 1, 6, window=window)
    for _ in range(2):
        tracer.target_clear()
        tracer.dispatch(1, 6, 1)
        tracer.finish()
        img = tracer.readback()
        assert n_bit_mismatch(img, ref) == 0
    outside = np.ones((128, 128), bool)
    outside[y0:y0 + h, x0:x0 + w] = False
    assert np.all(img[outside][:, :3] == 0.0) and np.all(img[..., 3] == 1.0)
    tracer.set_window()
    tracer.set_accel_mode(0)
    tracer.program_destroy(prog)


@pytest.mark.parametrize("depth", [0, 1, 2, 3])
@pytest.mark.parametrize("reuse", [True, False])
def test_bounce_limits(tracer, depth, reuse):
    """u_Depth bounds the bounce loop (fs:388): 0 = no closest-hit query at all, the sample is the sky in the camera
    direction; 1 = the camera ray only."""
    fx = load_case("cornell_geo_128")
    prog, scene, u = upload(tracer, fx, 128, 128, depth=depth, cam=fx["camera"])
    tracer.set_primary_reuse(reuse)
    tracer.reset_stats()
    tracer.dispatch(1, 4, 1)
    tracer.finish()
    ref, rays = po.render(scene, u, 1, 4, threads=8)
    assert n_bit_mismatch(tracer.readback(), ref) == 0
    st = tracer.stats()
    assert st["rays"] == rays
    if depth == 0:
        assert rays == 0 and st["rays_traced"] == 0
    if depth == 1:
        assert rays == 128 * 128 * 4 and st["rays_traced"] == (128 * 128 if reuse else rays)
    tracer.set_primary_reuse(True)
    tracer.program_destroy(prog)


def test_no_triangles_traced(tracer):
    fx = load_case("monkey_mesh_128")
    prog, scene, u = upload(tracer, fx, 64, 48, ntri=0)
    ids, _ = tracer.readback_primary()
    tracer.dispatch(3, 2, 4)
    tracer.finish()
    assert np.all(ids == -1)
    img = tracer.readback()
    # every sample is the sky colour of the camera ray: the mean of equal values, folded with weights 1/3 and 1/7
    sky = np.zeros((48, 64, 4), np.float32)
    sky[..., 3] = 1.0
    u1 = po.uniforms(np.asarray(po.camera(60.0, 64, 48, (0, 0, 0), (0, 0, 1), (0, 1, 0))), 64, 48, 0, 10, 1, 0)
    a, _ = po.render(scene, u1, 3, 1, sky.copy())
    b, _ = po.render(scene, u1, 7, 1, a)
    assert n_bit_mismatch(img, b) == 0
    tracer.program_destroy(prog)


@pytest.mark.parametrize("accel", [1, 2, 4])
def test_degenerate_and_duplicated_triangles(tracer, accel):
    """Zero-area triangles (collapsed to a point, to a line), exact duplicates (the lower index must win) and triangles
    seen edge-on, mixed into a random soup."""
    rng = np.random.default_rng(5)
    nt = 240
    base = rng.uniform(-1.5, 1.5, (nt, 1, 3)) + np.array([0, 0, 5.0])
    tri = (base + rng.uniform(-0.8, 0.8, (nt, 3, 3))).astype(np.float32)
    tri[10:20, 1] = tri[10:20, 0]; tri[10:20, 2] = tri[10:20, 0]          # points
    tri[20:30, 2] = (tri[20:30, 0] + tri[20:30, 1]) * np.float32(0.5)     # lines
    tri[120:180] = tri[30:90]                                              # duplicates of earlier triangles
    tri[180:190, :, 0] = 0.0                                               # in the plane x = 0 that contains the camera
    nrm = rng.normal(size=(nt * 3, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    verts = np.concatenate([tri.reshape(-1, 3), nrm], axis=1).astype(np.float32)
    idx = np.arange(nt * 3, dtype=np.uint32).reshape(-1, 3)
    sph, mat = po.reference_spheres_and_materials()
    fx = {"vertices": verts, "indices": idx, "spheres": sph, "materials": mat, "normal_mode": 0}
    prog, scene, u = upload(tracer, fx, 96, 80)
    tracer.build_accel()
    tracer.set_accel_mode(accel)
    ids, ts = tracer.readback_primary()
    tracer.dispatch(1, 4, 1)
    tracer.finish()
    oids, ots, _ = po.primary(scene, u, threads=8)
    ref, rays = po.render(scene, u, 1, 4, threads=8)
    assert np.array_equal(ids, oids) and n_bit_mismatch(ts, ots) == 0
    assert not np.isin(ids, np.arange(120, 180)).any(), "a duplicate with the higher index won a tie"
    assert not np.isin(ids, np.arange(10, 30)).any()
    assert n_bit_mismatch(tracer.readback(), ref) == 0
    assert tracer.stats()["rays"] >= 96 * 80 * 4
    tracer.set_accel_mode(0)
    tracer.reset_stats()
    tracer.program_destroy(prog)


def test_long_strided_sample_range(tracer):
    """200 samples with stride 5 from index 3 in one dispatch equal 200 single-sample dispatches with those indices."""
    fx = load_case("world_128")
    prog, scene, u = upload(tracer, fx, 40, 24, kind=0)
    tracer.dispatch(3, 200, 5)
    tracer.finish()
    batch = tracer.readback()
    tracer.target_clear()
    for k in range(200):
        tracer.dispatch(3 + 5 * k, 1, 1)
    tracer.finish()
    assert n_bit_mismatch(tracer.readback(), batch) == 0
    rgba = None
    for k in range(200):
        rgba, _ = po.render(scene, u, 3 + 5 * k, 1, rgba)
    assert n_bit_mismatch(batch, rgba) == 0
    tracer.program_destroy(prog)


def test_argument_errors(tracer):
    from pathtracer_amd import PtError
    t = tracer
    fx = load_case("monkey_mesh_128")
    prog, _, _ = upload(t, fx, 16, 16)
    for bad in [(0, 1, 1), (1, 0, 1), (1, 1, 0), (1 << 24, 2, 1), (1, 1 << 23, 3)]:
        with pytest.raises(PtError):
            t.dispatch(*bad)
    with pytest.raises(PtError):
        t.target_create(0, 16)
    with pytest.raises(PtError):
        t.target_create(16, 70000)
    t.target_create(16, 16)
    with pytest.raises(PtError):
        t.set_window(8, 8, 16, 4)          # sticks out of the image
    with pytest.raises(PtError):
        t.set_uniforms(fx["camera"], len(fx["indices"]) + 1, 10)
        t.dispatch(1, 1, 1)                # more triangles than uploaded
    t.set_uniforms(fx["camera"], len(fx["indices"]), 10)
    with pytest.raises(PtError):
        t.set_accel_mode(2)
        t.upload_mesh(fx["vertices"], fx["indices"])   # drops the hierarchy
        t.dispatch(1, 1, 1)                # PT_ACCEL_BVH without one
    t.set_accel_mode(0)
    t.dispatch(1 << 24, 1, 1)              # the last representable sample index is fine
    t.finish()
    t.program_destroy(prog)


def test_short_arithmetic_sequences_equal_ieee_for_every_input(tracer):
    """The kernels' 4-instruction reciprocal, unscaled square root and log2 quotient against the compiler's IEEE division and
    square root: all 2^32 float inputs (2^23 mantissas for the quotient), results compared bit for bit on the device."""
    assert tracer.selftest_arithmetic() == (0, 0, 0)


@pytest.mark.parametrize("where", ["reference", "inside", "grazing", "far", "tiny"])
def test_sphere_program_from_awkward_camera_positions(tracer, where):
    """The sphere program with the camera at the reference position,
    inside a sphere, almost on a surface, far away (spheres of a few pixels), and with the radii shrunk to a hundredth:
    primary ids, distances, image and ray count against the oracle."""
    fx = dict(load_case("world_128"))
    sph = np.array(fx["spheres"], np.float32).reshape(10, 4).copy()
    eye, target = (0.0, 0.0, 0.0), (0.0, 0.0, 1.0)
    if where == "inside":
        eye = tuple(sph[3, :3] + np.float32(0.3) * sph[3, 3])
        target = tuple(sph[5, :3])
    elif where == "grazing":
        eye = tuple(sph[2, :3] + np.array([0, 1.0002, 0], np.float32) * sph[2, 3])
        target = tuple(np.array(eye, np.float32) + np.array([1.0, -0.05, 0.2], np.float32))   # along the surface, slightly down
    elif where == "far":
        eye = (40.0, 25.0, -150.0)
        target = tuple(sph[:, :3].mean(0))
    elif where == "tiny":
        sph[:, 3] *= np.float32(0.01)
    fx["spheres"] = sph
    W, H = 160, 120
    cam = po.camera(60.0 if where != "far" else 8.0, W, H, eye, target, (0, 1, 0))
    prog, scene, u = upload(tracer, fx, W, H, cam=cam, kind=0)
    ids, ts = tracer.readback_primary()
    tracer.reset_stats()
    tracer.dispatch(1, 12, 1)
    tracer.finish()
    oids, ots, _ = po.primary(scene, u, threads=8)
    ref, rays = po.render(scene, u, 1, 12, threads=8)
    assert np.array_equal(ids, oids) and n_bit_mismatch(ts, ots) == 0
    img = tracer.readback()
    both_nan = np.isnan(img) & np.isnan(ref)
    assert int(((img.view(np.uint32) != ref.view(np.uint32)) & ~both_nan).sum()) == 0
    assert tracer.stats()["rays"] == rays
    if where in ("reference", "inside", "far"):
        assert (ids >= 0).sum() > 20
    tracer.reset_stats()
    tracer.program_destroy(prog)


@pytest.mark.parametrize("kind", [0, 1])
def test_camera_ray_hits_kept_between_dispatches(kind):
    """Frame loop of short dispatches: the camera ray's hit is kept from one dispatch to the next of the same view. Every
    change of the view (camera direction, position, window, normal mode, search mode, triangle count, clear) must drop
    or keep it correctly: the images equal those of a context with the mechanism switched off after every dispatch, the
    reference's ray count is the same, fewer queries are evaluated, and the first frames equal the oracle's."""
    import os
    from pathtracer_amd import Tracer
    fx = load_case("monkey_mesh_128")
    W, H = 96, 64
    old = os.environ.get("PT_PRIMARY_CACHE")
    tracers = []
    try:
        for flag in ("1", "0"):
            os.environ["PT_PRIMARY_CACHE"] = flag
            tracers.append(Tracer(0))
    finally:
        if old is None:
            del os.environ["PT_PRIMARY_CACHE"]
        else:
            os.environ["PT_PRIMARY_CACHE"] = old
    try:
        scene = u = None
        for t in tracers:
            _, scene, u = upload(t, fx, W, H, kind=kind)
            t.build_accel()
        ntri = len(fx["indices"])
        cam_turned = po.camera(60.0, W, H, (0, 0, 0), (0.3, 0.1, 1), (0, 1, 0))     # same position, other direction
        cam_moved = po.camera(60.0, W, H, (0.2, -0.1, 0.1), (0.3, 0.1, 1), (0, 1, 0))
        sample = [1]

        def frames(n, spp=1):
            for _ in range(n):
                for t in tracers:
                    t.dispatch(sample[0], spp, 1)
                    t.finish()
                sample[0] += spp
                a, b = tracers[0].readback(), tracers[1].readback()
                assert n_bit_mismatch(a, b) == 0

        frames(3)
        ref, _ = po.render(scene, u, 1, 3)
        assert n_bit_mismatch(tracers[0].readback(), ref) == 0
        for change in ("turn", "move", "window", "normals", "accel", "fewer", "clear", "long"):
            for t in tracers:
                if change == "turn":
                    t.set_uniforms(cam_turned, ntri, 10)
                elif change == "move":
                    t.set_uniforms(cam_moved, ntri, 10)
                elif change == "window":
                    t.set_window(10, 5, 50, 40)
                elif change == "normals":
                    t.set_normal_mode(1 - int(fx["normal_mode"]))
                elif change == "accel":
                    t.set_accel_mode(2 if kind == 1 else 0)
                elif change == "fewer" and kind == 1:
                    t.set_accel_mode(1)
                    t.set_uniforms(cam_moved, ntri // 2, 10)
                elif change == "clear":
                    t.target_clear()
            if change == "clear":
                sample[0] = 1
            frames(2, spp=40 if change == "long" else 1)
        sa, sb = tracers[0].stats(), tracers[1].stats()
        assert sa["rays"] == sb["rays"]
        assert sa["rays_traced"] < sb["rays_traced"]
    finally:
        for t in tracers:
            t.close()


@pytest.mark.parametrize("kth", [1, 3, 7, 8])
def test_failed_grid_allocation_leaves_no_half_made_grid(kth):
    """The k-th device allocation of the direction-grid build reports out-of-memory (PT_FAIL_GRID_ALLOC, a test knob read
    at pt_create): the dispatch returns PT_ERR_HIP, and the next one builds the grid from scratch and renders the oracle's
    image -- no launch on null arrays."""
    import os
    from pathtracer_amd import PtError, Tracer
    fx = load_case("monkey_mesh_128")
    os.environ["PT_FAIL_GRID_ALLOC"] = str(kth)
    try:
        t = Tracer(0)
    finally:
        del os.environ["PT_FAIL_GRID_ALLOC"]
    try:
        prog, scene, u = upload(t, fx, 64, 48)
        t.build_accel()
        t.set_accel_mode(4)
        with pytest.raises(PtError) as err:
            t.dispatch(1, 3, 1)
        assert err.value.status == 2 and "grid" in str(err.value)
        t.dispatch(1, 3, 1)
        t.finish()
        ref, _ = po.render(scene, u, 1, 3)
        assert n_bit_mismatch(t.readback(), ref) == 0
        assert t.stats()["grid_entries"] > 0
        t.program_destroy(prog)
    finally:
        t.close()


def test_tile_costs_are_fresh_after_a_move_and_short_dispatches(tracer, capfd):
    """After a camera move, single-sample frames gather costs for the new view; the long dispatch that follows is scheduled
    from them in one go (one scheduler run over all its samples) instead of being treated as a view without costs, which
    would render an 8-sample probe first."""
    import os
    fx = load_case("monkey_mesh_128")
    prog, scene, u = upload(tracer, fx, 128, 128)
    ntri = len(fx["indices"])
    for k in range(3):
        tracer.dispatch(1 + k, 1, 1)
    tracer.set_uniforms(po.camera(60.0, 128, 128, (0.1, 0.0, 0.05), (0, 0, 1), (0, 1, 0)), ntri, 10)   # the move
    tracer.target_clear()
    for k in range(3):
        tracer.dispatch(1 + k, 1, 1)
    tracer.finish()
    capfd.readouterr()
    os.environ["PT_DEBUG_SCHED"] = "1"
    try:
        tracer.dispatch(4, 64, 1)
        tracer.finish()
    finally:
        del os.environ["PT_DEBUG_SCHED"]
    lines = [l for l in capfd.readouterr().err.splitlines() if l.startswith("[pt sched]")]
    assert lines and all(" 64 samples" in l for l in lines), lines
    u2 = po.uniforms(po.camera(60.0, 128, 128, (0.1, 0.0, 0.05), (0, 0, 1), (0, 1, 0)), 128, 128, ntri, 10, 1, int(fx["normal_mode"]))
    ref, _ = po.render(scene, u2, 1, 67, threads=8)
    assert n_bit_mismatch(tracer.readback(), ref) == 0
    tracer.program_destroy(prog)


@pytest.mark.parametrize("workload,size,spp", [("cornell", (7680, 4320), 2), ("cornell", (65535, 5), 3), ("monkeyhead2", (3, 30000), 3)])
def test_targets_far_beyond_the_benchmark_sizes(workload, size, spp):
    """A frame four times the largest benchmark frame (518,400 tiles) and the two thinnest targets a 16-bit image size allows
    (aspect ratios of 13,107 and 1/10,000: nearly every tile is cut by the image's edge), each whole frame against the oracle:
    camera-ray ids and distances, radiance and ray count, bit for bit."""
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(workload, *size)
    W, H = scene.width, scene.height
    assert (W, H) == size
    with Tracer(0) as t:
        scene.upload(t)
        t.target_create(W, H)
        ids, ts = t.readback_primary()
        t.dispatch(1, spp, 1)
        t.finish()
        img = t.readback()
        rays = t.stats()["rays"]
    sc = po.Scene(scene.vertices, scene.indices, scene.spheres, scene.materials)
    u = po.uniforms(scene.camera, W, H, scene.num_triangles, 10, po.VARIANT_MESH, scene.normal_mode)
    oids, ots, _ = po.primary(sc, u, threads=16)
    assert np.array_equal(ids, oids) and n_bit_mismatch(ts, ots) == 0
    assert (oids >= 0).any()
    del oids, ots
    ref, orays = po.render(sc, u, 1, spp, threads=16)
    assert n_bit_mismatch(img, ref) == 0
    assert rays == orays > W * H * spp


@pytest.mark.parametrize("workload,size,shading,spp", [("suzanne", (640, 360), "reference", 40), ("instanced", (960, 540), "materials", 12),
                                                        ("cornell", (333, 201), "reference", 33)])
def test_work_list_dealt_to_the_xcds_changes_nothing(workload, size, shading, spp):
    """PT_XCD_GROUPS=1 cuts the tiles into eight image bands of equal cost and gives each XCD one of them (list positions g,
    g + 8, ...; bands differ in length, the gaps hold no work). Any list that covers every (tile, sample) once renders the same
    image: compared with the single cost-ordered list over several dispatches (probe, first list, rebuilt list, split tiles)."""
    import os
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(workload, *size)
    out = {}
    for groups in ("0", "1"):
        old = os.environ.get("PT_XCD_GROUPS")
        os.environ["PT_XCD_GROUPS"] = groups
        try:
            t = Tracer(0)
        finally:
            if old is None:
                del os.environ["PT_XCD_GROUPS"]
            else:
                os.environ["PT_XCD_GROUPS"] = old
        with t:
            scene.upload(t, shading=shading)
            t.target_create(scene.width, scene.height)
            for k in range(4):
                t.dispatch(1 + k * spp, spp, 1)
            t.set_window(17, 9, scene.width - 40, scene.height - 30)
            t.dispatch(1 + 4 * spp, spp, 1)
            t.dispatch(1 + 5 * spp, spp, 1)
            t.finish()
            st = t.stats()
            out[groups] = (t.readback().copy(), st["rays"], st["samples"])
    assert np.array_equal(out["0"][0].view(np.uint32), out["1"][0].view(np.uint32))
    assert out["0"][1:] == out["1"][1:]


@pytest.mark.parametrize("workload,size,spp", [("cornell", (333, 201), 33), ("suzanne", (640, 360), 40), ("monkeyhead2", (480, 270), 24),
                                               ("instanced", (960, 540), 16), ("torus", (400, 300), 20)])
def test_both_forms_of_the_grid_kernels_loop_render_the_same(workload, size, spp):
    """The direction-grid kernel exists twice: the wave runs its list loop as often as its longest list (PT_GRID_CARRY=0), or a lane
    tests one triangle per trip and carries the rest of its cell's list into its next trip (=1). The library times both on a view
    and keeps the quicker (unset). Same tests, same lowest-index rule: images, ray counts, camera-ray triangles and distances must
    be identical bit for bit, over a cold dispatch, list rebuilds, split tiles, a window, a row partition and a camera move;
    geometric and interpolated normals."""
    import os
    from pathtracer_amd import Tracer, capi
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(workload, *size)
    out = {}
    for form in ("0", "1", None):
        old = os.environ.get("PT_GRID_CARRY")
        if form is None:
            os.environ.pop("PT_GRID_CARRY", None)
        else:
            os.environ["PT_GRID_CARRY"] = form
        try:
            t = Tracer(0)
        finally:
            if old is None:
                os.environ.pop("PT_GRID_CARRY", None)
            else:
                os.environ["PT_GRID_CARRY"] = old
        with t:
            scene.upload(t)
            t.set_accel_mode(capi.ACCEL_GRID)
            t.target_create(scene.width, scene.height)
            forms = set()
            for k in range(5):
                t.dispatch(1 + k * spp, spp, 1)
                forms.add(t.stats()["grid_lists_carried"])
            ids, ts = t.readback_primary()
            t.set_window(17, 9, scene.width - 40, scene.height - 30)
            t.dispatch(1 + 5 * spp, spp, 1)
            t.set_window(0, 0, 0, 0)
            t.set_row_interleave(8, 1, 3)
            t.dispatch(1 + 6 * spp, spp, 1)
            t.set_row_interleave(16, 0, 1)
            cam = np.array(scene.camera, np.float32).copy()
            cam[0:3] += np.float32(0.01)        # the camera moves: new grid, the choice starts again
            t.set_uniforms(cam, scene.num_triangles, 10)
            t.dispatch(1 + 7 * spp, spp, 1)
            t.dispatch(1 + 8 * spp, 1, 1)       # a single-sample frame
            t.finish()
            st = t.stats()
            out[form] = (t.readback().copy(), ids.copy(), ts.copy(), st["rays"], st["rays_traced"], st["samples"], forms)
    assert out["0"][6] == {0} and out["1"][6] == {1}
    for form in ("1", None):
        assert np.array_equal(out["0"][0].view(np.uint32), out[form][0].view(np.uint32)), form
        assert np.array_equal(out["0"][1], out[form][1]) and np.array_equal(out["0"][2].view(np.uint32), out[form][2].view(np.uint32)), form
        assert out["0"][3:6] == out[form][3:6], form
