"""GPU parity tests proper: the HIP path, called through the C ABI, against (a) the golden images the reference
shader produced on llvmpipe and (b) the CPU oracle on the same seeded inputs. fp32 throughout; the bar is
BIT-EXACT for ids, hit distances and radiance (the product evaluates the same operation sequence)."""
import numpy as np
import pytest

import pt_oracle as po
from helpers import CASES, case_shading_mode, load_case, n_bit_mismatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tracer():
    from pathtracer_amd import Tracer
    t = Tracer(0)
    yield t
    t.close()


def setup_case(t, fx):
    W, H = int(fx["width"]), int(fx["height"])
    kind = int(fx["variant"])
    prog = t.program_create(kind)
    t.use_program(prog)
    t.upload_spheres(fx["spheres"])
    t.upload_materials(fx["materials"])
    t.upload_mesh(fx["vertices"], fx["indices"])
    t.set_normal_mode(int(fx["normal_mode"]))
    t.set_shading_mode(case_shading_mode(fx))
    if case_shading_mode(fx):
        t.upload_material_ids(fx["material_ids"])
    t.set_accel_mode(0)
    t.set_accum_mode(0)
    t.target_create(W, H)
    t.set_uniforms(fx["camera"], len(fx["indices"]), int(fx["depth"]))
    return prog


@pytest.mark.parametrize("name", CASES)
def test_golden_case_bit_exact(tracer, name):
    fx = load_case(name)
    t = tracer
    prog = setup_case(t, fx)
    x0, y0, w, h = (int(v) for v in fx["crop"])
    t.set_window(x0, y0, w, h)
    win = (slice(y0, y0 + h), slice(x0, x0 + w))
    ids, ts = t.readback_primary()
    assert np.array_equal(ids[win], fx["lp_id"]), "primary-hit ids"
    hit = fx["lp_id"] >= 0
    assert n_bit_mismatch(ts[win][hit], fx["lp_t"][hit]) == 0, "primary-hit distances"
    done = 0
    for k in sorted(int(s) for s in fx["spp"]):
        # one dispatch per reference frame, like the reference's render loop
        for s in range(done + 1, k + 1):
            t.dispatch(s, 1, 1)
        t.finish()
        done = k
        img = t.readback()
        assert n_bit_mismatch(img[win][..., :3], fx[f"lp_rgb_{k}"]) == 0, f"radiance after {k} passes"
        assert np.all(img[win][..., 3] == 1.0)
    if "lp_present_rgba8" in fx:
        # the reference's copy.fs pass over the final image, as llvmpipe wrote it into an RGBA8 target
        assert np.array_equal(t.present(), fx["lp_present_rgba8"]), "present pass bytes"
    t.program_destroy(prog)


@pytest.mark.parametrize("name", ["monkey_mesh_128", "world_128", "cornell_geo_128"])
def test_multi_sample_dispatch_equals_frame_by_frame(tracer, name):
    """num_samples > 1 keeps the running mean in registers; the image must equal frame-by-frame dispatching."""
    fx = load_case(name)
    t = tracer
    prog = setup_case(t, fx)
    t.dispatch(1, 16, 1)
    t.finish()
    img = t.readback()
    assert n_bit_mismatch(img[..., :3], fx["lp_rgb_16"]) == 0
    st = t.stats()
    assert st["rays"] > 0
    t.program_destroy(prog)


def _tracer_with_lane_walk(value):
    import os
    from pathtracer_amd import Tracer
    old = os.environ.get("PT_LANE_WALK")
    os.environ["PT_LANE_WALK"] = value
    try:
        return Tracer(0)
    finally:
        if old is None:
            del os.environ["PT_LANE_WALK"]
        else:
            os.environ["PT_LANE_WALK"] = old


@pytest.fixture(scope="module")
def wave_walk_tracer():
    """A context whose materials mode walks the hierarchy once per wave (the default is once per lane, lanes decoupled)."""
    t = _tracer_with_lane_walk("0")
    yield t
    t.close()


@pytest.fixture(scope="module")
def in_step_walk_tracer():
    """A context whose materials mode walks the hierarchy once per lane with all lanes of a wave in step (the kernel the
    decoupled one replaced as the default; kept for comparisons)."""
    t = _tracer_with_lane_walk("2")
    yield t
    t.close()


@pytest.mark.parametrize("accel", ["brute", "bvh", "bvh_morton", "bvh_sah", "bvh_ploc", "bvh_wave"])
def test_materials_mode_random_scene_against_oracle(tracer, wave_walk_tracer, accel):
    """Diffuse/emissive mode on a seeded cloud of random triangles seen from inside (rays bounce from their hit points, hit emitters,
    leave through gaps): the all-triangles loop and the LBVH walk must both equal the CPU oracle bit for bit."""
    from pathtracer_amd import capi
    rng = np.random.default_rng(11)
    nt = 400
    base = rng.uniform(-2, 2, (nt, 1, 3)) + np.array([0, 0, 6.0])
    tri = (base + rng.uniform(-1.6, 1.6, (nt, 3, 3))).astype(np.float32)
    verts = np.concatenate([tri.reshape(-1, 3), np.zeros((nt * 3, 3), np.float32)], axis=1).astype(np.float32)
    idx = np.arange(nt * 3, dtype=np.uint32).reshape(-1, 3)
    mats = np.zeros((10, 4), np.float32)
    mats[:, :3] = rng.uniform(0.2, 0.95, (10, 3))
    mats[3, 3] = 7.5
    mats[8, 3] = 2.0
    mids = rng.integers(0, 10, nt).astype(np.int32)
    W, H = 150, 70
    cam = po.camera(75.0, W, H, (0.2, 0.1, 5.0), (0.0, 0.0, 7.0), (0.0, 1.0, 0.0))   # inside the cloud
    sph, _ = po.reference_spheres_and_materials()
    fx = {"width": W, "height": H, "variant": 1, "normal_mode": 1, "depth": 10, "camera": cam, "vertices": verts,
          "indices": idx, "spheres": sph, "materials": mats, "shading_mode": 1, "material_ids": mids}
    t = wave_walk_tracer if accel == "bvh_wave" else tracer
    prog = setup_case(t, fx)
    if accel != "brute":
        t.set_bvh_builder({"bvh_morton": capi.BVH_BUILDER_MORTON, "bvh_sah": capi.BVH_BUILDER_SAH, "bvh_ploc": capi.BVH_BUILDER_PLOC}.get(accel, capi.BVH_BUILDER_BINNED))
        t.build_accel()
        t.set_bvh_builder(capi.BVH_BUILDER_BINNED)
        t.set_accel_mode(capi.ACCEL_BVH)
    else:
        t.set_accel_mode(capi.ACCEL_BRUTE)
    t.reset_stats()
    t.dispatch(1, 3, 1)
    t.finish()
    img = t.readback()
    st = t.stats()
    scene = po.Scene(verts, idx, sph, mats, mids)
    u = po.uniforms(cam, W, H, nt, 10, 1, 1, 1)
    ref, rays = po.render(scene, u, 1, 3, threads=8)
    assert n_bit_mismatch(img, ref) == 0
    assert st["rays"] == rays and rays > 2 * W * H * 3
    if accel in ("bvh", "bvh_morton", "bvh_sah", "bvh_ploc"):
        assert st["grid_triangle_tests"] > 0 and st["bvh_node_fetches"] == 0     # per-lane walk: records counted per ray
    if accel == "bvh_wave":
        assert st["bvh_node_fetches"] > 0
    t.set_accel_mode(capi.ACCEL_AUTO)
    t.set_shading_mode(0)
    t.program_destroy(prog)


def _awkward_meshes():
    """Layouts that stress the four-child hierarchy with 16-bit boxes: no extent along an axis, a few tiny triangles very far from
    a dense cluster (a frame step is then larger than the cluster's triangles), every triangle the same, triangles with NaN
    coordinates (never hit, and they must not upset a builder), and triangle counts just above what one leaf holds."""
    rng = np.random.default_rng(23)
    out = {}
    g = 24
    xs, ys = np.meshgrid(np.linspace(-3, 3, g + 1), np.linspace(-2, 2, g + 1))
    P = np.stack([xs, ys, np.full_like(xs, 7.0)], -1).astype(np.float32)
    quads = []
    for j in range(g):
        for i in range(g):
            a, b, c, d = P[j, i], P[j, i + 1], P[j + 1, i + 1], P[j + 1, i]
            quads += [[a, c, b], [a, d, c]]
    out["flat"] = np.array(quads, np.float32)                                  # z extent 0
    cluster = (rng.uniform(-1, 1, (600, 1, 3)) * 0.5 + np.array([0, 0, 6.0]) + rng.uniform(-0.05, 0.05, (600, 3, 3))).astype(np.float32)
    far = (np.array([[4000.0, -3000.0, 9000.0], [-5000.0, 2500.0, 7000.0], [0.0, 6000.0, -8000.0]], np.float32)[:, None, :]
           + rng.uniform(-0.01, 0.01, (3, 3, 3)).astype(np.float32))
    big = np.array([[[-40, -6, 2.0], [40, -6, 2.0], [0, -6, 60.0]]], np.float32)   # a floor the cluster's bounce rays hit
    out["spread"] = np.concatenate([cluster, far, big]).astype(np.float32)
    one = np.array([[[-1, -1, 6.0], [0, 1, 6.0], [1, -1, 6.0]]], np.float32)
    out["same"] = np.repeat(one, 37, axis=0)
    holes = (rng.uniform(-1, 1, (300, 1, 3)) * 0.8 + np.array([0, 0, 6.0]) + rng.uniform(-0.2, 0.2, (300, 3, 3))).astype(np.float32)
    holes[7, 1, 0] = np.nan                                                  # one coordinate, one corner, whole triangles
    holes[50, 2] = np.nan
    holes[120:124] = np.nan
    out["nan"] = holes
    for n in (3, 5, 9):
        out[f"few{n}"] = (rng.uniform(-1.5, 1.5, (n, 1, 3)) + np.array([0, 0, 6.0]) + rng.uniform(-1.0, 1.0, (n, 3, 3))).astype(np.float32)
    return out


@pytest.mark.parametrize("layout", ["flat", "spread", "same", "nan", "few3", "few5", "few9"])
def test_materials_walk_on_awkward_layouts(tracer, layout):
    """The decoupled walk over the four-child form against the loop over all triangles, image and ray count bit for bit, and the
    device-side self-check of the form, on layouts chosen against its box encoding."""
    from pathtracer_amd import capi
    tri = _awkward_meshes()[layout]
    nt = len(tri)
    rng = np.random.default_rng(5)
    verts = np.concatenate([tri.reshape(-1, 3), np.zeros((nt * 3, 3), np.float32)], axis=1).astype(np.float32)
    idx = np.arange(nt * 3, dtype=np.uint32).reshape(-1, 3)
    mats = np.zeros((6, 4), np.float32)
    mats[:, :3] = rng.uniform(0.3, 0.9, (6, 3))
    mats[2, 3] = 4.0
    mids = rng.integers(0, 6, nt).astype(np.int32)
    W, H = 120, 72
    cam = po.camera(60.0, W, H, (0.1, 0.2, 1.0), (0.0, 0.0, 6.0), (0.0, 1.0, 0.0))
    sph, _ = po.reference_spheres_and_materials()
    fx = {"width": W, "height": H, "variant": 1, "normal_mode": 1, "depth": 6, "camera": cam, "vertices": verts,
          "indices": idx, "spheres": sph, "materials": mats, "shading_mode": 1, "material_ids": mids}
    prog = setup_case(tracer, fx)
    images, rays = {}, {}
    for accel in ("brute", "bvh_morton", "bvh_sah", "bvh_ploc", "bvh"):
        if accel != "brute":
            tracer.set_bvh_builder({"bvh_morton": capi.BVH_BUILDER_MORTON, "bvh_sah": capi.BVH_BUILDER_SAH, "bvh_ploc": capi.BVH_BUILDER_PLOC}.get(accel, capi.BVH_BUILDER_BINNED))
            tracer.build_accel()
            tracer.set_bvh_builder(capi.BVH_BUILDER_BINNED)
            tracer.set_accel_mode(capi.ACCEL_BVH)
        else:
            tracer.set_accel_mode(capi.ACCEL_BRUTE)
        tracer.target_clear()
        tracer.reset_stats()
        tracer.dispatch(1, 4, 1)
        tracer.finish()
        images[accel] = tracer.readback()
        rays[accel] = tracer.stats()["rays"]
        if accel != "brute" and nt > 4:   # (up to four triangles are one leaf: pt_build_accel keeps no hierarchy and PT_ACCEL_BVH runs the loop)
            compared, bad_id, bad_t = tracer.verify_accel(3)
            assert compared >= W * H and bad_id == 0 and bad_t == 0
    for accel in ("bvh", "bvh_morton", "bvh_sah", "bvh_ploc"):
        assert n_bit_mismatch(images[accel], images["brute"]) == 0, accel
    assert rays["bvh"] == rays["bvh_morton"] == rays["bvh_sah"] == rays["bvh_ploc"] == rays["brute"] > W * H * 4
    tracer.set_accel_mode(capi.ACCEL_AUTO)
    tracer.set_shading_mode(0)
    tracer.program_destroy(prog)


@pytest.mark.parametrize("limit", [0, 2, 5])
def test_lane_walk_hands_over_when_its_stack_is_full(limit):
    """The per-lane walk keeps at most 24 pending nodes per lane; a lane that would need more repeats its query with the
    per-wave walk. With the limit lowered to 0, 2 or 5 entries that hand-over happens all the time: the image must still
    equal the oracle, and the per-wave counters show that it ran."""
    import os
    from pathtracer_amd import Tracer, capi
    old = os.environ.get("PT_LANE_STACK_LIMIT")
    os.environ["PT_LANE_STACK_LIMIT"] = str(limit)
    try:
        t = Tracer(0)
    finally:
        if old is None:
            del os.environ["PT_LANE_STACK_LIMIT"]
        else:
            os.environ["PT_LANE_STACK_LIMIT"] = old
    try:
        fx = load_case("cornell_materials_128")
        prog = setup_case(t, fx)
        t.build_accel()
        t.set_accel_mode(capi.ACCEL_BVH)
        t.reset_stats()
        t.dispatch(1, 16, 1)
        t.finish()
        assert n_bit_mismatch(t.readback()[..., :3], fx["lp_rgb_16"]) == 0
        st = t.stats()
        assert st["grid_triangle_tests"] > 0 and st["bvh_node_fetches"] > 0
        t.program_destroy(prog)
    finally:
        t.close()


@pytest.mark.parametrize("name", ["cornell_materials_128", "cornell_materials_unit_128", "cornell_materials_tiny_96", "monkey_materials_128"])
@pytest.mark.parametrize("walk", ["lane", "lane_in_step", "wave"])
def test_materials_golden_through_bvh(tracer, wave_walk_tracer, in_step_walk_tracer, walk, name):
    """The llvmpipe goldens of the diffuse/emissive modes, traced through the LBVH with absolute boxes by each of its three
    walks: per lane with the lanes decoupled (default), per lane in step, per wave."""
    from pathtracer_amd import capi
    fx = load_case(name)
    t = {"lane": tracer, "lane_in_step": in_step_walk_tracer, "wave": wave_walk_tracer}[walk]
    prog = setup_case(t, fx)
    t.build_accel()
    t.set_accel_mode(capi.ACCEL_BVH)
    t.reset_stats()
    t.dispatch(1, 16, 1)
    t.finish()
    last = max(int(k) for k in fx["spp"])
    if last != 16:
        t.target_clear()
        t.dispatch(1, last, 1)
        t.finish()
    assert n_bit_mismatch(t.readback()[..., :3], fx[f"lp_rgb_{last}"]) == 0
    st = t.stats()
    assert (st["grid_triangle_tests"] if walk != "wave" else st["bvh_node_fetches"]) > 0
    t.set_accel_mode(capi.ACCEL_AUTO)
    t.set_shading_mode(0)
    t.program_destroy(prog)


def test_materials_mode_state_errors(tracer):
    from pathtracer_amd import PtError, capi
    fx = load_case("cornell_materials_128")
    t = tracer
    prog = setup_case(t, fx)
    with pytest.raises(PtError):
        t.upload_material_ids(fx["material_ids"][:-1])      # one id per triangle
    bad = fx["material_ids"].copy()
    bad[0] = -1
    with pytest.raises(PtError):
        t.upload_material_ids(bad)
    t.upload_mesh(fx["vertices"], fx["indices"])             # a new mesh drops the ids
    t.set_uniforms(fx["camera"], len(fx["indices"]), 10)
    with pytest.raises(PtError):
        t.dispatch(1, 1, 1)
    t.upload_material_ids(fx["material_ids"])
    t.set_accel_mode(capi.ACCEL_CULL)
    with pytest.raises(PtError):
        t.dispatch(1, 1, 1)
    t.set_accel_mode(capi.ACCEL_AUTO)
    t.upload_materials(fx["materials"][:2])                   # ids beyond the table
    with pytest.raises(PtError):
        t.dispatch(1, 1, 1)
    t.upload_materials(fx["materials"])
    t.dispatch(1, 1, 1)
    t.finish()
    t.set_shading_mode(0)
    t.program_destroy(prog)


@pytest.mark.parametrize("name", ["monkey_mesh_128", "world_128", "cornell_geo_128", "cornell_materials_128"])
def test_split_tiles_fold_in_sample_order(tracer, name):
    """From the second dispatch of a window on, expensive tiles are rendered by several blocks (each a range of the
    samples, parked in scratch memory) and folded in sample order afterwards. The image must not change by a bit:
    against the golden in MEAN mode, and against the unsplit first dispatch in SUM mode with a sample stride."""
    fx = load_case(name)
    t = tracer
    prog = setup_case(t, fx)
    t.dispatch(1, 16, 1)            # no costs known yet: one block per tile
    t.finish()
    first = t.readback()
    for _ in range(2):              # costs known: at 128x128 every tile that hits something is split
        t.target_clear()
        t.dispatch(1, 16, 1)
        t.finish()
        img = t.readback()
        assert n_bit_mismatch(img, first) == 0
    assert n_bit_mismatch(img[..., :3], fx["lp_rgb_16"]) == 0
    # continue an image (previous value read by the fold pass), odd sample count
    t.dispatch(17, 5, 1)
    t.finish()
    cont = t.readback()
    t.set_accum_mode(1)
    t.target_clear()
    t.dispatch(2, 7, 3)
    t.finish()
    split_sum = t.readback()
    t.set_window(0, 0, 127, 128)    # a new window forgets the costs: unsplit again
    t.target_clear()
    t.dispatch(2, 7, 3)
    t.finish()
    plain_sum = t.readback()
    assert n_bit_mismatch(split_sum[:, :127], plain_sum[:, :127]) == 0
    assert np.all(split_sum[..., 3] == 7.0)
    t.set_window()
    t.set_accum_mode(0)
    t.target_clear()
    t.dispatch(1, 21, 1)
    t.finish()
    assert n_bit_mismatch(t.readback(), cont) == 0
    t.program_destroy(prog)


def test_ray_count_matches_oracle(tracer):
    fx = load_case("monkey_mesh_128")
    t = tracer
    prog = setup_case(t, fx)
    t.reset_stats()
    t.dispatch(1, 4, 1)
    t.finish()
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    u = po.uniforms(fx["camera"], 128, 128, len(fx["indices"]), 10, 1, 0)
    _, rays = po.render(scene, u, 1, 4, threads=8)
    st = t.stats()
    assert st["rays"] == rays
    assert st["samples"] == 128 * 128 * 4
    t.program_destroy(prog)


def test_random_scene_against_oracle(tracer):
    """Seeded random triangle soup + random camera at an odd resolution, 3 passes, vs the CPU oracle."""
    rng = np.random.default_rng(7)
    nt = 300
    base = rng.uniform(-2, 2, (nt, 1, 3)) + np.array([0, 0, 6.0])
    tri = (base + rng.uniform(-0.7, 0.7, (nt, 3, 3))).astype(np.float32)
    nrm = rng.normal(size=(nt * 3, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    verts = np.concatenate([tri.reshape(-1, 3), nrm], axis=1).astype(np.float32)
    idx = np.arange(nt * 3, dtype=np.uint32).reshape(-1, 3)
    W, H = 173, 61
    cam = po.camera(50.0, W, H, (0.3, -0.2, -1.0), (0.1, 0.0, 6.0), (0.1, 1.0, 0.0))
    sph, mat = po.reference_spheres_and_materials()
    for nm in (0, 1):
        fx = {"width": W, "height": H, "variant": 1, "normal_mode": nm, "depth": 10, "camera": cam,
              "vertices": verts, "indices": idx, "spheres": sph, "materials": mat}
        prog = setup_case(tracer, fx)
        ids, ts = tracer.readback_primary()
        tracer.dispatch(1, 3, 1)
        tracer.finish()
        img = tracer.readback()
        scene = po.Scene(verts, idx, sph, mat)
        u = po.uniforms(cam, W, H, nt, 10, 1, nm)
        oids, ots, _ = po.primary(scene, u, threads=8)
        ref, _ = po.render(scene, u, 1, 3, threads=8)
        assert np.array_equal(ids, oids)
        assert (oids >= 0).sum() > 500
        assert n_bit_mismatch(ts, ots) == 0
        assert n_bit_mismatch(img, ref) == 0
        tracer.program_destroy(prog)


def test_triangle_prefix_and_camera_move(tracer):
    """u_NumTriangles below the uploaded count traces the first triangles only (the shader's loop bound, fs:129); and a
    camera move rebuilds the per-triangle records, including the list of triangles the new position can see at all.
    Both against the CPU oracle, bit for bit."""
    fx = load_case("monkey_mesh_128")
    t = tracer
    prog = setup_case(t, fx)
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    half = len(fx["indices"]) // 2 + 3
    t.set_uniforms(fx["camera"], half, 10)
    ids, ts = t.readback_primary()
    t.dispatch(1, 3, 1)
    t.finish()
    u = po.uniforms(fx["camera"], 128, 128, half, 10, 1, 0)
    oids, ots, _ = po.primary(scene, u, threads=8)
    ref, _ = po.render(scene, u, 1, 3, threads=8)
    assert np.array_equal(ids, oids) and n_bit_mismatch(ts, ots) == 0
    assert n_bit_mismatch(t.readback(), ref) == 0
    for eye, target in [((1.5, 0.8, 1.0), (0.0, 0.0, 3.0)), ((0.0, 0.2, 6.0), (0.0, 0.0, 3.0)), ((0.05, 0.1, 3.0), (0.0, 0.0, 4.0))]:
        cam = po.camera(60.0, 128, 128, eye, target, (0.0, 1.0, 0.0))     # from the side, from behind, from inside the head
        t.set_uniforms(cam, len(fx["indices"]), 10)
        t.target_clear()
        ids, ts = t.readback_primary()
        t.dispatch(1, 3, 1)
        t.finish()
        u = po.uniforms(cam, 128, 128, len(fx["indices"]), 10, 1, 0)
        oids, ots, _ = po.primary(scene, u, threads=8)
        ref, _ = po.render(scene, u, 1, 3, threads=8)
        # inside the head every triangle shows its back: nothing can be hit, the loop has no triangle left to visit
        assert (oids >= 0).sum() > 4000 or eye[2] == 3.0 and (oids >= 0).sum() == 0
        assert np.array_equal(ids, oids) and n_bit_mismatch(ts, ots) == 0
        assert n_bit_mismatch(t.readback(), ref) == 0
    t.program_destroy(prog)


def test_window_and_sum_mode(tracer):
    """Tile split: two half-frame windows equal the full frame bit for bit. Sample split: two residue classes
    summed equal the sequential mean to rounding."""
    fx = load_case("cornell_geo_128")
    t = tracer
    prog = setup_case(t, fx)
    t.dispatch(1, 4, 1)
    t.finish()
    full = t.readback()
    t.target_clear()
    t.set_window(0, 0, 128, 50)
    t.dispatch(1, 4, 1)
    t.finish()
    top = t.readback()
    t.target_clear()
    t.set_window(0, 50, 128, 78)
    t.dispatch(1, 4, 1)
    t.finish()
    bottom = t.readback()
    assert n_bit_mismatch(top[:50], full[:50]) == 0 and n_bit_mismatch(bottom[50:], full[50:]) == 0
    assert np.all(top[50:, :, :3] == 0) and np.all(bottom[:50, :, :3] == 0)
    # sample split
    t.set_window()
    t.set_accum_mode(1)
    sums = []
    for r in range(2):
        t.target_clear()
        t.dispatch(1 + r, 2, 2)
        t.finish()
        sums.append(t.readback())
    merged = (sums[0] + sums[1])[..., :3] * np.float32(0.25)
    assert np.abs(merged - full[..., :3]).max() < 2e-6
    assert np.all(sums[0][..., 3] == 2.0)
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    u = po.uniforms(fx["camera"], 128, 128, len(fx["indices"]), 10, 1, 1)
    osum, _ = po.render_sum(scene, u, 2, 2, 2, threads=8)
    assert n_bit_mismatch(sums[1][..., :3], osum[..., :3]) == 0
    t.set_accum_mode(0)
    t.program_destroy(prog)


def test_error_paths(tracer):
    from pathtracer_amd import PtError
    t = tracer
    with pytest.raises(PtError):
        t.upload_spheres(np.zeros((3, 4), np.float32))
    with pytest.raises(PtError):
        t.upload_mesh(np.zeros((3, 6), np.float32), np.array([[0, 1, 3]], np.uint32))
    with pytest.raises(PtError):
        t.use_program(12345)
    with pytest.raises(PtError):
        t.dispatch(0, 1, 1)


def test_cpp_host_render_loop(tmp_path):
    """The C++ host (Shader + scene construction + render loop, pathtracer_amd/host) end to end: monkeyhead2 at
    128x128 with the reference's camera, 16 frames, one dispatch per frame; the Framebuffer it fills must hold the
    golden image (accumulationBuffer = mean x frames, colorBuffer = display colour in B,G,R order)."""
    import os
    from pathtracer_amd import host_api
    from pathtracer_amd.scenes import ASSETS
    fx = load_case("monkey_mesh_128")
    tga = str(tmp_path / "out.tga")
    rgb_sum, rgba8, st = host_api.render(os.path.join(ASSETS, "monkeyhead2.obj"), 128, 128, 16, samples_per_dispatch=1,
                                         program_kind=1, output_tga=tga)
    ref = fx["lp_rgb_16"]
    assert n_bit_mismatch(rgb_sum, ref * np.float32(16.0)) == 0
    want = (np.float32(255.49) * np.clip(np.sqrt(ref), 0, 1)).astype(np.uint8)
    assert np.array_equal(rgba8[..., 0], want[..., 2]) and np.array_equal(rgba8[..., 2], want[..., 0])
    assert np.all(rgba8[..., 3] == 255)
    assert st["samples"] == 128 * 128 * 16 and st["rays"] >= st["samples"]
    data = open(tga, "rb").read()
    assert len(data) == 18 + 128 * 128 * 4 and data[2] == 2 and data[16] == 32
    # batching frames into one dispatch gives the same picture
    rgb_sum2, _, _ = host_api.render(os.path.join(ASSETS, "monkeyhead2.obj"), 128, 128, 16, samples_per_dispatch=16, program_kind=1)
    assert n_bit_mismatch(rgb_sum2, rgb_sum) == 0


@pytest.mark.parametrize("accel", ["loop", "auto"])
def test_cpp_host_materials_mode(accel):
    """C++ host end to end in the diffuse/emissive mode: OBJ + MTL -> material table and per-triangle ids -> 16 frames;
    the Framebuffer must hold the llvmpipe golden of the Cornell box."""
    import os
    from pathtracer_amd import host_api
    from pathtracer_amd.scenes import ASSETS
    fx = load_case("cornell_materials_128")
    rgb_sum, _, st = host_api.render(os.path.join(ASSETS, "cornell_box.obj"), 128, 128, 16, samples_per_dispatch=4, program_kind=1,
                                     mtl_basedir=ASSETS + "/", shading="materials", accel=accel,
                                     camera_eye_target_fov=((278, 273, -800), (278, 273, 0), 39.3), translate_z=0.0)
    assert n_bit_mismatch(rgb_sum, fx["lp_rgb_16"] * np.float32(16.0)) == 0
    assert st["samples"] == 128 * 128 * 16


def test_c1_full_size_against_llvmpipe_digests(tracer):
    """Survey config C1 at full size: cornell_box.obj through the C++ host loader, 512x512, passes 1..16. The golden holds
    SHA-256 digests of the RGBA32F image the reference shader produced on llvmpipe after 1, 4 and 16 passes
    (oracle/time_c1.py); the device image must hash to the same bytes, frame by frame and as one 16-sample dispatch."""
    import hashlib
    import json
    import os
    from helpers import GOLDEN
    from pathtracer_amd.scenes import SceneData
    gold = json.load(open(os.path.join(GOLDEN, "c1_cornell_512.json")))
    scene = SceneData("cornell", 512, 512)
    assert n_bit_mismatch(scene.camera, np.asarray(gold["camera"], np.float32)) == 0
    t = tracer
    prog = scene.upload(t)
    t.set_accel_mode(0)
    t.set_accum_mode(0)
    t.target_create(512, 512)
    digest = lambda: hashlib.sha256(t.readback().tobytes()).hexdigest()
    done = 0
    for k in (1, 4, 16):
        for s in range(done + 1, k + 1):
            t.dispatch(s, 1, 1)
        t.finish()
        done = k
        assert digest() == gold["sha256"][str(k)], f"after {k} passes"
    for _ in range(2):   # second time: from a work list with split tiles
        t.target_clear()
        t.dispatch(1, 16, 1)
        t.finish()
        assert digest() == gold["sha256"]["16"]
    t.program_destroy(prog)


@pytest.mark.parametrize("accel", [0, 2])
def test_c1_materials_against_llvmpipe_digests(tracer, accel):
    """Config C1's size in the diffuse/emissive mode: cornell_box.obj with the table the C++ host derives from cornell_box.mtl,
    512x512, passes 1..16, against SHA-256 digests of what the reference shader with the recorded edits rendered on llvmpipe
    (oracle/time_c1.py --materials) -- through the all-triangles loop (what PT_ACCEL_AUTO picks for 36 triangles) and through
    the LBVH's per-lane walk; frame by frame, then as one dispatch, then from a work list."""
    import hashlib
    import json
    import os
    from helpers import GOLDEN
    from pathtracer_amd.scenes import SceneData
    gold = json.load(open(os.path.join(GOLDEN, "c1_cornell_512_materials.json")))
    scene = SceneData("cornell", 512, 512)
    assert n_bit_mismatch(scene.camera, np.asarray(gold["camera"], np.float32)) == 0
    t = tracer
    prog = scene.upload(t, shading="materials")
    t.set_accel_mode(accel)
    t.set_accum_mode(0)
    t.target_create(512, 512)
    digest = lambda: hashlib.sha256(t.readback().tobytes()).hexdigest()
    done = 0
    for k in (1, 4, 16):
        for s in range(done + 1, k + 1):
            t.dispatch(s, 1, 1)
        t.finish()
        done = k
        assert digest() == gold["sha256"][str(k)], f"after {k} passes"
    for _ in range(2):
        t.target_clear()
        t.dispatch(1, 16, 1)
        t.finish()
        assert digest() == gold["sha256"]["16"]
    t.set_accel_mode(0)
    t.set_shading_mode(0)
    t.program_destroy(prog)


def test_c2_benchmark_configuration_against_llvmpipe_digests(tracer):
    """The benchmark workload itself (cornell_box.obj, 1920x1080, passes 1..256, what bench.py times): SHA-256 digests of the
    RGBA32F image the reference shader produced on llvmpipe after 1, 16, 64 and 256 passes (oracle/time_c1.py --c2). The
    device image must hash to the same bytes when the passes are dispatched in four ranges, and when all 256 samples are one
    dispatch as in the benchmark (twice: the second time from a work list with split tiles)."""
    import hashlib
    import json
    import os
    from helpers import GOLDEN
    from pathtracer_amd.scenes import SceneData
    gold = json.load(open(os.path.join(GOLDEN, "c2_cornell_1080p.json")))
    scene = SceneData("cornell")
    assert (scene.width, scene.height) == (1920, 1080)
    assert n_bit_mismatch(scene.camera, np.asarray(gold["camera"], np.float32)) == 0
    t = tracer
    prog = scene.upload(t)
    t.set_accel_mode(0)
    t.set_accum_mode(0)
    t.target_create(1920, 1080)
    digest = lambda: hashlib.sha256(t.readback().tobytes()).hexdigest()
    done = 0
    for k in (1, 16, 64, 256):
        t.dispatch(done + 1, k - done, 1)
        t.finish()
        done = k
        assert digest() == gold["sha256"][str(k)], f"after {k} passes"
    for _ in range(2):
        t.target_clear()
        t.dispatch(1, 256, 1)
        t.finish()
        assert digest() == gold["sha256"]["256"]
    t.program_destroy(prog)


def test_long_first_dispatch_probes_then_schedules():
    """A long dispatch that has no tile costs to go by (new context; later: new camera position) renders its first samples on
    their own and schedules the rest from their costs. Same image as the llvmpipe digest / as frame-by-frame dispatches, and
    it still counts as one dispatch."""
    import hashlib
    import json
    import os
    from helpers import GOLDEN
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    gold = json.load(open(os.path.join(GOLDEN, "c2_cornell_1080p.json")))
    scene = SceneData("cornell")
    t = Tracer(0)
    try:
        scene.upload(t)
        t.set_accum_mode(0)
        t.target_create(1920, 1080)
        t.dispatch(1, 256, 1)
        t.finish()
        assert hashlib.sha256(t.readback().tobytes()).hexdigest() == gold["sha256"]["256"]
        assert t.stats()["dispatches"] == 1
        assert t.stats()["samples"] == 1920 * 1080 * 256
        # another camera position: the costs are stale, the next long dispatch probes again
        small = SceneData("cornell", 320, 200)
        cam = small.camera.copy()
        cam[0] += 40.0
        cam[3] += 40.0  # eye and lower-left corner moved together
        t.target_create(320, 200)
        t.set_uniforms(cam, small.num_triangles, 10)
        t.dispatch(1, 64, 1)
        t.finish()
        whole = t.readback()
        t.target_clear()
        for s in range(1, 65):
            t.dispatch(s, 1, 1)
        t.finish()
        assert n_bit_mismatch(whole, t.readback()) == 0
    finally:
        t.close()


def test_sparse_view_with_eager_split_equals_frame_by_frame():
    """monkeyhead2 at 1920x1080: the work sits in a few hundred tiles, so after the first list the scheduler splits eagerly
    (many pieces per tile, parked samples folded afterwards). Three 64-sample dispatches -- probe + first list, rebuilt list
    with the eager factor, reused list -- must each give the image of 64 single-sample dispatches, and the reference's ray
    count."""
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    scene = SceneData("monkeyhead2")
    t = Tracer(0)
    try:
        scene.upload(t)
        t.set_accum_mode(0)
        t.target_create(scene.width, scene.height)
        for s in range(1, 65):
            t.dispatch(s, 1, 1)
        t.finish()
        frames = t.readback()
        rays = t.stats()["rays"]
        for _ in range(3):
            t.target_clear()
            t.reset_stats()
            t.dispatch(1, 64, 1)
            t.finish()
            assert n_bit_mismatch(t.readback(), frames) == 0
            assert t.stats()["rays"] == rays
    finally:
        t.close()


@pytest.mark.parametrize("block,count", [(16, 3), (5, 2), (8, 8)])
def test_row_interleave_partition_is_exact(tracer, block, count):
    """Image-tile partition in one dispatch per renderer: each renders the row blocks `index` modulo `count`; its rows are bit
    for bit those of the whole-frame render, everything else stays at the clear value, the parts sum to the whole frame, and
    the sample counts add up. Twice per part, so that the second dispatch runs from a work list."""
    from pathtracer_amd import dist as ptdist
    fx = load_case("monkey_mesh_128")
    t = tracer
    prog = setup_case(t, fx)
    H = 128
    t.dispatch(1, 6, 1)
    t.finish()
    full = t.readback()
    total = np.zeros_like(full)
    samples = 0
    for index in range(count):
        t.set_row_interleave(block, index, count)
        for _ in range(2):
            t.target_clear()
            t.reset_stats()
            t.dispatch(1, 6, 1)
            t.finish()
        part = t.readback()
        own = np.zeros(H, bool)
        for y0, h in ptdist.row_blocks(index, count, H, block):
            own[y0:y0 + h] = True
        assert n_bit_mismatch(part[own], full[own]) == 0
        assert np.all(part[~own][..., :3] == 0)
        total[own] = part[own]
        samples += t.stats()["samples"]
    t.set_row_interleave(block, 0, 1)
    assert n_bit_mismatch(total, full) == 0
    assert samples == 128 * 128 * 6
    t.program_destroy(prog)


@pytest.mark.parametrize("workload", ["monkeyhead2", "suzanne"])
def test_sample_split_over_eight_at_full_size(workload):
    """Configuration C3's partition at its own size: monkeyhead2 and suzanne at 1920x1080, 1024 samples as eight residue classes
    of 128 (what eight GPUs render), summed and scaled, against the sequential running mean of the same 1024 samples: equal to
    rounding (|difference| < 2e-6 on values in [0, 1]); ray counts of the parts add up to the whole."""
    from pathtracer_amd import Tracer, capi
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(workload)
    t = Tracer(0)
    try:
        scene.upload(t)
        t.target_create(scene.width, scene.height)
        t.set_accum_mode(capi.ACCUM_MEAN)
        t.dispatch(1, 1024, 1)
        t.finish()
        mean = t.readback()[..., :3].copy()
        rays = t.stats()["rays"]
        t.reset_stats()
        t.set_accum_mode(capi.ACCUM_SUM)
        total = np.zeros_like(mean)
        for r in range(8):
            t.target_clear()
            t.dispatch(1 + r, 128, 8)
            t.finish()
            total += t.readback()[..., :3]      # fp32 additions, like the all-reduce
        assert t.stats()["rays"] == rays
        assert total.dtype == np.float32
        assert np.abs(total * np.float32(1.0 / 1024.0) - mean).max() < 2e-6
    finally:
        t.close()


@pytest.mark.parametrize("name,workload", [("monkey_materials_1080p_crop", "monkeyhead2"), ("suzanne_materials_1080p_crop", "suzanne")])
def test_config3_in_materials_mode_sample_split_over_eight(name, workload):
    """Configuration C3 as the benchmark words it -- monkeyhead2.obj / suzanne.obj WITH materials, 1920x1080, 1024 samples, sample
    split -- through the kernel that mode runs on (hierarchy walk per lane, lanes decoupled). The whole frame after 1 and 4
    sequential passes holds, in the golden's crop, exactly what the reference shader (with the recorded material edits) rendered
    there on llvmpipe; the 1024 samples as eight strided residue classes of 128 in running-sum mode (what eight GPUs render), added
    in fp32 and scaled, equal the sequential running mean to rounding (|difference| < 2e-6 on radiance of order 1); the parts' ray
    counts add up to the whole's."""
    from pathtracer_amd import Tracer, capi
    from pathtracer_amd.scenes import SceneData
    fx = load_case(name)
    scene = SceneData(workload)
    assert (scene.width, scene.height) == (1920, 1080) == (int(fx["width"]), int(fx["height"]))
    assert n_bit_mismatch(scene.vertices, fx["vertices"]) == 0 and np.array_equal(scene.indices, fx["indices"])
    assert scene.normal_mode == int(fx["normal_mode"])
    x0, y0, w, h = (int(v) for v in fx["crop"])
    with Tracer(0) as t:
        scene.upload(t, shading="materials")
        t.upload_materials(fx["materials"])          # the golden's table and per-triangle rows (the files bring no materials)
        t.upload_material_ids(fx["material_ids"])
        t.target_create(scene.width, scene.height)
        t.set_accum_mode(capi.ACCUM_MEAN)
        done = 0
        for k in (1, 4):
            t.dispatch(done + 1, k - done, 1)
            t.finish()
            done = k
            assert n_bit_mismatch(t.readback()[y0:y0 + h, x0:x0 + w, :3], fx[f"lp_rgb_{k}"]) == 0, f"crop after {k} passes"
        t.dispatch(5, 1020, 1)
        t.finish()
        mean = t.readback()[..., :3].copy()
        st = t.stats()
        rays = st["rays"]
        assert st["lane_walk_records"] > 0 and st["samples"] == 1920 * 1080 * 1024   # the decoupled walk rendered it
        t.reset_stats()
        t.set_accum_mode(capi.ACCUM_SUM)
        total = np.zeros_like(mean)
        for r in range(8):
            t.target_clear()
            t.dispatch(1 + r, 128, 8)
            t.finish()
            total += t.readback()[..., :3]      # fp32 additions, like the all-reduce
        assert t.stats()["rays"] == rays
        scaled = total * np.float32(1.0 / 1024.0)
        assert np.abs(scaled - mean).max() < 2e-6 * max(1.0, float(mean.max()))
        # and the merged image still agrees with llvmpipe's 4-pass crop where the two can be compared: the pixels whose every
        # sample is the same (sky, emitters seen directly) -- the 1024-sample mean of a constant is that constant
        ref = fx["lp_rgb_4"]
        const = (fx["lp_rgb_1"] == ref).all(-1) & (fx["lp_id"] < 0)
        assert const.sum() > 500 and np.abs(scaled[y0:y0 + h, x0:x0 + w][const] - ref[const]).max() < 2e-6


def test_resume_from_an_uploaded_image(tracer):
    """The accumulation image and the sample index are the whole state of a progressive render: passes 1..8, read back, a NEW
    context, pt_upload_rgba32f, passes 9..16 -- the same bits as passes 1..16 in one go and as llvmpipe's 16-pass image."""
    from pathtracer_amd import Tracer
    for name in ("monkey_mesh_128", "cornell_materials_128"):
        fx = load_case(name)
        t = tracer
        prog = setup_case(t, fx)
        t.dispatch(1, 8, 1)
        t.finish()
        saved = t.readback().copy()
        t.program_destroy(prog)
        with Tracer(0) as t2:
            setup_case(t2, fx)
            t2.upload_image(saved)
            assert n_bit_mismatch(t2.readback(), saved) == 0
            t2.dispatch(9, 8, 1)
            t2.finish()
            assert n_bit_mismatch(t2.readback()[..., :3], fx["lp_rgb_16"]) == 0
            # frame by frame from the upload as well (each pass reads the image the one before wrote)
            t2.upload_image(saved)
            for s_ in range(9, 17):
                t2.dispatch(s_, 1, 1)
            t2.finish()
            assert n_bit_mismatch(t2.readback()[..., :3], fx["lp_rgb_16"]) == 0
            with pytest.raises(ValueError):
                t2.upload_image(saved[:-1])


def test_c5_at_its_full_length():
    """Configuration C5 at its own size and length: the ~1M-triangle instanced scene at 3840x2160 with 4096 samples. One dispatch
    of 4096 (probe, work list, split tiles, fold) against sixteen dispatches of 256: the same image bit for bit and the same ray
    count; then the window of a 64x64 crop with all 4096 samples through the direction grid and through the hierarchy walk: the
    same bits again, and equal to the crop of the full frame."""
    from pathtracer_amd import Tracer, capi
    from pathtracer_amd.scenes import SceneData
    scene = SceneData("instanced")
    t = Tracer(0)
    try:
        scene.upload(t)
        W, H = scene.width, scene.height
        t.target_create(W, H)
        t.dispatch(1, 4096, 1)
        t.finish()
        whole = t.readback().copy()
        rays = t.stats()["rays"]
        t.reset_stats()
        t.target_clear()
        for k in range(16):
            t.dispatch(1 + 256 * k, 256, 1)
        t.finish()
        assert n_bit_mismatch(t.readback(), whole) == 0
        assert t.stats()["rays"] == rays > W * H * 4096
        x0, y0 = 1900, 1000
        crops = {}
        for accel in (capi.ACCEL_GRID, capi.ACCEL_BVH):
            t.set_accel_mode(accel)
            t.set_window(x0, y0, 64, 64)
            t.target_clear()
            t.dispatch(1, 4096, 1)
            t.finish()
            crops[accel] = t.readback()[y0:y0 + 64, x0:x0 + 64].copy()
        assert n_bit_mismatch(crops[capi.ACCEL_GRID], crops[capi.ACCEL_BVH]) == 0
        assert n_bit_mismatch(crops[capi.ACCEL_GRID], whole[y0:y0 + 64, x0:x0 + 64]) == 0
        assert (crops[capi.ACCEL_GRID][..., :3] != crops[capi.ACCEL_GRID][0, 0, :3]).any()   # the crop is not all one colour
    finally:
        t.close()


CONFIG4 = ["torus_geo_128", "torus_distorted_geo_128", "torus_distorted_geo_256x144", "torus_distorted_4k_crop"]


@pytest.mark.parametrize("accel", [1, 2, 4])
@pytest.mark.parametrize("name", CONFIG4)
def test_config4_goldens_through_every_search_mode(tracer, name, accel):
    """The benchmark's fourth configuration (catmark torus + its distorted copy) against what the reference shader rendered on llvmpipe,
    through the all-triangles loop, the LBVH and the direction grid: ids, distances and radiance bit for bit, the latter from one
    multi-sample dispatch per checkpoint."""
    fx = load_case(name)
    t = tracer
    prog = setup_case(t, fx)
    t.build_accel()
    t.set_accel_mode(accel)
    x0, y0, w, h = (int(v) for v in fx["crop"])
    t.set_window(x0, y0, w, h)
    win = (slice(y0, y0 + h), slice(x0, x0 + w))
    ids, ts = t.readback_primary()
    assert np.array_equal(ids[win], fx["lp_id"])
    hit = fx["lp_id"] >= 0
    assert hit.any() and (~hit).any()
    assert n_bit_mismatch(ts[win][hit], fx["lp_t"][hit]) == 0
    done = 0
    for k in sorted(int(s) for s in fx["spp"]):
        t.dispatch(done + 1, k - done, 1)
        t.finish()
        done = k
        assert n_bit_mismatch(t.readback()[win][..., :3], fx[f"lp_rgb_{k}"]) == 0, f"radiance after {k} passes"
    t.set_accel_mode(0)
    t.set_window()
    t.program_destroy(prog)


@pytest.mark.parametrize("count", [2, 4, 8])
def test_config4_row_partition_at_full_size(count):
    """Configuration 4 as it is meant to run: the torus and its distorted copy at 3840x2160, the rows shared by 2, 4 or 8
    renderers in interleaved blocks of 16 (pt_set_row_interleave; what `bench.py --partition tile` gives each GPU). Every
    part's rows are bit for bit those of the single-renderer frame, the parts together are that frame, and its crop is what
    the reference shader rendered there on llvmpipe."""
    from pathtracer_amd import Tracer, dist as ptdist
    from pathtracer_amd.scenes import SceneData
    fx = load_case("torus_distorted_4k_crop")
    scene = SceneData("torus_distorted")
    assert (scene.width, scene.height) == (3840, 2160)
    assert n_bit_mismatch(scene.vertices, fx["vertices"]) == 0 and np.array_equal(scene.indices, fx["indices"])
    W, H, spp = scene.width, scene.height, 4
    t = Tracer(0)
    try:
        scene.upload(t)
        t.target_create(W, H)
        t.dispatch(1, spp, 1)
        t.finish()
        full = t.readback()
        x0, y0, w, h = (int(v) for v in fx["crop"])
        assert n_bit_mismatch(full[y0:y0 + h, x0:x0 + w, :3], fx["lp_rgb_4"]) == 0
        merged = np.zeros_like(full)
        samples = 0
        for index in range(count):
            t.set_row_interleave(16, index, count)
            t.target_clear()
            t.reset_stats()
            t.dispatch(1, spp, 1)
            t.finish()
            part = t.readback()
            own = np.zeros(H, bool)
            for r0, rh in ptdist.row_blocks(index, count, H, 16):
                own[r0:r0 + rh] = True
            assert np.all(part[~own][..., :3] == 0)
            merged[own] = part[own]
            samples += t.stats()["samples"]
        assert n_bit_mismatch(merged, full) == 0
        assert samples == W * H * spp
    finally:
        t.close()
