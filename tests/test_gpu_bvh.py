"""The hierarchy (either builder's) must not change a single result: closest-hit ids, distances and radiance through the BVH
equal the all-triangles loop (the shader's own formulation) bit for bit -- including exact ties, which the shader's
ascending loop with a strict '<' gives to the lowest triangle index."""
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


def both_modes(t, verts, idx, cam, W, H, normal_mode, spp=3, window=None):
    from pathtracer_amd import capi
    sph, mat = po.reference_spheres_and_materials()
    prog = t.program_create(capi.PROGRAM_MESH)
    t.use_program(prog)
    t.upload_spheres(sph)
    t.upload_mesh(verts, idx)
    t.set_normal_mode(normal_mode)
    t.set_accum_mode(capi.ACCUM_MEAN)
    t.target_create(W, H)
    if window:
        t.set_window(*window)
    t.set_uniforms(cam, len(idx), 10)
    t.build_accel()
    res = {}
    for name, mode in (("brute", capi.ACCEL_BRUTE), ("bvh", capi.ACCEL_BVH), ("cull", capi.ACCEL_CULL), ("grid", capi.ACCEL_GRID)):
        t.set_accel_mode(mode)
        t.target_clear()
        t.reset_stats()
        ids, ts = t.readback_primary()
        if spp:
            t.dispatch(1, spp, 1)
            t.finish()
        res[name] = (ids, ts, t.readback(), t.stats()["rays"])
    # the hierarchies of the other three builders (the default is the device's binned surface-area builder)
    for name, builder in (("bvh_morton", capi.BVH_BUILDER_MORTON), ("bvh_sah", capi.BVH_BUILDER_SAH), ("bvh_ploc", capi.BVH_BUILDER_PLOC)):
        t.set_bvh_builder(builder)
        t.build_accel()
        t.set_bvh_builder(capi.BVH_BUILDER_BINNED)
        t.set_accel_mode(capi.ACCEL_BVH)
        t.target_clear()
        t.reset_stats()
        ids, ts = t.readback_primary()
        if spp:
            t.dispatch(1, spp, 1)
            t.finish()
        res[name] = (ids, ts, t.readback(), t.stats()["rays"])
    t.set_accel_mode(capi.ACCEL_AUTO)
    t.program_destroy(prog)
    return res


def assert_same(res):
    b = res["brute"]
    for other in ("bvh", "bvh_morton", "bvh_sah", "bvh_ploc", "cull", "grid"):
        a = res[other]
        assert np.array_equal(a[0], b[0]), f"{other}: {(a[0] != b[0]).sum()} primary ids differ"
        assert n_bit_mismatch(a[1], b[1]) == 0, other
        assert n_bit_mismatch(a[2], b[2]) == 0, other
        assert a[3] == b[3], other


@pytest.mark.parametrize("name", ["monkey_mesh_128", "cornell_geo_128", "suzanne_geo_96"])
def test_bvh_equals_brute_force_on_golden_scenes(tracer, name):
    fx = load_case(name)
    W, H = int(fx["width"]), int(fx["height"])
    res = both_modes(tracer, fx["vertices"], fx["indices"], fx["camera"], W, H, int(fx["normal_mode"]), spp=2)
    assert_same(res)
    # and both equal what the reference shader produced
    assert np.array_equal(res["bvh"][0], fx["lp_id"])
    assert n_bit_mismatch(res["bvh"][2][..., :3], fx["lp_rgb_2"]) == 0


def test_exact_ties_go_to_the_lowest_index(tracer):
    """Every triangle twice (ids k and k+N share geometry), in shuffled order: each hit is an exact tie."""
    fx = load_case("monkey_mesh_128")
    idx = fx["indices"]
    rng = np.random.default_rng(3)
    doubled = np.concatenate([idx, idx])[rng.permutation(2 * len(idx))]
    res = both_modes(tracer, fx["vertices"], doubled, fx["camera"], 128, 128, 0, spp=2)
    assert_same(res)
    ids = res["bvh"][0]
    hit = ids >= 0
    assert hit.sum() > 3000
    # the winner is the first occurrence of that triangle in the index list
    first_pos = {}
    for pos, tri in enumerate(map(tuple, doubled)):
        first_pos.setdefault(tri, pos)
    assert all(first_pos[tuple(doubled[i])] == i for i in np.unique(ids[hit]))
    scene = po.Scene(fx["vertices"], doubled, fx["spheres"], fx["materials"])
    u = po.uniforms(fx["camera"], 128, 128, len(doubled), 10, 1, 0)
    oids, _, _ = po.primary(scene, u, threads=8)
    assert np.array_equal(ids, oids)


def test_random_soup_and_degenerate_layouts(tracer):
    rng = np.random.default_rng(11)
    W, H = 200, 120
    cam = po.camera(55.0, W, H, (0.2, 0.1, -2.0), (0.0, 0.0, 5.0), (0.0, 1.0, 0.0))
    # (a) random soup, (b) all triangles in one plane sharing edges (grid), (c) many coincident centroids
    nt = 5000
    base = rng.uniform(-3, 3, (nt, 1, 3)) + np.array([0, 0, 7.0])
    soup = (base + rng.uniform(-0.5, 0.5, (nt, 3, 3))).astype(np.float32)
    g = 40
    xs, ys = np.meshgrid(np.linspace(-3, 3, g + 1), np.linspace(-2, 2, g + 1))
    P = np.stack([xs, ys, np.full_like(xs, 6.0)], -1).astype(np.float32)
    quads = []
    for j in range(g):
        for i in range(g):
            a, b, c, d = P[j, i], P[j, i + 1], P[j + 1, i + 1], P[j + 1, i]
            quads += [[a, c, b], [a, d, c]]
    grid = np.array(quads, np.float32)
    fan = np.array([[[0, 0, 5.0], [np.cos(t + 0.3) * 2, np.sin(t + 0.3) * 2, 5.0 + 0.01 * k], [np.cos(t) * 2, np.sin(t) * 2, 5.0 + 0.01 * k]]
                    for k, t in enumerate(np.linspace(0, 6.28, 300))], np.float32)
    for tris in (soup, grid, fan):
        verts = np.concatenate([tris.reshape(-1, 3), np.zeros((tris.shape[0] * 3, 3), np.float32)], 1)
        idx = np.arange(len(verts), dtype=np.uint32).reshape(-1, 3)
        res = both_modes(tracer, verts, idx, cam, W, H, 1, spp=2)
        assert_same(res)
        assert (res["bvh"][0] >= 0).sum() > 1000


def test_culling_is_conservative_in_awkward_views(tracer):
    """Views that stress the wave-level cull and the padded boxes: camera inside the mesh's bounding box, camera lying
    in the plane of many triangles, needle-thin triangles, and a very wide field of view (large ray bundles)."""
    rng = np.random.default_rng(23)
    g = 24
    xs, zs = np.meshgrid(np.linspace(-4, 4, g + 1), np.linspace(0.5, 9, g + 1))
    floor = []
    for j in range(g):
        for i in range(g):
            a, b, c, d = [np.array([xs[j + dj, i + di], 0.0, zs[j + dj, i + di]]) for dj, di in ((0, 0), (0, 1), (1, 1), (1, 0))]
            floor += [[a, b, c], [a, c, d]]                      # seen from above
    floor = np.array(floor, np.float32)
    needles = np.array([[[x, -1.0, 5.0], [x + 1e-3, -1.0, 5.0], [x + 5e-4, 2.0, 5.0 + 0.3 * k]] for k, x in enumerate(np.linspace(-3, 3, 200))],
                       np.float32)[:, ::-1].copy()
    nt = 800
    base = rng.uniform(-2.5, 2.5, (nt, 1, 3)) + np.array([0, 0, 1.0])
    around = (base + rng.uniform(-0.6, 0.6, (nt, 3, 3))).astype(np.float32)   # surrounds the camera
    tris = np.concatenate([floor, needles, around])
    verts = np.concatenate([tris.reshape(-1, 3), np.zeros((len(tris) * 3, 3), np.float32)], 1)
    idx = np.arange(len(verts), dtype=np.uint32).reshape(-1, 3)
    views = [(100.0, (0.0, 1.0, 0.0), (0.0, 0.2, 5.0)),     # above the floor, very wide
             (70.0, (0.1, 0.0, 0.2), (0.0, 0.0, 6.0)),      # eye exactly in the floor's plane
             (140.0, (0.0, 0.5, 1.0), (0.3, 0.4, 4.0))]     # inside the cloud, extreme field of view
    for fov, eye, target in views:
        cam = po.camera(fov, 160, 96, eye, target, (0.0, 1.0, 0.0))
        res = both_modes(tracer, verts, idx, cam, 160, 96, 1, spp=2)
        assert_same(res)
        assert (res["brute"][0] >= 0).sum() > 500
        scene = po.Scene(verts, idx, *po.reference_spheres_and_materials())
        u = po.uniforms(cam, 160, 96, len(idx), 10, 1, 1)
        oids, _, _ = po.primary(scene, u, threads=8)
        assert np.array_equal(res["cull"][0], oids)


def test_million_triangle_scene_crops(tracer):
    """~1M triangles (254 suzanne instances): BVH vs all-triangles loop on crops of the 4K frame."""
    from pathtracer_amd.scenes import SceneData
    sc = SceneData("instanced")
    assert 990_000 < sc.num_triangles < 1_010_000
    for window in [(1900, 1060, 64, 64), (1235, 520, 64, 48), (2430, 1470, 48, 64)]:
        res = both_modes(tracer, sc.vertices, sc.indices, sc.camera, sc.width, sc.height, sc.normal_mode, spp=1, window=window)
        assert_same(res)
        x0, y0, w, h = window
        assert (res["bvh"][0][y0:y0 + h, x0:x0 + w] >= 0).sum() > 100


@pytest.mark.parametrize("workload,size,spp", [("instanced", (1920, 1080), 8), ("instanced", (3840, 2160), 16), ("cornell", (960, 540), 32),
                                                ("suzanne", (640, 360), 48), ("suzanne", (1920, 1080), 64), ("monkeyhead2", (1920, 1080), 64)])
def test_direction_grid_equals_hierarchy_on_full_frames(tracer, workload, size, spp):
    """Full frames, tens of millions of closest-hit queries in every direction a bounce can take: the direction grid
    against the LBVH (itself pinned to the all-triangles loop above): ids, distances, radiance, ray counts. Includes the
    1M-triangle scene and the two reference heads at the sizes the benchmark configurations name (4K; 1080p)."""
    from pathtracer_amd import capi
    from pathtracer_amd.scenes import SceneData
    sc = SceneData(workload, *size)
    t = tracer
    prog = sc.upload(t)
    t.set_accum_mode(capi.ACCUM_MEAN)
    t.target_create(sc.width, sc.height)
    out = {}
    for name, mode in (("bvh", capi.ACCEL_BVH), ("grid", capi.ACCEL_GRID)):
        t.set_accel_mode(mode)
        t.target_clear()
        t.reset_stats()
        ids, ts = t.readback_primary()
        t.dispatch(1, spp, 1)
        t.finish()
        st = t.stats()
        out[name] = (ids, ts, t.readback(), st["rays"])
    assert out["grid"][3] == out["bvh"][3] and st["rays_traced"] > 1_000_000
    assert np.array_equal(out["grid"][0], out["bvh"][0])
    assert n_bit_mismatch(out["grid"][1], out["bvh"][1]) == 0
    assert n_bit_mismatch(out["grid"][2], out["bvh"][2]) == 0
    t.set_accel_mode(capi.ACCEL_AUTO)
    t.program_destroy(prog)


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("res", [0, 32, 1024])
def test_direction_grid_from_inside_a_cloud(seed, res):
    """Camera in the middle of a random cloud: triangles on all six cube-map faces, across face edges and corners, some
    huge, some almost touching the camera, some with an edge pointing straight at it, some seen at a grazing angle.
    The grid (coarse, default and fine resolutions) must return what the all-triangles loop returns: ids, distances,
    radiance with bounces in every direction."""
    import os
    from pathtracer_amd import Tracer, capi
    rng = np.random.default_rng(100 + seed)
    nt = 500
    centre = rng.normal(size=(nt, 1, 3))
    centre *= rng.uniform(0.05, 6.0, (nt, 1, 1)) / np.linalg.norm(centre, axis=2, keepdims=True)
    tri = centre + rng.uniform(-1, 1, (nt, 3, 3)) * rng.choice([0.02, 0.3, 1.5, 8.0], (nt, 1, 1))
    eye = np.array([0.1, -0.2, 0.05])
    # edges pointing at the camera: C = A + s (A - eye)
    for k in range(40):
        tri[k, 2] = tri[k, 0] + rng.uniform(0.2, 2.0) * (tri[k, 0] - eye)
    # grazing: third vertex almost in the plane through the camera and the first edge
    for k in range(40, 80):
        n = np.cross(tri[k, 0] - eye, tri[k, 1] - eye)
        n /= np.linalg.norm(n)
        mid = 0.5 * (tri[k, 0] + tri[k, 1])
        tri[k, 2] = mid + (mid - eye) * 0.3 + n * rng.choice([1e-3, -1e-3, 1e-2]) * np.linalg.norm(mid - eye)
    tri = tri.astype(np.float32)
    verts = np.concatenate([tri.reshape(-1, 3), np.zeros((nt * 3, 3), np.float32)], axis=1).astype(np.float32)
    idx = np.arange(nt * 3, dtype=np.uint32).reshape(-1, 3)
    W, H = 160, 120
    old = os.environ.get("PT_GRID_RES")
    if res:
        os.environ["PT_GRID_RES"] = str(res)
    try:
        t = Tracer(0)
    finally:
        if res:
            if old is None:
                del os.environ["PT_GRID_RES"]
            else:
                os.environ["PT_GRID_RES"] = old
    try:
        for target in [(1.0, 0.3, 0.2), (-0.2, -1.0, 0.4), (0.3, 0.2, -1.0)]:
            cam = po.camera(100.0, W, H, eye, eye + np.array(target), (0.0, 1.0, 0.1))
            res_ = both_modes(t, verts, idx, cam, W, H, 1, spp=4)
            assert_same(res_)
            assert (res_["brute"][0] >= 0).mean() > 0.5
    finally:
        t.close()


def test_direction_grid_with_lists_that_would_not_fit():
    """Thousands of triangles that each cover a large part of the sphere, on the finest grid: the lists would need billions
    of entries, so the build coarsens the grid (or gives it up and PT_ACCEL_AUTO falls back). Either way the image equals
    the all-triangles loop."""
    import os
    from pathtracer_amd import Tracer, capi
    rng = np.random.default_rng(77)
    nt = 3000
    centre = rng.normal(size=(nt, 1, 3))
    centre *= rng.uniform(0.5, 3.0, (nt, 1, 1)) / np.linalg.norm(centre, axis=2, keepdims=True)
    tri = (centre + rng.uniform(-30, 30, (nt, 3, 3))).astype(np.float32)
    verts = np.concatenate([tri.reshape(-1, 3), np.zeros((nt * 3, 3), np.float32)], axis=1).astype(np.float32)
    idx = np.arange(nt * 3, dtype=np.uint32).reshape(-1, 3)
    W, H = 96, 64
    cam = po.camera(90.0, W, H, (0, 0, 0), (0.2, 0.1, 1.0), (0, 1, 0))
    sph, _ = po.reference_spheres_and_materials()
    old = os.environ.get("PT_GRID_RES")
    os.environ["PT_GRID_RES"] = "2048"
    try:
        t = Tracer(0)
    finally:
        if old is None:
            del os.environ["PT_GRID_RES"]
        else:
            os.environ["PT_GRID_RES"] = old
    try:
        prog = t.program_create(capi.PROGRAM_MESH)
        t.use_program(prog)
        t.upload_spheres(sph)
        t.upload_mesh(verts, idx)
        t.set_normal_mode(1)
        t.target_create(W, H)
        t.set_uniforms(cam, nt, 10)
        t.build_accel()
        # the second camera position rebuilds the grid at the resolution the first one settled on
        for eye in ((0.0, 0.0, 0.0), (0.05, -0.1, 0.02)):
            cam = po.camera(90.0, W, H, eye, (0.2, 0.1, 1.0), (0, 1, 0))
            t.set_uniforms(cam, nt, 10)
            out = {}
            for name, mode in (("auto", capi.ACCEL_AUTO), ("brute", capi.ACCEL_BRUTE)):
                t.set_accel_mode(mode)
                t.target_clear()
                ids, ts = t.readback_primary()
                t.dispatch(1, 3, 1)
                t.finish()
                out[name] = (ids, ts, t.readback())
            st = t.stats()
            assert st["grid_entries"] < (1 << 30)
            assert np.array_equal(out["auto"][0], out["brute"][0])
            assert n_bit_mismatch(out["auto"][1], out["brute"][1]) == 0
            assert n_bit_mismatch(out["auto"][2], out["brute"][2]) == 0
            assert (out["brute"][0] >= 0).mean() > 0.9
    finally:
        t.close()


@pytest.mark.parametrize("scene_kind", ["small_mesh", "soup"])
def test_direction_grid_when_the_queue_of_large_jobs_overflows(scene_kind):
    """The grid build queues (triangle, face) jobs that cover many cells; a queue that cannot hold them means the resolution
    is too fine, and the build starts again at half of it, down to giving the grid up (PT_ACCEL_AUTO then uses another
    search). Forced here with a small queue. small_mesh: monkeyhead2 queues every job (about 1,200) at any resolution, so
    a queue of 16 never suffices and the grid is given up. soup: 20,000 triangles about a dozen cells across at the
    preferred resolution (all large jobs there) fit a queue of 2,000 only on a coarser grid. The image equals the
    all-triangles loop either way, also after a camera move (which rebuilds at the resolution settled on)."""
    import os
    from pathtracer_amd import Tracer, capi
    W, H = 128, 96
    if scene_kind == "small_mesh":
        fx = load_case("monkey_mesh_128")
        verts, idx, nm, queue = fx["vertices"], fx["indices"], int(fx["normal_mode"]), 16
        eyes = ((0.0, 0.0, 0.0), (0.1, 0.05, -0.2))
    else:
        rng = np.random.default_rng(5)
        nt = 20000
        centre = rng.normal(size=(nt, 1, 3))
        centre *= rng.uniform(1.0, 2.0, (nt, 1, 1)) / np.linalg.norm(centre, axis=2, keepdims=True)
        tri = (centre + rng.uniform(-0.04, 0.04, (nt, 3, 3))).astype(np.float32)
        verts = np.concatenate([tri.reshape(-1, 3), np.zeros((nt * 3, 3), np.float32)], axis=1).astype(np.float32)
        idx = np.arange(nt * 3, dtype=np.uint32).reshape(-1, 3)
        nm, queue = 1, 2000
        eyes = ((0.0, 0.0, 0.0), (0.02, -0.01, 0.03))
    sph, _ = po.reference_spheres_and_materials()
    old = os.environ.get("PT_GRID_QUEUE")
    os.environ["PT_GRID_QUEUE"] = str(queue)
    try:
        t = Tracer(0)
    finally:
        if old is None:
            del os.environ["PT_GRID_QUEUE"]
        else:
            os.environ["PT_GRID_QUEUE"] = old
    try:
        prog = t.program_create(capi.PROGRAM_MESH)
        t.use_program(prog)
        t.upload_spheres(sph)
        t.upload_mesh(verts, idx)
        t.set_normal_mode(nm)
        t.target_create(W, H)
        t.build_accel()
        nt = len(idx)
        entries = []
        for eye in eyes:
            cam = po.camera(60.0, W, H, eye, (eye[0], eye[1], eye[2] + 1.0), (0, 1, 0))
            t.set_uniforms(cam, nt, 10)
            out = {}
            for name, mode in (("auto", capi.ACCEL_AUTO), ("brute", capi.ACCEL_BRUTE)):
                t.set_accel_mode(mode)
                t.target_clear()
                ids, ts = t.readback_primary()
                t.dispatch(1, 3, 1)
                t.finish()
                out[name] = (ids, ts, t.readback())
                if name == "auto":
                    entries.append(t.stats()["grid_entries"])
            assert np.array_equal(out["auto"][0], out["brute"][0])
            assert n_bit_mismatch(out["auto"][1], out["brute"][1]) == 0
            assert n_bit_mismatch(out["auto"][2], out["brute"][2]) == 0
            assert (out["brute"][0] >= 0).mean() > 0.05
        if scene_kind == "small_mesh":
            assert entries == [0, 0]           # no grid: nothing fits a queue of 16
        else:
            assert all(e > nt for e in entries)  # a grid was built (on a coarser resolution than the preferred one)
    finally:
        t.close()


# ---- pt_verify_accel: the selected search structure against the shader's own all-triangles loop, on the device

def _verify(t, accel, bounces):
    t.set_accel_mode(accel)
    compared, bad_id, bad_t = t.verify_accel(bounces)
    t.set_accel_mode(0)
    return compared, bad_id, bad_t


@pytest.mark.parametrize("workload,size", [("cornell", (480, 270)), ("suzanne", (480, 270)), ("instanced", (960, 540))])
def test_block_bits_of_the_direction_grid_change_nothing(workload, size):
    """The grid's bit per 4x4 block of cells lets a query end before it fetches its cell word. Consulted always (PT_GRID_OCC=1),
    never (=0) or as the view's statistics decide (the default): the same image and the same ray count, and the self-check (which
    looks at the bits whenever they exist) finds every query equal to the loop over all triangles."""
    import os
    from pathtracer_amd import Tracer, capi
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(workload, *size)
    images, rays = {}, {}
    for mode in ("0", "1", None):
        old = os.environ.pop("PT_GRID_OCC", None)
        if mode is not None:
            os.environ["PT_GRID_OCC"] = mode
        try:
            t = Tracer(0)
        finally:
            os.environ.pop("PT_GRID_OCC", None)
            if old is not None:
                os.environ["PT_GRID_OCC"] = old
        with t:
            scene.upload(t)
            t.target_create(scene.width, scene.height)
            t.set_accel_mode(capi.ACCEL_GRID)
            for k in range(3):                       # the third dispatch runs after a list rebuild, i.e. after the flag was set
                t.dispatch(1 + 24 * k, 24, 1)
            t.finish()
            images[mode] = t.readback().copy()
            rays[mode] = t.stats()["rays"]
            compared, bad_id, bad_t = t.verify_accel(2)
            assert compared >= scene.width * scene.height and (bad_id, bad_t) == (0, 0)
    assert n_bit_mismatch(images["1"], images["0"]) == 0 and n_bit_mismatch(images[None], images["0"]) == 0
    assert rays["0"] == rays["1"] == rays[None]


def test_fine_direction_grid_for_a_view_that_stays():
    """A large mesh gets a finer direction grid (4096 cells per face edge instead of 2048) once a camera position has gathered
    enough pixel-samples to repay its longer build; a camera move goes back to the coarse one, and both sets of arrays are kept.
    With the threshold lowered (PT_GRID_FINE_AFTER) so that a small frame crosses it: the grid's entry count shows which one is
    in use, the images and ray counts equal those of a context held to the coarse grid (PT_GRID_FINE_AFTER=-1) bit for bit at
    every step, and the self-check finds the fine grid's answers equal to the loop over all triangles."""
    import os
    from pathtracer_amd import Tracer, capi, host_api
    from pathtracer_amd.scenes import SceneData
    scene = SceneData("instanced", 640, 360)
    cam2 = host_api.camera(60.0, scene.width, scene.height, (150.0, -90.0, -1300.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))

    def run(fine_after):
        old = os.environ.pop("PT_GRID_FINE_AFTER", None)
        os.environ["PT_GRID_FINE_AFTER"] = fine_after
        try:
            t = Tracer(0)
        finally:
            os.environ.pop("PT_GRID_FINE_AFTER", None)
            if old is not None:
                os.environ["PT_GRID_FINE_AFTER"] = old
        out = []
        with t:
            scene.upload(t)
            t.target_create(scene.width, scene.height)
            t.set_accel_mode(capi.ACCEL_GRID)

            def step(first, n, verify=False):
                t.dispatch(first, n, 1)
                t.finish()
                st = t.stats()
                if verify:
                    compared, bad_id, bad_t = t.verify_accel(1)
                    assert compared >= scene.width * scene.height and (bad_id, bad_t) == (0, 0)
                out.append((t.readback().copy(), st["rays"], st["grid_entries"]))

            step(1, 4)                     # 0.9e6 pixel-samples: coarse
            step(5, 4, verify=True)        # 1.8e6 with this dispatch: the fine grid is built for it
            step(9, 4)                     # stays fine
            t.set_uniforms(cam2, scene.num_triangles, 10)
            t.target_clear()
            step(1, 2, verify=True)        # new camera position: coarse again (arrays at hand)
            step(3, 16)                    # long enough by itself: fine
            t.set_uniforms(scene.camera, scene.num_triangles, 10)
            t.target_clear()
            step(1, 16, verify=True)       # a move brought by a dispatch that is long enough: fine at once
        return out

    switching, coarse = run("1.5e6"), run("-1")
    for k, ((img, rays, _), (img0, rays0, _)) in enumerate(zip(switching, coarse)):
        assert n_bit_mismatch(img, img0) == 0 and rays == rays0, f"step {k}"
    e = [entries for _, _, entries in switching]
    e0 = [entries for _, _, entries in coarse]
    assert e0[0] == e0[1] == e0[2] and e0[3] == e0[4]       # the coarse grid of each camera position
    assert e[0] == e0[0] and e[1] > e[0] and e[2] == e[1]   # coarse, then fine (more cells: more entries)
    assert e[3] == e0[3] and e[4] > e[3]
    assert e[5] == e[1]                                     # back at the first position, fine from the start


@pytest.mark.parametrize("workload,size,spp", [("monkeyhead2", (480, 270), 6), ("suzanne", (640, 360), 5), ("instanced", (960, 540), 3),
                                               ("torus_distorted", (480, 270), 6)])
def test_materials_mode_camera_rays_through_the_direction_grid(workload, size, spp):
    """With the search left to the library (PT_ACCEL_AUTO) the materials mode takes the camera rays' hits from the direction grid,
    from the second dispatch of a camera position on, and walks the hierarchy for the bounce rays only. Against the hierarchy for
    every ray (PT_ACCEL_BVH, and PT_CAMERA_RAY_GRID=0): the same image and ray count bit for bit, in one dispatch, in single-sample
    frames, in a window, and after a camera move; the camera rays' ids and distances are the same too."""
    import os
    from pathtracer_amd import Tracer, capi
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(workload, *size)
    W, H = scene.width, scene.height

    def run(t, accel):
        out = {}
        scene.upload(t, shading="materials")
        t.target_create(W, H)
        t.set_accel_mode(accel)
        t.reset_stats()
        t.dispatch(1, spp, 1)
        for k in range(3):                                     # frames of one sample each
            t.dispatch(spp + 1 + k, 1, 1)
        t.finish()
        out["full"] = t.readback().copy()
        out["rays"] = t.stats()["rays"]
        out["ids"], out["t"] = t.readback_primary()
        t.set_window(W // 5, H // 4, W // 2, H // 3)
        t.target_clear()
        t.dispatch(1, spp, 1)
        t.finish()
        out["window"] = t.readback().copy()
        t.set_window(0, 0, W, H)
        cam = scene.camera.copy()                               # the camera a little to the side: records and grid are redone
        shift = np.float32(0.03) * np.abs(cam[6:9]).max() * np.array([1.0, 0.5, 0.0], np.float32)
        cam[0:3] += shift
        cam[3:6] += shift
        t.set_uniforms(cam, scene.num_triangles, 10)
        t.target_clear()
        t.dispatch(1, 2, 1)                                      # (first dispatch from the new position: no grid yet)
        t.dispatch(3, spp, 1)
        t.finish()
        out["moved"] = t.readback().copy()
        out["grid_entries"] = t.stats()["grid_entries"]
        # (without the grid this is the loop over the compacted records, which no materials dispatch has brought up to date
        # for the new position: the call has to do that itself)
        out["ids_moved"], out["t_moved"] = t.readback_primary()
        return out

    results = {}
    for label, accel, env in (("grid", capi.ACCEL_AUTO, None), ("walk", capi.ACCEL_BVH, None), ("auto_walk", capi.ACCEL_AUTO, "0")):
        old = os.environ.pop("PT_CAMERA_RAY_GRID", None)
        if env is not None:
            os.environ["PT_CAMERA_RAY_GRID"] = env
        try:
            t = Tracer(0)
        finally:
            os.environ.pop("PT_CAMERA_RAY_GRID", None)
            if old is not None:
                os.environ["PT_CAMERA_RAY_GRID"] = old
        with t:
            results[label] = run(t, accel)
    g = results["grid"]
    for other in ("walk", "auto_walk"):
        o = results[other]
        assert n_bit_mismatch(g["full"], o["full"]) == 0 and n_bit_mismatch(g["window"], o["window"]) == 0
        assert n_bit_mismatch(g["moved"], o["moved"]) == 0 and n_bit_mismatch(g["moved"], g["full"]) > 0
        assert g["rays"] == o["rays"]
        assert np.array_equal(g["ids"], o["ids"]) and np.array_equal(g["t"].view(np.uint32), o["t"].view(np.uint32))
        assert np.array_equal(g["ids_moved"], o["ids_moved"]) and np.array_equal(g["t_moved"].view(np.uint32), o["t_moved"].view(np.uint32))
    assert not np.array_equal(g["ids_moved"], g["ids"])
    assert g["rays"] > W * H * (spp + 3)
    assert g["grid_entries"] > 0 and results["walk"]["grid_entries"] == 0 and results["auto_walk"]["grid_entries"] == 0


@pytest.mark.parametrize("workload,size", [("cornell", (640, 360)), ("monkeyhead2", (640, 360)), ("suzanne", (480, 270)),
                                           ("torus_distorted", (960, 540))])
def test_self_check_of_every_structure(workload, size):
    """Camera rays and three generations of bounce rays: direction grid, LBVH and the compacted loop give the triangle and
    the distance the loop over all triangles gives, for every query."""
    from pathtracer_amd import PtError, Tracer
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(workload, *size)
    with Tracer(0) as t:
        scene.upload(t, accel=False)
        t.target_create(scene.width, scene.height)
        compared, bad_id, bad_t = _verify(t, 0, 3)          # no pt_build_accel yet: the loop over the compacted records
        assert compared > scene.width * scene.height and (bad_id, bad_t) == (0, 0)
        t.build_accel()
        results = {accel: _verify(t, accel, 3) for accel in (2, 4)}
        for accel, (c, bi, bt) in results.items():
            assert c == compared and (bi, bt) == (0, 0), (accel, c, bi, bt)
        assert _verify(t, 1, 1)[1:] == (0, 0)                # PT_ACCEL_BRUTE still walks the compacted records
        t.set_uniforms(scene.camera, scene.num_triangles - 1, 10)   # a prefix of the mesh: the plain loop, nothing to check
        with pytest.raises(PtError) as err:
            t.verify_accel(1)
        assert err.value.status == 3
        t.set_uniforms(scene.camera, scene.num_triangles, 10)
        # rays with their own origins (diffuse/emissive mode): per-lane and per-wave hierarchy walk
        if workload != "cornell":                            # (36 triangles: AUTO keeps the loop there)
            t.upload_materials(scene.material_table)
            t.upload_material_ids(scene.material_ids)
            t.set_shading_mode(1)
            c, bi, bt = _verify(t, 2, 3)
            assert c > scene.width * scene.height and (bi, bt) == (0, 0)


def test_self_check_on_the_full_4k_million_triangle_frame():
    """The frame of the benchmark's fifth configuration: every camera ray of the 3840x2160 view of the 999,744-triangle scene and the first
    generation of bounce rays, through the direction grid and through the LBVH, against the loop over all triangles."""
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    scene = SceneData("instanced")
    with Tracer(0) as t:
        scene.upload(t)
        t.target_create(scene.width, scene.height)
        c4, bi, bt = _verify(t, 4, 1)
        assert c4 > scene.width * scene.height and (bi, bt) == (0, 0)
        c2, bi, bt = _verify(t, 2, 0)
        assert c2 == scene.width * scene.height and (bi, bt) == (0, 0)


def _tracer_with_env(**env):
    """A context created while the given environment variables are set (pt_create reads its knobs once)."""
    import os
    from pathtracer_amd import Tracer
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return Tracer(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def test_self_check_follows_a_moving_camera():
    """A moved camera gets rebuilt records, boxes and grid (stale ones would show up as mismatches): beside, just in front of, and
    next to the mesh, through every structure."""
    import pt_oracle as po
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    scene = SceneData("monkeyhead2", 320, 180)
    with Tracer(0) as t:
        scene.upload(t)
        t.target_create(320, 180)
        assert _verify(t, 4, 2)[1:] == (0, 0)
        for eye in ((0.4, 0.2, 0.5), (0.0, 0.0, 2.2), (-1.5, 0.3, 3.0)):
            t.set_uniforms(po.camera(60.0, 320, 180, eye, (0, 0, 3), (0, 1, 0)), scene.num_triangles, 10)
            for accel in (0, 2, 4):
                c, bi, bt = _verify(t, accel, 2)
                assert c >= 320 * 180 and (bi, bt) == (0, 0), (eye, accel, c, bi, bt)


@pytest.mark.parametrize("workload,size", [("monkeyhead2", (320, 180)), ("suzanne", (320, 180)), ("instanced", (960, 540))])
def test_self_check_reports_a_broken_structure(workload, size):
    """The check is not vacuous. PT_FAULT_INJECT damages a structure after it is built -- `grid`: a sixteenth of the rows of cells, through the
    middle of the +z face, forget their triangles; `bvh`: the root of every form of the hierarchy forgets its first child --
    and pt_verify_accel must then count queries whose triangle AND distance differ from the loop over all triangles, through the
    grid, the per-wave walk and the per-lane walk, while the undamaged structures of the same context still count none. The
    damaged structure also renders a different image than the sound one (the check sees what a dispatch would use)."""
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(workload, *size)
    px = scene.width * scene.height

    def images(t, accel, shading):
        scene.upload(t, shading=shading)
        t.target_create(scene.width, scene.height)
        t.set_accel_mode(accel)
        t.dispatch(1, 2, 1)
        t.finish()
        return t.readback().copy()

    with Tracer(0) as t:
        sound = {(4, "reference"): images(t, 4, "reference"), (2, "reference"): images(t, 2, "reference"),
                 (2, "materials"): images(t, 2, "materials")}
    with _tracer_with_env(PT_FAULT_INJECT="grid") as t:
        broken = images(t, 4, "reference")
        c, bi, bt = t.verify_accel(1)
        assert c >= px and 0 < bt <= bi, (c, bi, bt)             # a lost triangle changes the hit, hence id and distance
        assert bi < c // 2                                        # ... in the damaged rows only
        assert n_bit_mismatch(broken, sound[(4, "reference")]) > 0
        assert _verify(t, 2, 1)[1:] == (0, 0)                     # the hierarchy of the same context is sound
    with _tracer_with_env(PT_FAULT_INJECT="bvh") as t:
        broken = images(t, 2, "reference")
        c, bi, bt = t.verify_accel(1)                             # per-wave walk over the binary form
        assert c >= px and 0 < bt <= bi, (c, bi, bt)
        assert n_bit_mismatch(broken, sound[(2, "reference")]) > 0
        assert _verify(t, 4, 1)[1:] == (0, 0)                     # the direction grid of the same context is sound
        broken = images(t, 2, "materials")
        c, bi, bt = t.verify_accel(1)                             # per-lane walk over the four-child form
        assert c >= px and 0 < bt <= bi, (c, bi, bt)
        assert n_bit_mismatch(broken, sound[(2, "materials")]) > 0
    with _tracer_with_env(PT_FAULT_INJECT="bvh", PT_LANE_WALK="0") as t:
        images(t, 2, "materials")
        c, bi, bt = t.verify_accel(1)                             # per-wave walk over the absolute binary form
        assert c >= px and bi > 0, (c, bi, bt)


def test_hierarchy_over_degenerate_layouts_stays_shallow():
    """Meshes on which a cost-driven split peels off a sliver per level: thousands of zero-area triangles strung along a line
    (every candidate plane costs 0, so the first one wins: a sixteenth of the range per level), and centroids in geometric
    progression. The host builder halves ranges beyond a fixed depth, the device builder gives its cluster merging up when a round
    merges next to nothing and builds the Morton-code hierarchy instead: either way the height stays below the walks' fixed stacks
    (PtStats::bvh_sweeps bounds it), and a camera at the far end of the string -- for which the deep child is the near one, the
    worst case for the stack -- gets the same hits as the loop over all triangles, in both shading modes."""
    from pathtracer_amd import Tracer, capi
    n_line = 6000
    x = np.linspace(0.0, 60.0, n_line, dtype=np.float32)
    line = np.stack([np.stack([x, 0 * x, 0 * x], 1), np.stack([x + 0.004, 0 * x, 0 * x], 1), np.stack([x + 0.008, 0 * x, 0 * x], 1)], 1)
    k = np.arange(110, dtype=np.float64)
    gx = (1.0e-4 * 1.18 ** k).astype(np.float32)                       # 1e-4 .. 8e3 in geometric progression
    geo = np.stack([np.stack([gx, 0 * gx, 0 * gx + 5], 1), np.stack([gx * 1.1, 0 * gx + 1, 0 * gx + 5], 1), np.stack([gx * 1.2, 0 * gx, 0 * gx + 6], 1)], 1)
    rng = np.random.default_rng(5)
    real = (rng.uniform(-1.5, 1.5, (300, 1, 3)) + np.array([30.0, 0.0, 0.0]) + rng.uniform(-0.8, 0.8, (300, 3, 3))).astype(np.float32)
    wall = np.array([[[x0, -25, -25], [x0, -25, 25], [x0, 25, 0]] for x0 in (-2.0, 63.0)] +
                    [[[x0, -25, -25], [x0, 25, 0], [x0, -25, 25]] for x0 in (-2.0, 63.0)], np.float32)   # both windings, both ends
    tris = np.concatenate([line, geo.astype(np.float32), real, wall]).astype(np.float32)
    verts = np.concatenate([tris.reshape(-1, 3), np.zeros((len(tris) * 3, 3), np.float32)], 1)
    idx = np.arange(len(verts), dtype=np.uint32).reshape(-1, 3)
    mats = np.zeros((10, 4), np.float32)
    mats[:, :3] = 0.7
    mats[1, 3] = 2.0
    mids = (np.arange(len(idx)) % 2).astype(np.int32)
    W, H = 96, 64
    sph, _ = po.reference_spheres_and_materials()
    with Tracer(0) as t:
        for builder, eye, target, shading in ((capi.BVH_BUILDER_SAH, (62.0, 0.02, 0.01), (0.0, 0.0, 0.0), 0), (capi.BVH_BUILDER_SAH, (62.0, 0.02, 0.01), (0.0, 0.0, 0.0), 1),
                                              (capi.BVH_BUILDER_SAH, (-1.0, 0.01, 0.02), (60.0, 0.0, 0.0), 1), (capi.BVH_BUILDER_PLOC, (62.0, 0.02, 0.01), (0.0, 0.0, 0.0), 0),
                                              (capi.BVH_BUILDER_PLOC, (-1.0, 0.01, 0.02), (60.0, 0.0, 0.0), 1), (capi.BVH_BUILDER_BINNED, (62.0, 0.02, 0.01), (0.0, 0.0, 0.0), 0),
                                              (capi.BVH_BUILDER_BINNED, (62.0, 0.02, 0.01), (0.0, 0.0, 0.0), 1), (capi.BVH_BUILDER_BINNED, (-1.0, 0.01, 0.02), (60.0, 0.0, 0.0), 1)):
            cam = po.camera(50.0, W, H, eye, target, (0.0, 1.0, 0.0))
            prog = t.program_create(capi.PROGRAM_MESH)
            t.use_program(prog)
            t.upload_spheres(sph)
            t.upload_materials(mats)
            t.upload_mesh(verts, idx)
            t.upload_material_ids(mids)
            t.set_normal_mode(1)
            t.set_shading_mode(shading)
            t.target_create(W, H)
            t.set_uniforms(cam, len(idx), 10)
            t.set_bvh_builder(builder)
            t.build_accel()
            t.set_bvh_builder(capi.BVH_BUILDER_BINNED)
            sweeps = t.stats()["bvh_sweeps"]
            # surface-area builders (host and device): 40 cost-driven levels, then halving (13 more for 6,400 triangles); the device's
            # cluster merging makes no headway on the string of zero-area boxes and hands over to the Morton-code hierarchy (at most
            # 95 levels, usually ~30)
            assert 8 <= sweeps <= (96 if builder == capi.BVH_BUILDER_PLOC else 64), (builder, sweeps)
            t.set_accel_mode(capi.ACCEL_BVH)
            c, bi, bt = t.verify_accel(2)
            assert c >= W * H and (bi, bt) == (0, 0), (builder, eye, shading, c, bi, bt)
            t.dispatch(1, 3, 1)
            t.finish()
            walked = t.readback().copy()
            t.set_accel_mode(capi.ACCEL_BRUTE)
            t.target_clear()
            t.dispatch(1, 3, 1)
            t.finish()
            assert n_bit_mismatch(walked, t.readback()) == 0
            assert (t.readback_primary()[0] >= 0).sum() > 1000
            t.set_accel_mode(capi.ACCEL_AUTO)
            t.program_destroy(prog)


@pytest.mark.parametrize("shading", ["reference", "materials"])
def test_million_triangle_crops_against_the_oracle(shading):
    """The 999,744-triangle frame of the benchmark's fifth configuration meets the CPU restatement of the shader directly (not
    only the GPU's own loop): 32x32 crops of the 3840x2160 view on silhouettes of the instances, camera-ray triangles and
    distances, radiance after 1 and 2 passes and the ray count, bit for bit, in the reference shading (direction grid) and in the
    diffuse/emissive mode (hierarchy walk per lane); the oracle tests all triangles for every ray (seconds per crop)."""
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    sc = SceneData("instanced")
    mode = 1 if shading == "materials" else 0
    scene = po.Scene(sc.vertices, sc.indices, sc.spheres, sc.material_table if mode else sc.materials, sc.material_ids if mode else None)
    u = po.uniforms(sc.camera, sc.width, sc.height, sc.num_triangles, 10, po.VARIANT_MESH, sc.normal_mode, mode)
    with Tracer(0) as t:
        sc.upload(t, shading=shading)
        t.target_create(sc.width, sc.height)
        for window in [(1932, 1092, 32, 32), (1251, 528, 32, 32), (2440, 1480, 32, 32)]:
            x0, y0, w, h = window
            win = (slice(y0, y0 + h), slice(x0, x0 + w))
            t.set_window(*window)
            t.target_clear()
            t.reset_stats()
            ids, ts = t.readback_primary()
            oids, ots, _ = po.primary(scene, u, window=window, threads=16)
            assert np.array_equal(ids[win], oids[win])
            hit = oids[win] >= 0
            assert 50 < hit.sum() < w * h, hit.sum()        # a silhouette: hits and sky
            assert n_bit_mismatch(ts[win][hit], ots[win][hit]) == 0
            rgba, rays = None, 0
            for first, count in ((1, 1), (2, 1)):
                rgba, r = po.render(scene, u, first, count, rgba, window=window, threads=16)
                rays += r
                t.dispatch(first, count, 1)
                t.finish()
                assert n_bit_mismatch(t.readback()[win], rgba[win]) == 0, (window, first)
            assert t.stats()["rays"] == rays




@pytest.mark.parametrize("workload,spp", [("monkeyhead2", 4), ("suzanne", 4), ("torus_distorted", 2)])
def test_device_surface_area_builder_makes_the_host_builders_tree(workload, spp):
    """PT_BVH_BUILDER_BINNED applies SahBuilder::split's rules on the device. Below 131,072 triangles nothing is sampled, so the two
    builders must choose the same planes and the walks must fetch exactly the same numbers of nodes and triangle records (the
    numbering of the wide nodes differs, the tree does not); images equal as for every builder."""
    from pathtracer_amd import Tracer, capi
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(workload)
    seen = {}
    for builder in (capi.BVH_BUILDER_SAH, capi.BVH_BUILDER_BINNED):
        with Tracer(0) as t:
            t.set_bvh_builder(builder)
            scene.upload(t, shading="materials")
            t.target_create(480, 270)
            t.set_accel_mode(capi.ACCEL_BVH)
            t.dispatch(1, spp, 1)
            t.finish()
            st = t.stats()
            seen[builder] = (st["grid_triangle_tests"], st["lane_walk_records"], st["rays_traced"], st["bvh_sweeps"], t.readback().copy())
    a, b = seen[capi.BVH_BUILDER_SAH], seen[capi.BVH_BUILDER_BINNED]
    assert a[:4] == b[:4] and a[0] > 0, (a[:4], b[:4])
    assert n_bit_mismatch(a[4], b[4]) == 0
