"""pt_dispatch_static (the dispatch that allocates nothing: its scheduling scratch is a fixed array in the library's device
image) against pt_dispatch, through the C ABI, bit for bit: two calls in a row of which the second is smaller (smaller
target, fewer samples), then windows, short ranges, the materials mode, the size limit and the state errors. The array
keeps the previous call's contents, so the order of the calls matters here: large before small."""
import ctypes

import numpy as np
import pytest

from helpers import load_case, n_bit_mismatch

pytestmark = pytest.mark.gpu


def _free_device_memory():
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def _setup(t, fx, W, H, shading=0):
    prog = t.program_create(1)
    t.use_program(prog)
    t.upload_spheres(fx["spheres"])
    t.upload_materials(fx["materials"])
    t.upload_mesh(fx["vertices"], fx["indices"])
    if shading:
        t.upload_material_ids(fx["material_ids"])
    t.set_normal_mode(int(fx["normal_mode"]))
    t.set_shading_mode(shading)
    t.set_accel_mode(0)
    t.set_accum_mode(0)
    t.target_create(W, H)
    return prog


def _both(t, first, n, stride=1):
    """The same range through pt_dispatch and through pt_dispatch_static, each from a cleared target."""
    t.target_clear()
    t.dispatch(first, n, stride)
    t.finish()
    want = t.readback()
    t.reset_stats()
    t.target_clear()
    t.dispatch_static(first, n, stride)
    t.finish()
    got = t.readback()
    return want, got


def test_static_dispatch_equals_dispatch_large_then_small():
    import pt_oracle as po
    from pathtracer_amd import PtError, Tracer
    fx = load_case("monkey_mesh_128")
    ntri = len(fx["indices"])
    with Tracer(0) as t:
        prog = _setup(t, fx, 640, 360)
        t.build_accel()
        cam = po.camera(60.0, 640, 360, (0, 0, 0), (0, 0, 1), (0, 1, 0))
        t.set_uniforms(cam, ntri, 10)
        # the direction grid of this view does not exist yet and building it allocates: refused, nothing rendered
        with pytest.raises(PtError) as err:
            t.dispatch_static(1, 4, 1)
        assert err.value.status == 3 and "pt_dispatch" in str(err.value)
        # call 1: 96 samples = an 8-sample probe, the scheduler, 88 samples from the work list with split tiles
        want, got = _both(t, 1, 96)
        assert n_bit_mismatch(got, want) == 0
        st = t.stats()
        assert st["samples"] == 640 * 360 * 96 and st["dispatches"] == 1
        # call 2, smaller: a 200x120 target and 40 samples; the array still holds call 1's costs, list and samples
        t.target_create(200, 120)
        t.set_uniforms(po.camera(60.0, 200, 120, (0, 0, 0), (0, 0, 1), (0, 1, 0)), ntri, 10)
        want, got = _both(t, 1, 40)
        assert n_bit_mismatch(got, want) == 0
        # ... and against the CPU oracle
        scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
        u = po.uniforms(po.camera(60.0, 200, 120, (0, 0, 0), (0, 0, 1), (0, 1, 0)), 200, 120, ntri, 10, 1, int(fx["normal_mode"]))
        ref, rays = po.render(scene, u, 1, 40, threads=8)
        assert n_bit_mismatch(got, ref) == 0 and t.stats()["rays"] == rays
        # a window that cuts tiles, a strided range, a range too short for a work list, continuing an image
        t.set_window(17, 33, 90, 41)
        want, got = _both(t, 3, 33, 2)
        assert n_bit_mismatch(got, want) == 0
        t.set_window()
        want, got = _both(t, 1, 8)
        assert n_bit_mismatch(got, want) == 0
        t.dispatch(9, 50, 1)          # samples 9..58 after the 8 just rendered, both ways
        t.finish()
        want = t.readback()
        t.target_clear()
        t.dispatch_static(1, 8, 1)
        t.dispatch_static(9, 50, 1)
        t.finish()
        assert n_bit_mismatch(t.readback(), want) == 0
        # every search mode
        for accel in (1, 2, 4):
            t.set_accel_mode(accel)
            want, got = _both(t, 1, 36)
            assert n_bit_mismatch(got, want) == 0
        t.set_accel_mode(0)
        # row-interleave partition (image-tile split across renderers)
        t.set_row_interleave(16, 1, 3)
        want, got = _both(t, 1, 34)
        assert n_bit_mismatch(got, want) == 0
        t.set_row_interleave(16, 0, 1)
        t.finish()
        # free device memory does not move over a run of static dispatches (long, short, windowed)
        before = _free_device_memory()
        for n, window in ((96, ()), (40, (17, 33, 90, 41)), (8, ()), (64, ())):
            t.set_window(*window)
            t.target_clear()
            t.dispatch_static(1, n, 1)
        t.finish()
        assert _free_device_memory() == before
        t.program_destroy(prog)


def test_static_dispatch_materials_mode_and_sphere_program():
    from pathtracer_amd import Tracer
    fx = load_case("cornell_materials_128")
    with Tracer(0) as t:
        prog = _setup(t, fx, 128, 128, shading=1)
        t.set_uniforms(fx["camera"], len(fx["indices"]), 10)
        want, got = _both(t, 1, 64)      # parked samples take 16 bytes in this mode
        assert n_bit_mismatch(got, want) == 0
        t.build_accel()
        t.set_accel_mode(2)
        want, got = _both(t, 1, 48)
        assert n_bit_mismatch(got, want) == 0
        t.program_destroy(prog)
    fx = load_case("world_128")
    with Tracer(0) as t:
        prog = t.program_create(0)
        t.use_program(prog)
        t.upload_spheres(fx["spheres"])
        t.upload_materials(fx["materials"])
        t.target_create(320, 200)
        import pt_oracle as po
        t.set_uniforms(po.camera(60.0, 320, 200, (0, 0, 0), (0, 0, 1), (0, 1, 0)), 0, 10)
        want, got = _both(t, 1, 80)
        assert n_bit_mismatch(got, want) == 0
        t.set_accum_mode(1)              # running sums (what sample-split ranks produce)
        want, got = _both(t, 2, 40, 3)
        assert n_bit_mismatch(got, want) == 0
        t.program_destroy(prog)


def test_static_dispatch_rejects_windows_beyond_its_array():
    import pt_oracle as po
    from pathtracer_amd import PtError, Tracer
    fx = load_case("world_128")
    with Tracer(0) as t:
        prog = t.program_create(0)
        t.use_program(prog)
        t.upload_spheres(fx["spheres"])
        t.upload_materials(fx["materials"])
        W, H = 4096, 2200                 # 512 x 275 = 140,800 tiles > 131,072
        t.target_create(W, H)
        t.set_uniforms(po.camera(60.0, W, H, (0, 0, 0), (0, 0, 1), (0, 1, 0)), 0, 10)
        with pytest.raises(PtError) as err:
            t.dispatch_static(1, 2, 1)
        assert err.value.status == 5 and "131072" in str(err.value)
        with pytest.raises(PtError):
            t.dispatch_static(0, 1, 1)
        # a window of it that fits is served, and equals pt_dispatch
        t.set_window(100, 50, 3840, 2100)
        want, got = _both(t, 1, 2)
        assert n_bit_mismatch(got, want) == 0
        t.program_destroy(prog)


def test_static_dispatch_refuses_to_share_its_array_with_a_running_call():
    """All contexts of a GPU share the scratch array: while one context's pt_dispatch_static is still running, another
    context's call is refused with PT_ERR_STATE and renders nothing; once the first has finished the second goes through and
    both images equal pt_dispatch's."""
    import pt_oracle as po
    from pathtracer_amd import PtError, Tracer
    fx = load_case("cornell_geo_128")
    ntri = len(fx["indices"])
    W, H = 1920, 1080
    cam = po.camera(39.3, W, H, (278, 273, -800), (278, 273, 0), (0, 1, 0))
    with Tracer(0) as a, Tracer(0) as b:
        for t in (a, b):
            _setup(t, fx, W, H)
            t.set_uniforms(cam, ntri, 10)
        a.dispatch(1, 512, 1)
        a.finish()
        want = a.readback().copy()
        a.target_clear()
        a.finish()
        a.dispatch_static(1, 512, 1)              # tens of milliseconds of device time, issued asynchronously
        with pytest.raises(PtError) as err:
            b.dispatch_static(1, 8, 1)
        assert err.value.status == 3 and "another context" in str(err.value)
        assert b.stats()["dispatches"] == 0 and np.all(b.readback()[..., :3] == 0)
        a.finish()
        assert n_bit_mismatch(a.readback(), want) == 0
        b.dispatch_static(1, 512, 1)              # the array is free again
        b.finish()
        assert n_bit_mismatch(b.readback(), want) == 0
        a.dispatch_static(513, 8, 1)              # ... and back
        a.finish()
