"""ctypes binding of the C ABI declared in include/pt_hip.h (library: pathtracer_amd/lib/libpt_hip.so).

There is no fallback: if the library is missing, or a call fails, an exception is raised. The library is built
in-tree by pathtracer_amd/build.py (hipcc, gfx950).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpt_hip.so")

PT_OK = 0
PT_ERR_INVALID_ARGUMENT, PT_ERR_HIP, PT_ERR_STATE, PT_ERR_NO_DEVICE, PT_ERR_UNSUPPORTED = 1, 2, 3, 4, 5
PROGRAM_WORLD, PROGRAM_MESH, PROGRAM_CPU_TRACER = 0, 1, 2
NORMALS_VERTEX, NORMALS_GEOMETRIC = 0, 1
ACCUM_MEAN, ACCUM_SUM = 0, 1
ACCEL_AUTO, ACCEL_BRUTE, ACCEL_BVH, ACCEL_CULL, ACCEL_GRID = 0, 1, 2, 3, 4
SHADE_REFERENCE, SHADE_MATERIALS, SHADE_MATERIALS_UNIT = 0, 1, 2
BVH_BUILDER_SAH, BVH_BUILDER_MORTON, BVH_BUILDER_PLOC, BVH_BUILDER_BINNED = 0, 1, 2, 3


class PtUniforms(C.Structure):
    _fields_ = [("camera_position", C.c_float * 3), ("camera_lower_left_corner", C.c_float * 3),
                ("camera_horizontal", C.c_float * 3), ("camera_vertical", C.c_float * 3),
                ("num_triangles", C.c_int32), ("depth", C.c_int32)]


class PtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("rays_traced", C.c_uint64), ("samples", C.c_uint64), ("lane_slots", C.c_uint64), ("bvh_node_fetches", C.c_uint64),
                ("bvh_triangle_fetches", C.c_uint64), ("grid_triangle_tests", C.c_uint64), ("grid_entries", C.c_uint64),
                ("shaded_hits", C.c_uint64), ("loop_triangle_visits", C.c_uint64), ("loop_full_tests", C.c_uint64), ("lane_walk_records", C.c_uint64),
                ("dispatches", C.c_uint32),
                ("last_dispatch_ms", C.c_float), ("total_dispatch_ms", C.c_double), ("bvh_sweeps", C.c_uint32), ("grid_lists_carried", C.c_uint32),
                ("gathered_pixels", C.c_uint32), ("gathered_tiles", C.c_uint32)]


class PtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"pt_hip status {status}: {message}")
        self.status = status


# name -> (restype, argtypes); every symbol include/pt_hip.h declares
SIGNATURES = {
    "pt_abi_version": (C.c_int32, []),
    "pt_last_error": (C.c_char_p, []),
    "pt_device_count": (C.c_int, [C.POINTER(C.c_int32)]),
    "pt_create": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p)]),
    "pt_destroy": (C.c_int, [C.c_void_p]),
    "pt_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pt_program_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]),
    "pt_program_destroy": (C.c_int, [C.c_void_p, C.c_uint32]),
    "pt_use_program": (C.c_int, [C.c_void_p, C.c_uint32]),
    "pt_upload_spheres": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "pt_upload_materials": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "pt_upload_mesh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    "pt_set_normal_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_upload_material_ids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "pt_set_shading_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_set_primary_reuse": (C.c_int, [C.c_void_p, C.c_int32]),
    "pt_build_accel": (C.c_int, [C.c_void_p]),
    "pt_set_bvh_builder": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_set_accel_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_target_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "pt_target_bind_external": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "pt_target_clear": (C.c_int, [C.c_void_p]),
    "pt_set_window": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "pt_set_row_interleave": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "pt_set_accum_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_set_cpu_tracer": (C.c_int, [C.c_void_p, C.c_uint32, C.c_float]),
    "pt_set_uniforms": (C.c_int, [C.c_void_p, C.POINTER(PtUniforms)]),
    "pt_dispatch": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "pt_dispatch_static": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "pt_finish": (C.c_int, [C.c_void_p]),
    "pt_readback_rgba32f": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pt_upload_rgba32f": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pt_present_rgba8": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pt_current_image": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "pt_readback_primary": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pt_enable_timing": (C.c_int, [C.c_void_p, C.c_int32]),
    "pt_get_stats": (C.c_int, [C.c_void_p, C.POINTER(PtStats)]),
    "pt_reset_stats": (C.c_int, [C.c_void_p]),
    "pt_verify_accel": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)]),
    "pt_selftest_arithmetic": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
}

_libs = {}


def load(path=None):
    """Loads libpt_hip.so (or an experimental build at `path`) and binds every entry point. Raises if the library
    has not been built."""
    path = path or LIB_PATH
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m pathtracer_amd.build` (needs hipcc). "
                               "There is no CPU fallback.")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _libs[path] = lib
    return _libs[path]


def check(status, lib=None):
    if status != PT_OK:
        raise PtError(status, (lib or load()).pt_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int32(0)
    check(load().pt_device_count(C.byref(n)))
    return n.value


class Tracer:
    """One context (device + stream) with the calls of include/pt_hip.h as methods. Thin: argument packing only."""

    def __init__(self, device=0, lib_path=None):
        self._lib = load(lib_path)
        self._ctx = C.c_void_p()
        check(self._lib.pt_create(device, C.byref(self._ctx)), self._lib)
        self.width = self.height = 0

    def close(self):
        if self._ctx:
            self._lib.pt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- program
    def program_create(self, kind):
        h = C.c_uint32(0)
        check(self._lib.pt_program_create(self._ctx, kind, C.byref(h)), self._lib)
        return h.value

    def program_destroy(self, program):
        check(self._lib.pt_program_destroy(self._ctx, program), self._lib)

    def use_program(self, program):
        check(self._lib.pt_use_program(self._ctx, program), self._lib)

    def set_stream(self, hip_stream):
        check(self._lib.pt_set_stream(self._ctx, C.c_void_p(hip_stream)), self._lib)

    # -- scene
    def upload_spheres(self, xyzr):
        a = np.ascontiguousarray(xyzr, np.float32).reshape(-1, 4)
        check(self._lib.pt_upload_spheres(self._ctx, a.ctypes.data, len(a)), self._lib)

    def upload_materials(self, rgbt):
        a = np.ascontiguousarray(rgbt, np.float32).reshape(-1, 4)
        check(self._lib.pt_upload_materials(self._ctx, a.ctypes.data, len(a)), self._lib)

    def upload_mesh(self, vertices, indices):
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 6)
        i = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
        check(self._lib.pt_upload_mesh(self._ctx, v.ctypes.data, len(v), i.ctypes.data, len(i)), self._lib)

    def build_accel(self):
        check(self._lib.pt_build_accel(self._ctx), self._lib)

    def set_accel_mode(self, mode):
        check(self._lib.pt_set_accel_mode(self._ctx, mode), self._lib)

    def set_normal_mode(self, mode):
        check(self._lib.pt_set_normal_mode(self._ctx, mode), self._lib)

    def upload_material_ids(self, ids):
        a = np.ascontiguousarray(ids, np.int32).reshape(-1)
        check(self._lib.pt_upload_material_ids(self._ctx, a.ctypes.data, len(a)), self._lib)

    def set_bvh_builder(self, builder):
        check(self._lib.pt_set_bvh_builder(self._ctx, builder), self._lib)

    def set_primary_reuse(self, enable):
        check(self._lib.pt_set_primary_reuse(self._ctx, 1 if enable else 0), self._lib)

    def set_shading_mode(self, mode):
        check(self._lib.pt_set_shading_mode(self._ctx, mode), self._lib)

    # -- target
    def target_create(self, width, height):
        check(self._lib.pt_target_create(self._ctx, width, height), self._lib)
        self.width, self.height = width, height

    def target_bind_external(self, ptr_a, ptr_b, width, height):
        check(self._lib.pt_target_bind_external(self._ctx, C.c_void_p(ptr_a), C.c_void_p(ptr_b), width, height), self._lib)
        self.width, self.height = width, height

    def target_clear(self):
        check(self._lib.pt_target_clear(self._ctx), self._lib)

    def set_window(self, x0=0, y0=0, w=0, h=0):
        check(self._lib.pt_set_window(self._ctx, x0, y0, w, h), self._lib)

    def set_row_interleave(self, block_rows=16, index=0, count=1):
        """Render only the row blocks `index` modulo `count` (image-tile partition in one dispatch); count <= 1: all rows."""
        check(self._lib.pt_set_row_interleave(self._ctx, block_rows, index, count), self._lib)

    def set_accum_mode(self, mode):
        check(self._lib.pt_set_accum_mode(self._ctx, mode), self._lib)

    def set_cpu_tracer(self, seed=1, ray_max_t=99.9):
        check(self._lib.pt_set_cpu_tracer(self._ctx, seed, ray_max_t), self._lib)

    # -- per frame
    def set_uniforms(self, camera12, num_triangles, depth=10):
        cam = np.asarray(camera12, np.float32).reshape(12)
        u = PtUniforms()
        for k in range(3):
            u.camera_position[k] = cam[k]
            u.camera_lower_left_corner[k] = cam[3 + k]
            u.camera_horizontal[k] = cam[6 + k]
            u.camera_vertical[k] = cam[9 + k]
        u.num_triangles, u.depth = int(num_triangles), int(depth)
        check(self._lib.pt_set_uniforms(self._ctx, C.byref(u)), self._lib)

    def dispatch(self, first_sample, num_samples=1, stride=1):
        check(self._lib.pt_dispatch(self._ctx, first_sample, num_samples, stride), self._lib)

    def dispatch_static(self, first_sample, num_samples=1, stride=1):
        """pt_dispatch that allocates nothing (fixed scratch array in the library's device image)."""
        check(self._lib.pt_dispatch_static(self._ctx, first_sample, num_samples, stride), self._lib)

    def finish(self):
        check(self._lib.pt_finish(self._ctx), self._lib)

    # -- results
    def readback(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        check(self._lib.pt_readback_rgba32f(self._ctx, out.ctypes.data), self._lib)
        return out

    def upload_image(self, rgba):
        """Makes `rgba` ([height, width, 4] float32) the current image: the state a saved progressive render continues from."""
        a = np.ascontiguousarray(rgba, np.float32)
        if a.shape != (self.height, self.width, 4):
            raise ValueError(f"image of shape {a.shape} for a {self.width}x{self.height} target")
        check(self._lib.pt_upload_rgba32f(self._ctx, a.ctypes.data), self._lib)

    def present(self):
        out = np.empty((self.height, self.width, 4), np.uint8)
        check(self._lib.pt_present_rgba8(self._ctx, out.ctypes.data), self._lib)
        return out

    def current_image_ptr(self):
        p = C.c_void_p()
        check(self._lib.pt_current_image(self._ctx, C.byref(p)), self._lib)
        return p.value

    def readback_primary(self):
        ids = np.empty((self.height, self.width), np.int32)
        t = np.empty((self.height, self.width), np.float32)
        check(self._lib.pt_readback_primary(self._ctx, ids.ctypes.data, t.ctypes.data), self._lib)
        return ids, t

    def enable_timing(self, on=True):
        check(self._lib.pt_enable_timing(self._ctx, 1 if on else 0), self._lib)

    def stats(self):
        s = PtStats()
        check(self._lib.pt_get_stats(self._ctx, C.byref(s)), self._lib)
        return {"rays": s.rays, "rays_traced": s.rays_traced, "samples": s.samples, "lane_slots": s.lane_slots, "bvh_node_fetches": s.bvh_node_fetches,
                "bvh_triangle_fetches": s.bvh_triangle_fetches, "grid_triangle_tests": s.grid_triangle_tests,
                "grid_entries": s.grid_entries, "shaded_hits": s.shaded_hits, "loop_triangle_visits": s.loop_triangle_visits,
                "loop_full_tests": s.loop_full_tests, "lane_walk_records": s.lane_walk_records, "dispatches": s.dispatches, "bvh_sweeps": s.bvh_sweeps, "grid_lists_carried": s.grid_lists_carried, "gathered_pixels": s.gathered_pixels, "gathered_tiles": s.gathered_tiles,
                "last_dispatch_ms": s.last_dispatch_ms, "total_dispatch_ms": s.total_dispatch_ms}

    def reset_stats(self):
        check(self._lib.pt_reset_stats(self._ctx), self._lib)

    def verify_accel(self, bounces=1):
        """(queries compared, ids differing, distances differing) of the selected search structure against the all-triangles loop."""
        out = (C.c_uint64 * 3)()
        check(self._lib.pt_verify_accel(self._ctx, bounces, out), self._lib)
        return tuple(int(v) for v in out)

    def selftest_arithmetic(self):
        """(reciprocal, square root, log2 quotient): inputs for which the kernels' short sequences differ from IEEE; all 0."""
        out = (C.c_uint64 * 3)()
        check(self._lib.pt_selftest_arithmetic(self._ctx, out), self._lib)
        return tuple(int(v) for v in out)
