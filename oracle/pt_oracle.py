"""TEST INFRASTRUCTURE ONLY: ctypes front end of the parity oracle (oracle/pt_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as the
checker or as the CPU figure reported beside the GPU one. Nothing in pathtracer_amd/ imports it.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libpt_oracle.so")
_lib = None


class PtoScene(C.Structure):
    _fields_ = [("num_triangles", C.c_int32), ("num_vertices", C.c_int32),
                ("indices", C.c_void_p), ("vertices", C.c_void_p),
                ("spheres", C.c_void_p), ("materials", C.c_void_p), ("material_ids", C.c_void_p)]


class PtoUniforms(C.Structure):
    _fields_ = [("cam_pos", C.c_float * 3), ("cam_llc", C.c_float * 3),
                ("cam_hor", C.c_float * 3), ("cam_ver", C.c_float * 3),
                ("num_triangles", C.c_int32), ("depth", C.c_int32),
                ("width", C.c_int32), ("height", C.c_int32),
                ("variant", C.c_int32), ("normal_mode", C.c_int32), ("shading_mode", C.c_int32)]


def build(force=False):
    """Compile the oracle with gcc (seconds). Safe to call repeatedly."""
    src = os.path.join(_HERE, "pt_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_build/libpt_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.pto_render.restype = C.c_uint64
        L.pto_render.argtypes = [C.POINTER(PtoScene), C.POINTER(PtoUniforms), C.c_int, C.c_int, C.c_void_p,
                                 C.c_int, C.c_int, C.c_int, C.c_int]
        L.pto_render_sum.restype = C.c_uint64
        L.pto_render_sum.argtypes = [C.POINTER(PtoScene), C.POINTER(PtoUniforms), C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.pto_primary.restype = None
        L.pto_primary.argtypes = [C.POINTER(PtoScene), C.POINTER(PtoUniforms), C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_int, C.c_int]
        L.pto_cpu_tracer.restype = C.c_uint64
        L.pto_cpu_tracer.argtypes = [C.POINTER(PtoScene), C.POINTER(PtoUniforms), C.c_float, C.c_uint32, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.pto_rng_stream.restype = None
        L.pto_rng_stream.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.pto_camera.restype = None
        L.pto_camera.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pto_present.restype = None
        L.pto_present.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.pto_lp_math.restype = None
        L.pto_lp_math.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


VARIANT_WORLD, VARIANT_MESH = 0, 1
NORMALS_VERTEX, NORMALS_GEOMETRIC = 0, 1


class Scene:
    """Host-side scene arrays in the reference's upload layouts (PathTracer.cpp:1253-1288)."""

    def __init__(self, vertices=None, indices=None, spheres=None, materials=None, material_ids=None):
        self.vertices = np.ascontiguousarray(vertices if vertices is not None else np.zeros((1, 6)), np.float32).reshape(-1, 6)
        self.indices = np.ascontiguousarray(indices if indices is not None else np.zeros((0, 3)), np.uint32).reshape(-1, 3)
        self.spheres = np.ascontiguousarray(spheres if spheres is not None else np.zeros((10, 4)), np.float32).reshape(10, 4)
        m = np.asarray(materials if materials is not None else np.zeros((10, 4)), np.float32).reshape(-1, 4)
        if len(m) > 10:
            raise ValueError("the reference's materials texture holds 10 entries")
        self.materials = np.zeros((10, 4), np.float32)
        self.materials[:len(m)] = m
        if self.indices.size and int(self.indices.max()) >= len(self.vertices):
            raise ValueError("triangle index out of range")
        self.material_ids = np.ascontiguousarray(material_ids if material_ids is not None else np.zeros(len(self.indices)), np.int32)
        if len(self.material_ids) != len(self.indices) or (self.material_ids.size and (self.material_ids.min() < 0 or self.material_ids.max() > 9)):
            raise ValueError("material ids must be one per triangle, in 0..9")

    def c_struct(self):
        return PtoScene(len(self.indices), len(self.vertices), self.indices.ctypes.data, self.vertices.ctypes.data,
                        self.spheres.ctypes.data, self.materials.ctypes.data, self.material_ids.ctypes.data)


SHADE_REFERENCE, SHADE_MATERIALS, SHADE_MATERIALS_UNIT = 0, 1, 2


def uniforms(camera12, width, height, num_triangles, depth=10, variant=VARIANT_MESH, normal_mode=NORMALS_VERTEX,
             shading_mode=SHADE_REFERENCE):
    cam = np.asarray(camera12, np.float32).reshape(12)
    u = PtoUniforms()
    for i in range(3):
        u.cam_pos[i], u.cam_llc[i], u.cam_hor[i], u.cam_ver[i] = cam[i], cam[3 + i], cam[6 + i], cam[9 + i]
    u.num_triangles, u.depth, u.width, u.height = num_triangles, depth, width, height
    u.variant, u.normal_mode, u.shading_mode = variant, normal_mode, shading_mode
    return u


def material_table_from_mtl(mat_values, unassigned=(0.5, 0.5, 0.5)):
    """Build-defined rule that turns MTL statements into the (r, g, b, e) table of the "materials" shading mode:
    emission = Ke, or Ka when Ke is zero and Ka exceeds 1 (the Cornell file's light is `Ka 20 20 20` and has no Ke);
    an emitter is stored as (colour / peak, peak); everything else as (Kd, 0). One extra grey diffuse entry is
    appended for triangles without a `usemtl`. mat_values: rows of Ka Kd Ks Kt Ke (3 each), ..."""
    rows = []
    for m in np.asarray(mat_values, np.float32).reshape(-1, 19):
        ka, kd, ke = m[0:3], m[3:6], m[12:15]
        em = ke if ke.max() > 0 else (ka if ka.max() > 1.0 else None)
        if em is not None:
            peak = np.float32(em.max())
            rows.append([em[0] / peak, em[1] / peak, em[2] / peak, peak])
        else:
            rows.append([kd[0], kd[1], kd[2], 0.0])
    rows.append([unassigned[0], unassigned[1], unassigned[2], 0.0])
    return np.asarray(rows, np.float32)


def camera(fov_deg, width, height, eye, target, up):
    out = np.zeros(12, np.float32)
    e, t, u = (np.asarray(v, np.float32) for v in (eye, target, up))
    lib().pto_camera(fov_deg, float(width), float(height), e.ctypes.data, t.ctypes.data, u.ctypes.data, out.ctypes.data)
    return out


def _row_chunks(y0, h, threads):
    threads = max(1, min(threads, h))
    edges = np.linspace(y0, y0 + h, threads + 1).astype(int)
    return [(int(a), int(b - a)) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def _parallel(fn, chunks):
    out = [None] * len(chunks)

    def run(i, c):
        out[i] = fn(*c)
    ts = [threading.Thread(target=run, args=(i, c)) for i, c in enumerate(chunks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return out


def render(scene, u, first_sample, num_samples, rgba=None, window=None, threads=1):
    """Progressive passes first_sample..first_sample+num_samples-1 applied in place; returns (rgba, rays)."""
    W, H = u.width, u.height
    if rgba is None:
        rgba = np.zeros((H, W, 4), np.float32)
        rgba[..., 3] = 1.0  # glClearColor(0,0,0,1), PathTracer.cpp:1108,1151
    assert rgba.dtype == np.float32 and rgba.shape == (H, W, 4) and rgba.flags.c_contiguous
    x0, y0, w, h = window if window else (0, 0, W, H)
    sc = scene.c_struct()
    L = lib()
    rays = _parallel(lambda ya, hh: L.pto_render(C.byref(sc), C.byref(u), first_sample, num_samples, rgba.ctypes.data,
                                                 x0, ya, w, hh), _row_chunks(y0, h, threads))
    return rgba, int(sum(rays))


def render_sum(scene, u, first_sample, stride, count, rgba_sum=None, window=None, threads=1):
    W, H = u.width, u.height
    if rgba_sum is None:
        rgba_sum = np.zeros((H, W, 4), np.float32)
    x0, y0, w, h = window if window else (0, 0, W, H)
    sc = scene.c_struct()
    L = lib()
    rays = _parallel(lambda ya, hh: L.pto_render_sum(C.byref(sc), C.byref(u), first_sample, stride, count,
                                                     rgba_sum.ctypes.data, x0, ya, w, hh), _row_chunks(y0, h, threads))
    return rgba_sum, int(sum(rays))


def cpu_tracer(scene, u, first_sample, stride, count, rgba_sum=None, max_t=99.9, seed=1, window=None, threads=1):
    """The reference's CPU tracer (Scene::color) restated with a counter-based generator: running sums of `count` samples added
    to rgba_sum ([H, W, 4]: rgb sums + the number of samples); returns (rgba_sum, rays)."""
    W, H = u.width, u.height
    if rgba_sum is None:
        rgba_sum = np.zeros((H, W, 4), np.float32)
    x0, y0, w, h = window if window else (0, 0, W, H)
    sc = scene.c_struct()
    L = lib()
    rays = _parallel(lambda ya, hh: L.pto_cpu_tracer(C.byref(sc), C.byref(u), max_t, seed, first_sample, stride, count,
                                                     rgba_sum.ctypes.data, x0, ya, w, hh), _row_chunks(y0, h, threads))
    return rgba_sum, int(sum(rays))


def primary(scene, u, window=None, threads=1):
    """Primary-hit ids (int32, -1 = miss), t and direction for every pixel of the window (full-frame arrays)."""
    W, H = u.width, u.height
    ids = np.full((H, W), -1, np.int32)
    ts = np.zeros((H, W), np.float32)
    rds = np.zeros((H, W, 3), np.float32)
    x0, y0, w, h = window if window else (0, 0, W, H)
    sc = scene.c_struct()
    L = lib()
    _parallel(lambda ya, hh: L.pto_primary(C.byref(sc), C.byref(u), ids.ctypes.data, ts.ctypes.data, rds.ctypes.data,
                                           x0, ya, w, hh), _row_chunks(y0, h, threads))
    return ids, ts, rds


def present(rgba):
    """copy.fs on an 8-bit target: [H, W, 4] float32 -> [H, W, 4] uint8."""
    rgba = np.ascontiguousarray(rgba, np.float32)
    out = np.zeros(rgba.shape, np.uint8)
    lib().pto_present(rgba.ctypes.data, rgba.size // 4, out.ctypes.data)
    return out


def rng_stream(px, py, samples, n):
    out = np.zeros(n, np.float32)
    lib().pto_rng_stream(px, py, float(samples), n, out.ctypes.data)
    return out


def lp_math(kind, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), np.float32)
    out = np.zeros_like(a)
    lib().pto_lp_math({"sin": 0, "cos": 1, "exp2": 2, "pow": 3}[kind], a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


# ---------------------------------------------------------------------------------------------------------
# Scene construction as the reference host does it (restated; used to build oracle/test inputs)

def mesh_from_obj_arrays(v, idx_v, vn=None, translation=(0.0, 0.0, 3.0), scale=1.0):
    """PathTracer.cpp:1178-1209: position = v*scale + (0,0,tz) (only z is translated), normal = attrib.normals
    read with the POSITION index; where the OBJ carries fewer normals than positions that read is out of bounds
    in the reference, so zeros are stored here and such scenes must be traced with geometric normals."""
    v = np.asarray(v, np.float32).reshape(-1, 3)
    n = len(v)
    verts = np.zeros((n, 6), np.float32)
    s = np.float32(scale)
    verts[:, 0] = v[:, 0] * s
    verts[:, 1] = v[:, 1] * s
    verts[:, 2] = v[:, 2] * s + np.float32(translation[2])
    if vn is not None and len(np.asarray(vn).reshape(-1, 3)) >= n:
        verts[:, 3:6] = np.asarray(vn, np.float32).reshape(-1, 3)[:n]
    return verts, np.asarray(idx_v, np.uint32).reshape(-1, 3)


def reference_spheres_and_materials():
    """PathTracer.cpp:1216-1244 with glibc: srand(2 ^ 17 - 1) is srand(18) (XOR), then 10 spheres and 10
    materials drawn from one rand() stream through nextFloat (PathTracer.cpp:765-768)."""
    libc = C.CDLL("libc.so.6")
    libc.srand(2 ^ 17 - 1)
    rmax = np.float32(2147483647)

    def nf(lo, hi):
        lo, hi = np.float32(lo), np.float32(hi)
        return np.float32(lo + (np.float32(libc.rand()) / rmax) * (hi - lo))
    sph = np.zeros((10, 4), np.float32)
    for i in range(10):
        sph[i] = [nf(-3, 3), nf(-3, 3), nf(0.5, 10), nf(0.1, 1)]
    mat = np.zeros((10, 4), np.float32)
    for i in range(10):
        kind = 2 if i == 7 else (0 if i < 7 else 1)  # LAMBERT / METAL / DIELECTRIC enum, PathTracer.cpp:89
        mat[i] = [nf(0, 1), nf(0, 1), nf(0, 1), kind]
    return sph, mat
