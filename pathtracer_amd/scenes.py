"""Named workloads: which OBJ, which camera, which program, which material table. Everything comes from the C++ host library
(pathtracer_amd/host/Workloads.cpp -- the code a C++ host and `pt_render --workload` use); nothing here computes geometry."""
import ctypes as C
import os

import numpy as np

from . import capi, host_api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")

WORKLOADS = ("cornell", "monkeyhead2", "suzanne", "torus", "world", "instanced", "torus_distorted")


def _lib():
    L = host_api.load()
    if not getattr(L, "_workloads_bound", False):
        L.pth_workload_build.restype = C.c_void_p
        L.pth_workload_build.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.pth_workload_free.argtypes = [C.c_void_p]
        L.pth_workload_info.argtypes = [C.c_void_p, C.c_void_p]
        L.pth_workload_copy.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.pth_instanced_mesh.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint,
                                         C.c_void_p, C.c_void_p]
        L.pth_distorted_copy.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_uint, C.c_void_p]
        L._workloads_bound = True
    return L


def instanced_mesh(vertices, indices, instances=254, columns=16, spacing=2.8, unit=30.0, seed=20260101):
    """`instances` jittered copies of a mesh on a grid facing the camera (pthost::instancedMesh). Returns (vertices6, indices)."""
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 6)
    idx = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
    out_v = np.zeros((instances * len(v), 6), np.float32)
    out_i = np.zeros((instances * len(idx), 3), np.uint32)
    _lib().pth_instanced_mesh(v.ctypes.data, len(v), idx.ctypes.data, len(idx), instances, columns, spacing, unit, seed,
                              out_v.ctypes.data, out_i.ctypes.data)
    return out_v, out_i


def distorted_copy(vertices, amplitude=0.12, offset=(0.0, 0.0, 0.0), seed=20260101):
    """A deterministically distorted copy of a mesh's vertices (pthost::distortedCopy). Returns vertices6."""
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 6)
    off = np.asarray(offset, np.float64)
    out = np.zeros((len(v), 6), np.float32)
    _lib().pth_distorted_copy(v.ctypes.data, len(v), amplitude, off.ctypes.data, seed, out.ctypes.data)
    return out


class SceneData:
    def __init__(self, name, width=None, height=None):
        L = _lib()
        err = C.create_string_buffer(1024)
        h = L.pth_workload_build(name.encode(), (ASSETS + "/").encode(), width or 0, height or 0, err, len(err))
        if not h:
            raise KeyError(err.value.decode(errors="replace") or name)
        try:
            info = (C.c_int * 7)()
            L.pth_workload_info(h, info)
            self.name = name
            self.program, self.width, self.height = info[0], info[1], info[2]
            nv, nt, rows = info[3], info[4], info[5]
            self.vertices = np.zeros((nv, 6), np.float32)
            self.indices = np.zeros((nt, 3), np.uint32)
            self.material_ids = np.zeros(nt, np.int32)
            # rows for the diffuse/emissive shading mode: MTL-derived table ("no material" -> its last row), or the workload's own
            self.material_table = np.zeros((rows, 4), np.float32)
            self.camera = np.zeros(12, np.float32)
            self.spheres, self.materials = np.zeros((10, 4), np.float32), np.zeros((10, 4), np.float32)
            L.pth_workload_copy(h, self.vertices.ctypes.data, self.indices.ctypes.data, self.material_ids.ctypes.data,
                                self.material_table.ctypes.data, self.camera.ctypes.data, self.spheres.ctypes.data, self.materials.ctypes.data)
            self.normal_mode = capi.NORMALS_VERTEX if info[6] else capi.NORMALS_GEOMETRIC
            self.num_triangles = nt
        finally:
            L.pth_workload_free(h)

    def upload(self, tracer, depth=10, accel=True, shading="reference"):
        """Scene upload + program + uniforms on a Tracer; the target is the caller's business. shading "materials"
        selects the diffuse/emissive mode with the scene's material table (mesh programs only)."""
        prog = tracer.program_create(self.program)
        tracer.use_program(prog)
        tracer.upload_spheres(self.spheres)
        tracer.upload_materials(self.materials)
        tracer.upload_mesh(self.vertices, self.indices)
        tracer.set_normal_mode(self.normal_mode)
        tracer.set_uniforms(self.camera, self.num_triangles, depth)
        if shading in ("materials", "materials_unit"):
            tracer.upload_materials(self.material_table)
            tracer.upload_material_ids(self.material_ids)
            tracer.set_shading_mode(capi.SHADE_MATERIALS if shading == "materials" else capi.SHADE_MATERIALS_UNIT)
        else:
            tracer.set_shading_mode(capi.SHADE_REFERENCE)
        if accel:
            tracer.build_accel()   # used by dispatches when the mesh is large enough (PT_ACCEL_AUTO)
        return prog
