"""Named workloads: which OBJ, which camera, which program. Scene data comes from the C++ host library (the same
code path a C++ host uses); nothing here computes geometry itself."""
import os

import numpy as np

from . import capi, host_api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")

# The reference hard-codes eye (0,0,0) looking down +z with a 60 degree vertical field of view and moves the mesh to
# z+3 (PathTracer.cpp:1160-1162, 1178-1183). That frames the monkey heads; the Cornell box (0..556 units) needs a
# camera of its own, which is a choice of this build: the classic Cornell view.
REFERENCE_VIEW = dict(fov=60.0, eye=(0.0, 0.0, 0.0), target=(0.0, 0.0, 1.0), up=(0.0, 1.0, 0.0), translate_z=3.0)
CORNELL_VIEW = dict(fov=39.3, eye=(278.0, 273.0, -800.0), target=(278.0, 273.0, 0.0), up=(0.0, 1.0, 0.0), translate_z=0.0)

WORKLOADS = {
    # name: (obj, view, program, default width, height)
    "cornell": ("cornell_box.obj", CORNELL_VIEW, capi.PROGRAM_MESH, 1920, 1080),
    "monkeyhead2": ("monkeyhead2.obj", REFERENCE_VIEW, capi.PROGRAM_MESH, 1920, 1080),
    "suzanne": ("suzanne.obj", REFERENCE_VIEW, capi.PROGRAM_MESH, 1920, 1080),
    "torus": ("catmark_torus_creases0.obj", REFERENCE_VIEW, capi.PROGRAM_MESH, 3840, 2160),
    "world": ("monkeyhead2.obj", REFERENCE_VIEW, capi.PROGRAM_WORLD, 1280, 720),
}


def _lcg_uniform(count, seed=20260101):
    """Deterministic uniform [0,1) stream: 32-bit LCG (Numerical Recipes constants), top 24 bits."""
    out = np.empty(count, np.float64)
    state = seed & 0xFFFFFFFF
    for i in range(count):
        state = (1664525 * state + 1013904223) & 0xFFFFFFFF
        out[i] = (state >> 8) / float(1 << 24)
    return out


def instanced_mesh(vertices, indices, instances=254, columns=16, spacing=2.8, unit=30.0, seed=20260101):
    """Synthetic large scene: `instances` copies of a mesh on a jittered columns-wide grid facing the camera, each
    with its own yaw, jitter and depth from a fixed LCG seed, everything scaled to Cornell-like units so that the
    shader's absolute `det >= 1e-4` cull keeps the small triangles. Returns (vertices6, indices)."""
    v = np.asarray(vertices, np.float32).reshape(-1, 6)[:, :3].astype(np.float64)
    idx = np.asarray(indices, np.uint32).reshape(-1, 3)
    rnd = _lcg_uniform(instances * 4, seed).reshape(instances, 4)
    rows = (instances + columns - 1) // columns
    out_v = np.zeros((instances * len(v), 6), np.float32)
    out_i = np.zeros((instances * len(idx), 3), np.uint32)
    for k in range(instances):
        cx = (k % columns - (columns - 1) / 2.0) * spacing + (rnd[k, 0] - 0.5) * 0.8
        cy = (k // columns - (rows - 1) / 2.0) * spacing + (rnd[k, 1] - 0.5) * 0.8
        cz = (rnd[k, 2] - 0.5) * 6.0
        yaw = (rnd[k, 3] - 0.5) * 1.0
        c, s_ = np.cos(yaw), np.sin(yaw)
        x = v[:, 0] * c + v[:, 2] * s_
        z = -v[:, 0] * s_ + v[:, 2] * c
        p = np.stack([x + cx, v[:, 1] + cy, z + cz], 1) * unit
        out_v[k * len(v):(k + 1) * len(v), :3] = p.astype(np.float32)
        out_i[k * len(idx):(k + 1) * len(idx)] = idx + np.uint32(k * len(v))
    return out_v, out_i


def distorted_copy(vertices, amplitude=0.12, offset=(0.0, 0.0, 0.0), seed=20260101):
    """A deterministically distorted copy of a mesh: every vertex moved by up to `amplitude` along each axis with offsets
    from the fixed LCG stream (integer arithmetic, so the same on every machine), then translated by `offset`. Stands in
    for the reference's data/distordedmesh/*.bin, a container that nothing in the reference loads. Returns vertices6."""
    v = np.asarray(vertices, np.float32).reshape(-1, 6).astype(np.float64)
    rnd = _lcg_uniform(len(v) * 3, seed).reshape(len(v), 3)
    out = np.zeros((len(v), 6), np.float32)
    out[:, :3] = (v[:, :3] + (rnd - 0.5) * (2.0 * amplitude) + np.asarray(offset, np.float64)).astype(np.float32)
    return out


def torus_with_distorted_copy(vertices, indices, separation=1.4):
    """The torus moved `separation` to the left, and its distorted copy the same distance to the right: one mesh of twice the
    triangles (ids of the copy follow those of the original). Edges are about 0.3-1 unit long, far above what the shader's
    absolute `det >= 1e-4` cull removes."""
    v = np.asarray(vertices, np.float32).reshape(-1, 6)
    idx = np.asarray(indices, np.uint32).reshape(-1, 3)
    left = v.copy()
    left[:, 0] = (left[:, 0].astype(np.float64) - separation).astype(np.float32)
    right = distorted_copy(v, offset=(separation, 0.0, 0.0))
    return np.concatenate([left, right]), np.concatenate([idx, idx + np.uint32(len(v))])


class SceneData:
    def __init__(self, name, width=None, height=None):
        if name == "instanced":
            self._init_instanced(width, height)
            return
        if name == "torus_distorted":
            self._init_torus_distorted(width, height)
            return
        obj, view, program, w0, h0 = WORKLOADS[name]
        self.name, self.program = name, program
        self.width, self.height = width or w0, height or h0
        o = host_api.ObjFile(os.path.join(ASSETS, obj), ASSETS + "/")
        mesh = o.mesh_default if view["translate_z"] == 3.0 else o.mesh_untranslated
        self.vertices, self.indices = mesh["vertices"], mesh["indices"]
        # rows for the diffuse/emissive shading mode: MTL-derived table, "no material" -> its last row
        self.material_table = o.material_table
        self.material_ids = np.where(mesh["material_ids"] < 0, len(o.material_table) - 1, mesh["material_ids"]).astype(np.int32)
        self.normal_mode = capi.NORMALS_VERTEX if mesh["has_vertex_normals"] else capi.NORMALS_GEOMETRIC
        self.camera = host_api.camera(view["fov"], self.width, self.height, view["eye"], view["target"], view["up"])
        self.spheres, self.materials = host_api.reference_spheres()
        self.num_triangles = len(self.indices)

    def _init_instanced(self, width, height):
        """~1M triangles: 254 copies of suzanne.obj (3936 triangles each = 999,744), geometric normals, camera on the
        grid's axis far enough to see all of it."""
        self.name, self.program = "instanced", capi.PROGRAM_MESH
        self.width, self.height = width or 3840, height or 2160
        o = host_api.ObjFile(os.path.join(ASSETS, "suzanne.obj"), ASSETS + "/")
        base = o.mesh_untranslated
        unit = 30.0
        self.vertices, self.indices = instanced_mesh(base["vertices"], base["indices"], unit=unit)
        # diffuse/emissive mode: one colour per instance, every 16th instance a lamp
        tris_per = len(base["indices"])
        inst = np.arange(len(self.indices)) // tris_per
        self.material_ids = (inst % 8).astype(np.int32)
        self.material_table = np.array([[0.8, 0.8, 0.8, 0], [0.8, 0.3, 0.3, 0], [0.3, 0.8, 0.3, 0], [0.3, 0.3, 0.8, 0],
                                        [0.8, 0.8, 0.3, 0], [0.3, 0.8, 0.8, 0], [0.8, 0.3, 0.8, 0], [1.0, 0.9, 0.7, 6.0]], np.float32)
        self.normal_mode = capi.NORMALS_GEOMETRIC
        self.camera = host_api.camera(60.0, self.width, self.height, (0.0, 0.0, -44.0 * unit), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
        self.spheres, self.materials = host_api.reference_spheres()
        self.num_triangles = len(self.indices)

    def _init_torus_distorted(self, width, height):
        """The benchmark's fourth configuration: catmark_torus_creases0.obj next to a distorted copy of itself, 3840x2160, the
        reference's camera and its z + 3 translation, geometric normals (the file has no vn)."""
        self.name, self.program = "torus_distorted", capi.PROGRAM_MESH
        self.width, self.height = width or 3840, height or 2160
        o = host_api.ObjFile(os.path.join(ASSETS, "catmark_torus_creases0.obj"), ASSETS + "/")
        base = o.mesh_default
        self.vertices, self.indices = torus_with_distorted_copy(base["vertices"], base["indices"])
        self.material_table = np.array([[0.75, 0.75, 0.75, 0], [0.3, 0.6, 0.8, 0]], np.float32)
        self.material_ids = (np.arange(len(self.indices)) >= len(base["indices"])).astype(np.int32)
        self.normal_mode = capi.NORMALS_GEOMETRIC
        v = REFERENCE_VIEW
        self.camera = host_api.camera(v["fov"], self.width, self.height, v["eye"], v["target"], v["up"])
        self.spheres, self.materials = host_api.reference_spheres()
        self.num_triangles = len(self.indices)

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
