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


class SceneData:
    def __init__(self, name, width=None, height=None):
        obj, view, program, w0, h0 = WORKLOADS[name]
        self.name, self.program = name, program
        self.width, self.height = width or w0, height or h0
        o = host_api.ObjFile(os.path.join(ASSETS, obj), ASSETS + "/")
        mesh = o.mesh_default if view["translate_z"] == 3.0 else o.mesh_untranslated
        self.vertices, self.indices = mesh["vertices"], mesh["indices"]
        self.material_ids = mesh["material_ids"]
        self.normal_mode = capi.NORMALS_VERTEX if mesh["has_vertex_normals"] else capi.NORMALS_GEOMETRIC
        self.camera = host_api.camera(view["fov"], self.width, self.height, view["eye"], view["target"], view["up"])
        self.spheres, self.materials = host_api.reference_spheres()
        self.num_triangles = len(self.indices)

    def upload(self, tracer, depth=10):
        """Scene upload + program + uniforms on a Tracer; the target is the caller's business."""
        prog = tracer.program_create(self.program)
        tracer.use_program(prog)
        tracer.upload_spheres(self.spheres)
        tracer.upload_materials(self.materials)
        tracer.upload_mesh(self.vertices, self.indices)
        tracer.set_normal_mode(self.normal_mode)
        tracer.set_uniforms(self.camera, self.num_triangles, depth)
        return prog
