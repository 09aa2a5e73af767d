"""Shared test helpers: golden-case loading and bit-pattern comparison."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = ["monkey_mesh_128", "monkey_mesh_160x90", "monkey_mesh_1080p_crop", "world_128", "world_1280x720_crop",
         "cornell_geo_128", "cornell_geo_1080p_crop", "suzanne_geo_96",
         # build-defined diffuse/emissive mode: llvmpipe ran the reference shader with oracle/llvmpipe/edits/mesh_materials_geo.txt
         "cornell_materials_128", "cornell_materials_1080p_crop",
         # the same mode with albedos of 1e-13..1e-15: throughputs, sky terms and the running mean pass through the denormal range
         "cornell_materials_tiny_96",
         # the Lambert bounce in the form of pathtracing.fs:272, rd = normalize(N + randomUnitVector()) (edits/mesh_materials_unit_geo.txt)
         "cornell_materials_unit_128",
         # the same mode on the reference's own mesh with its vertex normals (edits/mesh_materials.txt): 968 triangles, so
         # PT_ACCEL_AUTO renders it through the hierarchy
         "monkey_materials_128",
         # the benchmark's third configuration in that mode: crops of the 1920x1080 frames of monkeyhead2.obj and suzanne.obj on
         # their silhouettes, and suzanne (3,936 triangles, geometric normals) as a whole small frame
         "monkey_materials_1080p_crop", "suzanne_materials_1080p_crop", "suzanne_materials_96",
         # the benchmark's fourth configuration: the torus, and the torus next to its deterministically distorted copy
         "torus_geo_128", "torus_distorted_geo_128", "torus_distorted_geo_256x144", "torus_distorted_4k_crop"]


def case_shading_mode(fx):
    return int(fx["shading_mode"]) if "shading_mode" in fx else 0


def load_case(name):
    return dict(np.load(os.path.join(GOLDEN, f"case_{name}.npz")))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def n_bit_mismatch(a, b):
    return int((bits(a) != bits(b)).sum())
