"""Host-side logic on the CPU: the product's own OBJ/MTL reader, scene construction, camera and Shader stage
selection, checked against dumps of the REFERENCE's loader (tests/golden/obj_*.npz, made by running the
reference's vendored parser in the build container) and against the oracle's independent restatements."""
import os

import numpy as np
import pytest

import pt_oracle as po
from helpers import GOLDEN, n_bit_mismatch
from pathtracer_amd import host_api as host

ASSETS = ["cornell_box", "monkeyhead2", "suzanne", "catmark_torus_creases0", "cube", "no_material", "monkeyhead"]
A = os.path.join(os.path.dirname(os.path.dirname(GOLDEN)), "assets")


@pytest.mark.parametrize("name", ASSETS)
def test_obj_reader_equals_reference_reader(name):
    ref = np.load(f"{GOLDEN}/obj_{name}.npz")
    o = host.ObjFile(f"{A}/{name}.obj", A + "/")
    assert n_bit_mismatch(o.v, ref["v"]) == 0
    assert o.vn.shape == ref["vn"].shape and n_bit_mismatch(o.vn, ref["vn"]) == 0
    assert np.array_equal(o.idx_v, ref["idx_v"])
    assert np.array_equal(o.idx_vn, ref["idx_vn"])
    assert np.array_equal(o.mat_ids, ref["mat_ids"])
    assert np.array_equal(o.shape_tris, ref["shape_tris"])
    assert o.shape_names == [str(s) for s in ref["shape_names"]]
    ref_names = ["" if s == "<unnamed>" else str(s) for s in ref["mat_names"]]
    assert o.mat_names == ref_names
    assert n_bit_mismatch(o.mat_values, ref["mat_values"].reshape(-1, 19)) == 0


def test_triangle_counts_of_the_assets():
    counts = {"cornell_box": 36, "no_material": 36, "cube": 12, "monkeyhead2": 968, "monkeyhead": 968, "suzanne": 3936,
              "catmark_torus_creases0": 64}
    for n, c in counts.items():
        assert len(host.ObjFile(f"{A}/{n}.obj", A + "/").idx_v) == c


def test_mtl_lookup_follows_basedir_rule(tmp_path):
    # without a base directory the MTL is searched in the working directory only, like the reference
    o = host.ObjFile(f"{A}/cornell_box.obj", None)
    assert "not found" in o.err and len(o.mat_names) == 0 and np.all(o.mat_ids == -1)
    o = host.ObjFile(f"{A}/cornell_box.obj", A + "/")
    assert o.mat_names == ["white", "red", "green", "blue", "light"]
    light = o.mat_values[4]
    assert tuple(light[0:3]) == (20.0, 20.0, 20.0)  # Ka 20 20 20: the "light" is ambient, there is no Ke


def test_material_table_rule(tmp_path):
    """MTL -> (r, g, b, e) rows of the diffuse/emissive mode: the C++ host against the rule as the oracle front end states
    it, the table the golden case was rendered with, and a hand-written MTL with Ke."""
    o = host.ObjFile(f"{A}/cornell_box.obj", A + "/")
    assert o.material_table.shape == (6, 4)
    assert n_bit_mismatch(o.material_table, po.material_table_from_mtl(o.mat_values)) == 0
    fx = np.load(f"{GOLDEN}/case_cornell_materials_128.npz")
    assert n_bit_mismatch(o.material_table, fx["materials"][:6]) == 0
    assert tuple(o.material_table[4]) == (1.0, 1.0, 1.0, 20.0)          # light: Ka 20 20 20 -> white x 20
    assert tuple(o.material_table[5]) == (0.5, 0.5, 0.5, 0.0)            # row for triangles without usemtl
    ids = np.where(o.mat_ids < 0, 5, o.mat_ids)
    assert np.array_equal(ids, fx["material_ids"])
    (tmp_path / "m.mtl").write_text("newmtl lamp\nKd 0.1 0.2 0.3\nKe 2 4 1\nnewmtl wall\nKa 0.9 0.9 0.9\nKd 0.25 0.5 0.75\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl lamp\nf 1 2 3\nusemtl wall\nf 3 2 1\n")
    o = host.ObjFile(str(tmp_path / "m.obj"), str(tmp_path) + "/")
    assert np.allclose(o.material_table, [[0.5, 1.0, 0.25, 4.0], [0.25, 0.5, 0.75, 0.0], [0.5, 0.5, 0.5, 0.0]])
    assert list(o.mat_ids) == [0, 1]


def test_float_text_grammar():
    cases = {"1": 1.0, "-0": -0.0, "+3.1417e+2": 314.17, "-0.0E-3": -0.0, "1.0324": 1.0324, "11e2": 1100.0,
             "552.8": 552.8, "0.437500": 0.4375, "1.": 1.0, "1e": 0.0, ".5": 0.0, "abc": 0.0, "": 0.0, "2.5x": 2.5,
             "-1.41": -1.41, "1e-3": 1e-3, "123456789.123456789": 123456789.123456789}
    for text, want in cases.items():
        got = host.parse_float(text)
        assert got == np.float32(want), (text, got)
    assert np.signbit(host.parse_float("-0"))
    assert host.parse_float("nope", 7.0) == np.float32(7.0)


def test_reader_edge_cases(tmp_path):
    p = tmp_path / "edge.obj"
    p.write_text("# comment\r\n\r\nv 0 0 0\r\nv 1 0 0\nv 1 1 0\n v 0 1 0\nv 2 2 2\n"
                 "vn 0 0 1\nvt 0.5 0.5\n"
                 "f 1/1/1 2//1 3/1 4\n"       # quad, mixed corner forms -> 2 triangles (fan)
                 "f -1 -2 -3\n"                # relative indices
                 "f 1 2\n"                     # degenerate: fewer than 3 corners -> nothing
                 "f 1 2 3 4 5\n"               # pentagon -> 3 triangles
                 "g second\nf 3 2 1\n"
                 "usemtl missing\nf 1 3 5\n")  # unknown material keeps id -1 and the same shape
    o = host.ObjFile(str(p), None)
    assert o.v.shape == (5, 3) and o.vn.shape == (1, 3)
    assert o.idx_v.tolist() == [[0, 1, 2], [0, 2, 3], [4, 3, 2], [0, 1, 2], [0, 2, 3], [0, 3, 4], [2, 1, 0], [0, 2, 4]]
    assert o.idx_vn[0].tolist() == [0, 0, -1] and o.idx_vn[1].tolist() == [0, -1, -1]
    assert o.shape_tris.tolist() == [6, 2] and o.shape_names == ["", "second"]
    assert np.all(o.mat_ids == -1)
    with pytest.raises(FileNotFoundError):
        host.ObjFile(str(tmp_path / "absent.obj"), None)
    empty = tmp_path / "empty.obj"
    empty.write_text("")
    o = host.ObjFile(str(empty), None)
    assert len(o.idx_v) == 0 and len(o.v) == 0


def test_usemtl_then_group_drops_the_open_shape(tmp_path):
    """Kept quirk of the reference reader: triangles moved into the open shape by a material switch are lost when a
    new group starts before any further face."""
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1 0 0\nnewmtl b\nKd 0 1 0\n")
    p = tmp_path / "q.obj"
    p.write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\nusemtl b\ng next\nf 3 2 1\n")
    o = host.ObjFile(str(p), str(tmp_path) + "/")
    assert o.idx_v.tolist() == [[2, 1, 0]] and o.mat_ids.tolist() == [1] and o.shape_names == ["next"]


def test_mesh_build_matches_reference_rule():
    o = host.ObjFile(f"{A}/monkeyhead2.obj", None)
    m = o.mesh_default
    ref = np.load(f"{GOLDEN}/obj_monkeyhead2.npz")
    verts, idx = po.mesh_from_obj_arrays(ref["v"], ref["idx_v"], ref["vn"])
    assert m["has_vertex_normals"]
    assert n_bit_mismatch(m["vertices"], verts) == 0 and np.array_equal(m["indices"], idx)
    fx = np.load(f"{GOLDEN}/case_monkey_mesh_128.npz")
    assert n_bit_mismatch(m["vertices"], fx["vertices"]) == 0 and np.array_equal(m["indices"], fx["indices"])
    c = host.ObjFile(f"{A}/cornell_box.obj", A + "/")
    assert not c.mesh_default["has_vertex_normals"] and np.all(c.mesh_default["vertices"][:, 3:] == 0)
    assert c.mesh_default["material_ids"].tolist()[:8] == [0, 0, 0, 0, 0, 0, 4, 4]


def test_mesh_with_the_files_own_normal_indices():
    """monkeyhead2.obj has 507 positions and 529 normals: the reference pairs normal k with position k, the file pairs
    them through normal_index. The second builder must give the same triangles (same positions, same order) with the
    file's normals on every corner."""
    ref = np.load(f"{GOLDEN}/obj_monkeyhead2.npz")
    o = host.ObjFile(f"{A}/monkeyhead2.obj", A + "/")
    plain, fixed = o.mesh_default, o.mesh_indexed_normals
    assert fixed["has_vertex_normals"] and len(fixed["indices"]) == len(plain["indices"]) == 968
    assert len(fixed["vertices"]) >= 529
    pos_plain = plain["vertices"][plain["indices"].reshape(-1)][:, :3]
    pos_fixed = fixed["vertices"][fixed["indices"].reshape(-1)][:, :3]
    assert n_bit_mismatch(pos_plain, pos_fixed) == 0
    nrm_fixed = fixed["vertices"][fixed["indices"].reshape(-1)][:, 3:]
    assert n_bit_mismatch(nrm_fixed, ref["vn"][ref["idx_vn"].reshape(-1)]) == 0
    # and it differs from the reference's pairing somewhere, otherwise the option would be pointless for this file
    nrm_plain = plain["vertices"][plain["indices"].reshape(-1)][:, 3:]
    assert n_bit_mismatch(nrm_plain, nrm_fixed) > 0
    # files without normals: every corner gets a zero normal, geometric normals are to be used
    c = host.ObjFile(f"{A}/cornell_box.obj", A + "/")
    assert not c.mesh_indexed_normals["has_vertex_normals"] and np.all(c.mesh_indexed_normals["vertices"][:, 3:] == 0)


def test_camera_and_spheres_match_oracle_restatement():
    for (fov, w, h, eye, tgt, up) in [(60.0, 1280, 720, (0, 0, 0), (0, 0, 1), (0, 1, 0)),
                                      (39.3, 1920, 1080, (278, 273, -800), (278, 273, 0), (0, 1, 0)),
                                      (50.0, 173, 61, (0.3, -0.2, -1.0), (0.1, 0.0, 6.0), (0.1, 1.0, 0.0))]:
        assert n_bit_mismatch(host.camera(fov, w, h, eye, tgt, up), po.camera(fov, w, h, eye, tgt, up)) == 0
    s, m = host.reference_spheres()
    d = np.load(f"{GOLDEN}/world_tables_glibc.npz")
    assert n_bit_mismatch(s, d["spheres"]) == 0 and n_bit_mismatch(m, d["materials"]) == 0


def test_shader_stage_selection(tmp_path):
    world = "void main() {\n float hitIndex = IntersectWorld(ro, rd);\n //float hitIndex = IntersectMesh(ro, rd);\n}\n"
    mesh = "float IntersectWorld(vec3 a) { return 0.0; }\nvoid main() {\n //float h = IntersectWorld(ro, rd);\n float h = IntersectMesh(ro, rd);\n}\n"
    assert host.shader_classify_text(world) == "world"
    assert host.shader_classify_text(mesh) == "mesh"
    assert host.shader_classify_text("void main() { /* IntersectMesh( */ }") == "none"
    d = tmp_path / "shaders"
    d.mkdir()
    (d / "pathtracing.fs").write_text(mesh)
    assert host.shader_stage_for_file(str(d / "pathtracing.fs")) == "mesh"
    (d / "pathtracing.fs").write_text(world)
    assert host.shader_stage_for_file(str(d / "pathtracing.fs")) == "world"
    # file absent: the default kind (the shader as committed traces the sphere scene)
    assert host.shader_stage_for_file(str(tmp_path / "nowhere" / "pathtracing.fs")) == "world"
    assert host.shader_stage_for_file("data/shaders/copy.fs") == "present"
    assert host.shader_stage_for_file("data/shaders/opaque.fs") == "none"


def test_config4_scene_is_the_one_the_goldens_were_rendered_from():
    """The torus + distorted-copy workload (the benchmark's fourth configuration) is built from a fixed integer LCG, so it is the same on
    every machine: its arrays equal those stored with the llvmpipe goldens."""
    import numpy as np
    from helpers import load_case
    from pathtracer_amd.scenes import SceneData, distorted_copy
    fx = load_case("torus_distorted_geo_128")
    sc = SceneData("torus_distorted")
    assert sc.vertices.tobytes() == np.ascontiguousarray(fx["vertices"], np.float32).tobytes()
    assert np.array_equal(sc.indices, fx["indices"]) and sc.num_triangles == 128 and len(sc.vertices) == 64
    plain = SceneData("torus")
    assert np.array_equal(plain.indices, fx["indices"][:64])
    moved = np.abs(sc.vertices[32:, :3] - (plain.vertices[:, :3] + np.array([1.4, 0, 0], np.float32)))
    assert moved.max() <= 0.12 + 1e-6 and moved.mean() > 0.02          # distorted, within the stated amplitude
    assert np.array_equal(distorted_copy(plain.vertices), distorted_copy(plain.vertices))
