#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: regenerates tests/golden/ by RUNNING THE REFERENCE in the build container.

  * the reference's fragment shader (read from /root/reference at run time, never copied) is executed on Mesa
    llvmpipe through oracle/_ref/lp_harness (see oracle/llvmpipe/lp_harness.cpp), as committed ("world") and
    with the recorded line edits of oracle/llvmpipe/edits/*.txt ("mesh", id instrumentation, geometric normals);
  * the reference's OBJ parser is run through oracle/_ref/tinyobj_dump on every OBJ asset.

Needs /root/reference and Mesa's swrast_dri.so, so it only runs in the build container; the fixtures it writes
are data (inputs + the reference's outputs) and travel with the repository.

Usage:  make -C oracle ref && python3 oracle/gen_golden.py
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("PT_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
ASSETS_DIR = os.path.join(ROOT, "assets")
HARNESS = os.path.join(HERE, "_ref", "lp_harness")
OBJDUMP = os.path.join(HERE, "_ref", "tinyobj_dump")
EDITS = os.path.join(HERE, "llvmpipe", "edits")

sys.path.insert(0, HERE)
import pt_oracle as po  # noqa: E402

ASSETS = ["cornell_box", "monkeyhead2", "suzanne", "catmark_torus_creases0", "cube", "no_material", "monkeyhead"]


def run_obj_dump(name, basedir="data/"):
    out = tempfile.mkdtemp(prefix="tod_")
    r = subprocess.run([OBJDUMP, f"data/{name}.obj", out, basedir], cwd=REF, capture_output=True, text=True)
    info = json.loads(r.stdout.strip().splitlines()[-1])
    d = {
        "v": np.fromfile(f"{out}/v.f32", np.float32).reshape(-1, 3),
        "vn": np.fromfile(f"{out}/vn.f32", np.float32).reshape(-1, 3),
        "idx_v": np.fromfile(f"{out}/idx_v.i32", np.int32).reshape(-1, 3),
        "idx_vn": np.fromfile(f"{out}/idx_vn.i32", np.int32).reshape(-1, 3),
        "mat_ids": np.fromfile(f"{out}/mat_ids.i32", np.int32),
    }
    shapes = [l.rstrip("\n").split(" ", 1) for l in open(f"{out}/shapes.txt")]
    d["shape_tris"] = np.array([int(s[0]) for s in shapes], np.int32)
    d["shape_names"] = np.array([s[1] if len(s) > 1 else "" for s in shapes])
    mats = [l.split() for l in open(f"{out}/materials.txt")]
    d["mat_names"] = np.array([m[0] for m in mats])
    # columns: Ka Kd Ks Kt Ke (3 each), Ns Ni d, illum
    d["mat_values"] = np.array([[float.fromhex(x) for x in m[1:19]] + [float(m[19])] for m in mats], np.float32).reshape(-1, 19)
    shutil.rmtree(out)
    return d, info


def harness(job, tmp):
    """job: dict of harness keys. Returns (final image [h,w,4], dumps {k: image}, info)."""
    lines = []
    for k, v in job.items():
        if isinstance(v, (list, tuple, np.ndarray)):
            v = " ".join(repr(float(x)) if isinstance(x, (float, np.floating)) else str(x) for x in v)
        lines.append(f"{k} {v}")
    lines.append(f"out {tmp}/out.f32")
    open(f"{tmp}/job.txt", "w").write("\n".join(lines) + "\n")
    r = subprocess.run([HARNESS, f"{tmp}/job.txt"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    info = json.loads(r.stdout.strip().splitlines()[-1])
    cw, ch = info["crop"][2], info["crop"][3]
    img = np.fromfile(f"{tmp}/out.f32", np.float32).reshape(ch, cw, 4)
    dumps = {}
    for k in job.get("dump_after", []):
        dumps[int(k)] = np.fromfile(f"{tmp}/out.f32.{k}.f32", np.float32).reshape(ch, cw, 4)
    return img, dumps, info


def scene_files(tmp, scene, with_material_ids=False):
    scene.spheres.tofile(f"{tmp}/spheres.f32")
    mats = np.zeros((10, 4), np.float32)      # the reference's materials texture is 10 texels wide
    mats[:min(10, len(scene.materials))] = scene.materials[:10]
    mats.tofile(f"{tmp}/materials.f32")
    scene.vertices.tofile(f"{tmp}/vertices.f32")
    texw = max(1024, 2 * len(scene.vertices), len(scene.indices))
    out = {"spheres": f"{tmp}/spheres.f32", "materials": f"{tmp}/materials.f32", "vertices": f"{tmp}/vertices.f32",
           "indices": f"{tmp}/indices.u32", "num_triangles": len(scene.indices), "tex_width": texw}
    if with_material_ids:
        np.concatenate([scene.indices, scene.material_ids.astype(np.uint32)[:, None]], 1).astype(np.uint32).tofile(f"{tmp}/indices.u32")
        out["index_components"] = 4
    else:
        scene.indices.tofile(f"{tmp}/indices.u32")
    return out


def cam_floats(cam):
    return [float(np.float32(x)) for x in cam]


def render_case(name, scene, cam, W, H, variant, normal_mode, spp_list, crop=None, depth=10, shading_mode=0):
    """Runs the id-instrumented and the radiance variants on llvmpipe; checks the oracle; returns the fixture."""
    tmp = tempfile.mkdtemp(prefix="lp_")
    os.makedirs(tmp + "/m")
    base = scene_files(tmp, scene)
    rad_base = scene_files(tmp + "/m", scene, True) if shading_mode else None
    base.update({"width": W, "height": H, "depth": depth, "camera": cam_floats(cam)})
    if crop:
        base["crop"] = list(crop)
    geo = "_geo" if normal_mode == po.NORMALS_GEOMETRIC else ""
    ids_edit = {po.VARIANT_WORLD: "world_ids", po.VARIANT_MESH: "mesh_ids" + geo}[variant]
    rad_edit = {po.VARIANT_WORLD: None, po.VARIANT_MESH: "mesh" + geo}[variant]

    job = dict(base, num_samples=1, edits=f"{EDITS}/{ids_edit}.txt")
    idimg, _, _ = harness(job, tmp)
    job = dict(base, num_samples=max(spp_list), dump_after=list(spp_list))
    if rad_edit:
        job["edits"] = f"{EDITS}/{rad_edit}.txt"
    if shading_mode:
        assert variant == po.VARIANT_MESH and (normal_mode == po.NORMALS_GEOMETRIC or shading_mode == 1)
        job.update(rad_base)
        job["edits"] = f"{EDITS}/{'mesh_materials_unit_geo' if shading_mode == 2 else 'mesh_materials' + geo}.txt"
    if not crop:
        job["present_out"] = f"{tmp}/present.u8"   # the reference's copy.fs pass over the final image
    _, dumps, info = harness(job, tmp)
    present = None if crop else np.fromfile(f"{tmp}/present.u8", np.uint8).reshape(H, W, 4)
    shutil.rmtree(tmp)

    fx = {"width": W, "height": H, "depth": depth, "variant": variant, "normal_mode": normal_mode, "shading_mode": shading_mode,
          "material_ids": scene.material_ids,
          "camera": np.asarray(cam, np.float32), "crop": np.asarray(crop if crop else (0, 0, W, H), np.int32),
          "vertices": scene.vertices, "indices": scene.indices, "spheres": scene.spheres, "materials": scene.materials,
          "lp_id": idimg[..., 0].astype(np.int32), "lp_t": idimg[..., 1].copy(),
          "lp_rdx": idimg[..., 2].copy(), "lp_rdy": idimg[..., 3].copy(),
          "spp": np.asarray(spp_list, np.int32)}
    for k, img in dumps.items():
        assert np.all(img[..., 3] == 1.0)
        fx[f"lp_rgb_{k}"] = img[..., :3].copy()
    if present is not None:
        fx["lp_present_rgba8"] = present

    # ---- immediate check of the restatement against what llvmpipe produced
    x0, y0, w, h = fx["crop"]
    u = po.uniforms(cam, W, H, len(scene.indices), depth, variant, normal_mode, shading_mode)
    ids, ts, rds = po.primary(scene, u, window=(x0, y0, w, h), threads=8)
    ids, ts, rds = ids[y0:y0 + h, x0:x0 + w], ts[y0:y0 + h, x0:x0 + w], rds[y0:y0 + h, x0:x0 + w]
    hit = fx["lp_id"] >= 0
    rep = {"case": name, "pixels": int(w * h), "hits": int(hit.sum()),
           "id_mismatch": int((ids != fx["lp_id"]).sum()),
           "t_mismatch": int((ts[hit].view(np.uint32) != fx["lp_t"][hit].view(np.uint32)).sum()),
           "rd_mismatch": int((rds[..., 0].view(np.uint32) != fx["lp_rdx"].view(np.uint32)).sum()
                              + (rds[..., 1].view(np.uint32) != fx["lp_rdy"].view(np.uint32)).sum())}
    rgba = None
    done = 0
    for k in sorted(spp_list):
        rgba, _ = po.render(scene, u, done + 1, k - done, rgba, window=(x0, y0, w, h), threads=8)
        done = k
        mine = rgba[y0:y0 + h, x0:x0 + w, :3]
        ref = fx[f"lp_rgb_{k}"]
        rep[f"rgb_{k}_bit_mismatch"] = int((mine.view(np.uint32) != ref.view(np.uint32)).sum())
        rep[f"rgb_{k}_max_abs"] = float(np.abs(mine - ref).max())
    if present is not None:
        rep["present_byte_mismatch"] = int((po.present(rgba) != present).sum())
    rep["lp_pass_ms_median"] = float(np.median(info["pass_ms"]))
    print(json.dumps(rep))
    return fx, rep


def math_fixture():
    tmp = tempfile.mkdtemp(prefix="lpm_")
    fs = f"{tmp}/probe.fs"
    open(fs, "w").write(
        "uniform sampler2D u_Texture;\nout vec4 FragColor;\nvoid main() {\n"
        "  vec4 v = texelFetch(u_Texture, ivec2(gl_FragCoord.xy), 0);\n"
        "  FragColor = vec4(sin(v.x), cos(v.x), exp2(v.y), pow(v.z, v.w));\n}\n")
    rng = np.random.default_rng(20260101)
    W, H = 128, 64
    n = W * H
    a = np.zeros((n, 4), np.float32)
    a[:, 0] = rng.uniform(0, 2 * np.pi, n)
    a[: n // 8, 0] = rng.uniform(-40, 40, n // 8)           # beyond the shader's phi range, both signs
    a[:, 1] = rng.uniform(-8.7, 0.0, n)
    a[: n // 8, 1] = rng.uniform(-130, 130, n // 8)
    a[:, 2] = rng.uniform(0, 1, n)
    a[:, 3] = np.float32(1.0) / np.float32(1025.0)
    a[n // 2:, 3] = rng.uniform(0.0, 4.0, n - n // 2)
    a[:8, 2] = [0.0, 1.0, 0.5, 0.25, 1e-30, 0.99999994, 2.0, 1e-3]
    a.tofile(f"{tmp}/in.f32")
    img, _, _ = harness({"width": W, "height": H, "fs_override": fs, "init_accum": f"{tmp}/in.f32", "num_samples": 1}, tmp)
    shutil.rmtree(tmp)
    out = img.reshape(n, 4)
    rep = {"case": "lp_math"}
    for i, k in enumerate(["sin", "cos", "exp2", "pow"]):
        mine = po.lp_math(k, a[:, [0, 0, 1, 2][i]], a[:, 3])
        rep[f"{k}_bit_mismatch"] = int((mine.view(np.uint32) != out[:, i].view(np.uint32)).sum())
    print(json.dumps(rep))
    return {"inputs": a, "lp_out": out}, rep


def rng_fixture(scene):
    """Known answers of the reference's rand01() (fs:29-42) evaluated by llvmpipe: 4 successive draws per pixel."""
    tmp = tempfile.mkdtemp(prefix="lpr_")
    ed = f"{tmp}/rng.txt"
    open(ed, "w").write("415\tFragColor = vec4(source, 1.0);\tfloat k0 = rand01(); float k1 = rand01(); float k2 = rand01(); "
                        "float k3 = rand01(); FragColor = vec4(k0, k1, k2, k3);\n")
    W, H = 64, 48
    out = {}
    for s in (1, 7, 1000):
        job = dict(scene_files(tmp, scene), width=W, height=H, depth=0, camera=[0.0] * 12, num_samples=1, first_sample=s, edits=ed)
        img, _, _ = harness(job, tmp)
        out[f"draws_s{s}"] = img.copy()
    shutil.rmtree(tmp)
    bad = 0
    for s in (1, 7, 1000):
        for (x, y) in [(0, 0), (5, 3), (63, 47), (17, 40)]:
            bad += int((po.rng_stream(x, y, s, 4).view(np.uint32) != out[f"draws_s{s}"][y, x].view(np.uint32)).sum())
    print(json.dumps({"case": "rng", "spot_bit_mismatch": bad}))
    return out


def config4_cases():
    """The benchmark's fourth configuration: catmark_torus_creases0.obj and the torus next to its deterministically distorted copy
    (pathtracer_amd/scenes.py builds it; the reference's data/distordedmesh/*.bin is loaded by nothing). Both fit the
    reference's own 1024-texel scene textures (64 / 128 triangles, 32 / 64 vertices), so the unmodified upload path serves.
    Geometric normals: the file has no vn."""
    sys.path.insert(0, ROOT)
    from pathtracer_amd.scenes import SceneData
    sph, mat = po.reference_spheres_and_materials()
    ref_cam = lambda W, H: po.camera(60.0, W, H, (0, 0, 0), (0, 0, 1), (0, 1, 0))
    t = SceneData("torus")
    torus = po.Scene(t.vertices, t.indices, sph, mat)
    d = SceneData("torus_distorted")
    both = po.Scene(d.vertices, d.indices, sph, mat)
    geo = po.NORMALS_GEOMETRIC
    return [("torus_geo_128", torus, ref_cam(128, 128), 128, 128, po.VARIANT_MESH, geo, [1, 2, 16], None),
            ("torus_distorted_geo_128", both, ref_cam(128, 128), 128, 128, po.VARIANT_MESH, geo, [1, 2, 16], None),
            ("torus_distorted_geo_256x144", both, ref_cam(256, 144), 256, 144, po.VARIANT_MESH, geo, [1, 4], None),
            # 64x64 pixels of the 3840x2160 frame on the silhouette of the distorted copy (several triangles and sky)
            ("torus_distorted_4k_crop", both, ref_cam(3840, 2160), 3840, 2160, po.VARIANT_MESH, geo, [1, 4], (2585, 200, 64, 64))]


def tiny_materials_case():
    """The diffuse/emissive mode on the Cornell box with albedos of 1e-13..1e-15 and an emitter of 1e-12: throughputs reach
    1e-39 after three bounces, so products, sky terms, emitted radiance and the running mean all pass through the denormal
    range. Pins how the reference's GL implementation treats denormals (flushed to zero, inputs and results) in every
    operation of the path, including the `energy == 0` path end that a flushed throughput triggers."""
    d, _ = run_obj_dump("cornell_box")
    sph, _ = po.reference_spheres_and_materials()
    cv, ci = po.mesh_from_obj_arrays(d["v"], d["idx_v"], None, translation=(0, 0, 0))
    tab = np.array([[4e-13, 3e-13, 2e-13, 0], [1e-13, 2e-14, 3e-15, 0], [2e-14, 1e-13, 3e-14, 0], [3e-15, 5e-14, 1e-13, 0],
                    [1.0, 0.9, 0.8, 1e-12], [5e-14, 5e-14, 5e-14, 0]], np.float32)   # white red green blue light, no-material
    ids = np.where(d["mat_ids"] < 0, len(tab) - 1, d["mat_ids"])
    scene = po.Scene(cv, ci, sph, tab, ids)
    cam = po.camera(39.3, 96, 96, (278, 273, -800), (278, 273, 0), (0, 1, 0))
    return ("cornell_materials_tiny_96", scene, cam, 96, 96, po.VARIANT_MESH, po.NORMALS_GEOMETRIC, [1, 2, 8], None)


def unit_materials_case():
    """The diffuse/emissive mode with the Lambert bounce of fs:272, rd = normalize(N + randomUnitVector()): Cornell box with
    the table of cornell_box.mtl."""
    d, _ = run_obj_dump("cornell_box")
    sph, _ = po.reference_spheres_and_materials()
    cv, ci = po.mesh_from_obj_arrays(d["v"], d["idx_v"], None, translation=(0, 0, 0))
    tab = po.material_table_from_mtl(d["mat_values"])
    scene = po.Scene(cv, ci, sph, tab, np.where(d["mat_ids"] < 0, len(tab) - 1, d["mat_ids"]))
    cam = po.camera(39.3, 128, 128, (278, 273, -800), (278, 273, 0), (0, 1, 0))
    return ("cornell_materials_unit_128", scene, cam, 128, 128, po.VARIANT_MESH, po.NORMALS_GEOMETRIC, [1, 2, 16], None)


def monkey_materials_case():
    """The diffuse/emissive mode on the reference's own mesh with its vertex normals (monkeyhead2.obj, 968 triangles: more than
    the all-triangles loop is used for, so PT_ACCEL_AUTO renders it through the hierarchy). monkeyhead2.mtl is empty: row 0 is
    the grey row the MTL rule gives triangles without a material; three more rows (two albedos and an emitter) are dealt to
    the triangles by index so that every branch of the mode runs."""
    d, _ = run_obj_dump("monkeyhead2")
    sph, _ = po.reference_spheres_and_materials()
    mv, mi = po.mesh_from_obj_arrays(d["v"], d["idx_v"], d["vn"])
    tab = np.array([[0.5, 0.5, 0.5, 0.0], [0.85, 0.3, 0.2, 0.0], [0.25, 0.7, 0.35, 0.0], [1.0, 0.9, 0.7, 3.0]], np.float32)
    t = np.arange(len(mi) // 3 if mi.ndim == 1 else len(mi))
    ids = np.where(t % 11 == 5, 3, t % 3).astype(np.int32)
    scene = po.Scene(mv, mi, sph, tab, ids)
    cam = po.camera(60.0, 128, 128, (0, 0, 0), (0, 0, 1), (0, 1, 0))
    return ("monkey_materials_128", scene, cam, 128, 128, po.VARIANT_MESH, po.NORMALS_VERTEX, [1, 2, 16], None)


MONKEY_TABLE = np.array([[0.5, 0.5, 0.5, 0.0], [0.85, 0.3, 0.2, 0.0], [0.25, 0.7, 0.35, 0.0], [1.0, 0.9, 0.7, 3.0]], np.float32)


def config3_materials_cases():
    """The benchmark's third configuration in the diffuse/emissive mode: monkeyhead2.obj and suzanne.obj at 1920x1080 (64x64 crops
    on their silhouettes) and suzanne as a whole small frame. Neither file brings materials (monkeyhead2.mtl is empty, suzanne's
    monkey.mtl is not in the reference), so the MTL rule gives every triangle the grey row; as in monkey_materials_128 three more
    rows (two albedos and an emitter) are dealt to the triangles by index so that every branch of the mode runs."""
    sph, _ = po.reference_spheres_and_materials()
    ref_cam = lambda W, H: po.camera(60.0, W, H, (0, 0, 0), (0, 0, 1), (0, 1, 0))
    out = []
    for obj in ("monkeyhead2", "suzanne"):
        d, _ = run_obj_dump(obj)
        # the reference's own pairing: normals read with the position index (PathTracer.cpp:1191-1193); both files carry at least as
        # many vn as v, so the read stays inside the array (for suzanne.obj, 3,936 per-face normals for 2,012 positions, the pairs
        # are not the file's -- that is what the reference would shade with)
        v, i = po.mesh_from_obj_arrays(d["v"], d["idx_v"], d["vn"])
        t = np.arange(len(i))
        ids = np.where(t % 11 == 5, 3, t % 3).astype(np.int32)
        scene = po.Scene(v, i, sph, MONKEY_TABLE, ids)
        short = "monkey" if obj == "monkeyhead2" else obj
        out.append((f"{short}_materials_1080p_crop", scene, ref_cam(1920, 1080), 1920, 1080, po.VARIANT_MESH, po.NORMALS_VERTEX, [1, 4],
                    (1130, 700, 64, 64) if obj == "monkeyhead2" else (1270, 540, 64, 64)))
        if obj == "suzanne":
            # ... and the whole head small, with geometric normals (the defined rule for files whose normals do not pair up)
            vg, ig = po.mesh_from_obj_arrays(d["v"], d["idx_v"], None)
            out.append(("suzanne_materials_96", po.Scene(vg, ig, sph, MONKEY_TABLE, ids), ref_cam(96, 96), 96, 96, po.VARIANT_MESH,
                        po.NORMALS_GEOMETRIC, [1, 2], None))
    return out


def case_shading(name):
    return 2 if "materials_unit" in name else (1 if "materials" in name else 0)


def only(names):
    """Generates the named cases and leaves every other fixture as it is (python3 oracle/gen_golden.py --only a,b)."""
    path = f"{GOLD}/generation_report.json"
    report = json.load(open(path)) if os.path.exists(path) else []
    for (name, sc, cam, W, H, variant, nm, spp, crop) in config4_cases() + [tiny_materials_case(), unit_materials_case(), monkey_materials_case()] + config3_materials_cases():
        if name not in names:
            continue
        fx, rep = render_case(name, sc, cam, W, H, variant, nm, spp, crop, shading_mode=case_shading(name))
        np.savez_compressed(f"{GOLD}/case_{name}.npz", **fx)
        report = [r for r in report if r.get("case") != name] + [rep]
    json.dump(report, open(path, "w"), indent=1)


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--only":
        return only(sys.argv[2].split(","))
    os.makedirs(GOLD, exist_ok=True)
    os.makedirs(ASSETS_DIR, exist_ok=True)
    report = []

    # ---- OBJ assets: data files copied verbatim (they are inputs), plus what the reference's parser returns
    objs = {}
    for a in ASSETS:
        shutil.copyfile(f"{REF}/data/{a}.obj", f"{ASSETS_DIR}/{a}.obj")
        os.chmod(f"{ASSETS_DIR}/{a}.obj", 0o644)
        d, info = run_obj_dump(a)
        objs[a] = d
        np.savez_compressed(f"{GOLD}/obj_{a}.npz", **d)
        report.append({"case": f"obj_{a}", **info})
        print(json.dumps(report[-1]))
    for m in ["cornell_box.mtl", "cube.mtl", "monkeyhead2.mtl"]:
        shutil.copyfile(f"{REF}/data/{m}", f"{ASSETS_DIR}/{m}")
        os.chmod(f"{ASSETS_DIR}/{m}", 0o644)

    # ---- llvmpipe math + RNG known answers
    mf, rep = math_fixture()
    np.savez_compressed(f"{GOLD}/lp_math.npz", **mf)
    report.append(rep)

    sph, mat = po.reference_spheres_and_materials()
    np.savez(f"{GOLD}/world_tables_glibc.npz", spheres=sph, materials=mat)

    mv, mi = po.mesh_from_obj_arrays(objs["monkeyhead2"]["v"], objs["monkeyhead2"]["idx_v"], objs["monkeyhead2"]["vn"])
    monkey = po.Scene(mv, mi, sph, mat)
    np.savez_compressed(f"{GOLD}/lp_rng.npz", **rng_fixture(monkey))

    # ---- rendered cases
    ref_cam = lambda W, H: po.camera(60.0, W, H, (0, 0, 0), (0, 0, 1), (0, 1, 0))  # PathTracer.cpp:1160-1162
    cases = []
    cases.append(("monkey_mesh_128", monkey, ref_cam(128, 128), 128, 128, po.VARIANT_MESH, po.NORMALS_VERTEX, [1, 2, 16], None))
    cases.append(("monkey_mesh_160x90", monkey, ref_cam(160, 90), 160, 90, po.VARIANT_MESH, po.NORMALS_VERTEX, [1, 4], None))
    cases.append(("monkey_mesh_1080p_crop", monkey, ref_cam(1920, 1080), 1920, 1080, po.VARIANT_MESH, po.NORMALS_VERTEX, [1, 4],
                  (900, 560, 64, 64)))
    cases.append(("world_128", monkey, ref_cam(128, 128), 128, 128, po.VARIANT_WORLD, po.NORMALS_VERTEX, [1, 2, 16], None))
    cases.append(("world_1280x720_crop", monkey, ref_cam(1280, 720), 1280, 720, po.VARIANT_WORLD, po.NORMALS_VERTEX, [1, 8],
                  (600, 330, 96, 64)))
    cv, ci = po.mesh_from_obj_arrays(objs["cornell_box"]["v"], objs["cornell_box"]["idx_v"], None, translation=(0, 0, 0))
    cornell = po.Scene(cv, ci, sph, mat)
    ccam = lambda W, H: po.camera(39.3, W, H, (278, 273, -800), (278, 273, 0), (0, 1, 0))
    cases.append(("cornell_geo_128", cornell, ccam(128, 128), 128, 128, po.VARIANT_MESH, po.NORMALS_GEOMETRIC, [1, 2, 16], None))
    cases.append(("cornell_geo_1080p_crop", cornell, ccam(1920, 1080), 1920, 1080, po.VARIANT_MESH, po.NORMALS_GEOMETRIC, [1, 4],
                  (700, 300, 64, 64)))
    sv, si = po.mesh_from_obj_arrays(objs["suzanne"]["v"], objs["suzanne"]["idx_v"], None)
    suz = po.Scene(sv, si, sph, mat)
    cases.append(("suzanne_geo_96", suz, ref_cam(96, 96), 96, 96, po.VARIANT_MESH, po.NORMALS_GEOMETRIC, [1, 2], None))

    for (name, sc, cam, W, H, variant, nm, spp, crop) in cases:
        fx, rep = render_case(name, sc, cam, W, H, variant, nm, spp, crop)
        np.savez_compressed(f"{GOLD}/case_{name}.npz", **fx)
        report.append(rep)

    # ---- build-defined diffuse/emissive mode (edits/mesh_materials_geo.txt), MTL materials of the Cornell box and cube
    ctab = po.material_table_from_mtl(objs["cornell_box"]["mat_values"])
    cids = np.where(objs["cornell_box"]["mat_ids"] < 0, len(ctab) - 1, objs["cornell_box"]["mat_ids"])
    cornell_m = po.Scene(cv, ci, sph, ctab, cids)
    mcases = [("cornell_materials_128", cornell_m, ccam(128, 128), 128, 128, [1, 2, 16], None),
              ("cornell_materials_1080p_crop", cornell_m, ccam(1920, 1080), 1920, 1080, [1, 4], (700, 300, 64, 64))]
    for (name, sc, cam, W, H, spp, crop) in mcases:
        fx, rep = render_case(name, sc, cam, W, H, po.VARIANT_MESH, po.NORMALS_GEOMETRIC, spp, crop, shading_mode=1)
        np.savez_compressed(f"{GOLD}/case_{name}.npz", **fx)
        report.append(rep)

    for (name, sc, cam, W, H, variant, nm, spp, crop) in config4_cases() + [tiny_materials_case(), unit_materials_case(), monkey_materials_case()] + config3_materials_cases():
        fx, rep = render_case(name, sc, cam, W, H, variant, nm, spp, crop, shading_mode=case_shading(name))
        np.savez_compressed(f"{GOLD}/case_{name}.npz", **fx)
        report.append(rep)

    json.dump(report, open(f"{GOLD}/generation_report.json", "w"), indent=1)


if __name__ == "__main__":
    main()
