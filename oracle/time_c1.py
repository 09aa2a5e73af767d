#!/usr/bin/env python3
"""Config C1 of the survey (cornell_box.obj, 512x512, 16 passes, depth 10, mesh variant with geometric normals): runs the
reference shader on llvmpipe here, times its passes the way the reference times frames, checks the C oracle against the
full image, and writes tests/golden/c1_cornell_512.json: per-pass llvmpipe timings and SHA-256 digests of the RGBA32F image
after 1, 4 and 16 passes, so that the GPU test can check a full-size render against the reference's own output without a
multi-megabyte fixture. With --c2: the benchmark configuration (1920x1080, 256 passes; several minutes of llvmpipe) ->
tests/golden/c2_cornell_1080p.json. With --materials: C1's size in the build-defined diffuse/emissive mode (the shader with
edits/mesh_materials_geo.txt, the MTL-derived table and per-triangle material ids) -> tests/golden/c1_cornell_512_materials.json.
Needs /root/reference (shader text) and oracle/_ref/lp_harness (make -C oracle ref)."""
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
import pt_oracle as po  # noqa: E402


def main():
    # --c2: the same scene at the benchmark configuration (1920x1080, 256 passes) -> tests/golden/c2_cornell_1080p.json
    c2 = "--c2" in sys.argv[1:]
    d, _ = gg.run_obj_dump("cornell_box")
    sph, mat = po.reference_spheres_and_materials()
    cv, ci = po.mesh_from_obj_arrays(d["v"], d["idx_v"], None, translation=(0, 0, 0))
    materials = "--materials" in sys.argv[1:]
    if materials:
        tab = po.material_table_from_mtl(d["mat_values"])
        scene = po.Scene(cv, ci, sph, tab, np.where(d["mat_ids"] < 0, len(tab) - 1, d["mat_ids"]))
    else:
        scene = po.Scene(cv, ci, sph, mat)
    W, H, passes, dump_after = (1920, 1080, 256, [1, 16, 64, 256]) if c2 else (512, 512, 16, [1, 4, 16])
    cam = po.camera(39.3, W, H, (278, 273, -800), (278, 273, 0), (0, 1, 0))
    tmp = tempfile.mkdtemp(prefix="c1_")
    job = dict(gg.scene_files(tmp, scene, materials), width=W, height=H, depth=10, camera=gg.cam_floats(cam), num_samples=passes,
               dump_after=dump_after, edits=f"{gg.EDITS}/{'mesh_materials_geo' if materials else 'mesh_geo'}.txt")
    img, dumps, info = gg.harness(job, tmp)
    shutil.rmtree(tmp)
    u = po.uniforms(cam, W, H, len(ci), 10, po.VARIANT_MESH, po.NORMALS_GEOMETRIC, po.SHADE_MATERIALS if materials else po.SHADE_REFERENCE)
    t0 = time.perf_counter()
    rgba, rays = po.render(scene, u, 1, passes, threads=8)
    oracle_s = time.perf_counter() - t0
    out = {"config": f"cornell_box.obj {W}x{H}, passes 1..{passes}, depth 10, IntersectMesh with geometric normals (fs:166), camera eye (278,273,-800) -> (278,273,0) vfov 39.3"
                     + (", diffuse/emissive mode (edits/mesh_materials_geo.txt) with the table of cornell_box.mtl" if materials else ""),
           "camera": [float(x) for x in cam],
           "llvmpipe": {"renderer": info.get("renderer", "llvmpipe"), "threads": os.cpu_count(), "pass_ms": info["pass_ms"],
                        "pass_ms_median": float(np.median(info["pass_ms"])), "rays": int(rays),
                        "Mrays_per_s": rays / (sum(info["pass_ms"]) * 1e-3) * 1e-6},
           "oracle_8_threads": {"seconds": oracle_s, "Mrays_per_s": rays / oracle_s * 1e-6,
                                f"bit_mismatch_vs_llvmpipe_after_{passes}": int((rgba.view(np.uint32) != dumps[passes].view(np.uint32)).sum())},
           "sha256": {str(k): hashlib.sha256(np.ascontiguousarray(v, np.float32).tobytes()).hexdigest() for k, v in dumps.items()}}
    name = ("c2_cornell_1080p" if c2 else "c1_cornell_512") + ("_materials" if materials else "")
    json.dump(out, open(os.path.join(gg.GOLD, name + ".json"), "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("llvmpipe", "oracle_8_threads")}, indent=1))


if __name__ == "__main__":
    main()
