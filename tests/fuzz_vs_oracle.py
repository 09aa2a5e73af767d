#!/usr/bin/env python3
"""Randomised differential test of the HIP path against the CPU oracle (test infrastructure; run by test_gpu_fuzz.py with a
few cases, by hand with many). Each case draws a scene (triangle soup with degenerate and duplicated triangles, a reference asset
or the sphere scene), a camera (outside, inside, near a vertex), an image size, a window, a bounce limit, a search mode, the
primary-reuse switch, and a sequence of one to three dispatches with their own sample counts (up to 96, so that the
probe-then-schedule path, split tiles and list reuse all occur) -- and compares primary ids, distances, the image bit for bit
and the ray count with the oracle (a pixel that is NaN in both counts as equal, whatever the NaN's sign and payload: the CPU's
and the GPU's default NaNs differ in the sign bit). Prints one JSON line per case and a summary; exit status 1 on any difference.

    python tests/fuzz_vs_oracle.py --cases 200 --seed 1
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def soup(rng, nt):
    centre = rng.uniform(-2.0, 2.0, (nt, 1, 3)) + np.array([0, 0, 5.0])
    size = rng.choice([0.15, 0.6, 2.5])
    tri = (centre + rng.uniform(-size, size, (nt, 3, 3))).astype(np.float32)
    k = max(1, nt // 12)
    tri[:k, 1] = tri[:k, 0]                                  # degenerate: two corners equal
    tri[k:2 * k] = tri[2 * k:3 * k]                          # exact duplicates (lower index must win)
    nrm = rng.normal(size=(nt * 3, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    verts = np.concatenate([tri.reshape(-1, 3), nrm], axis=1).astype(np.float32)
    return verts, np.arange(nt * 3, dtype=np.uint32).reshape(-1, 3)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--only", type=int, default=-1, help="run this case only (the others are drawn and skipped) and list the differing pixels")
    args = ap.parse_args(argv)
    import pt_oracle as po
    from pathtracer_amd import Tracer, capi
    from pathtracer_amd.scenes import SceneData
    rng = np.random.default_rng(args.seed)
    sph, mat = po.reference_spheres_and_materials()
    assets = {w: SceneData(w, 64, 64) for w in ("cornell", "monkeyhead2", "torus")}
    bad_cases, total_rays = 0, 0
    t = Tracer(0)
    for case in range(args.cases):
        kind = int(rng.random() < 0.85)                      # 1 = mesh program, 0 = sphere program
        pick = rng.choice(["soup", "soup", "cornell", "monkeyhead2", "torus"])
        if pick == "soup":
            verts, idx = soup(rng, int(rng.choice([1, 7, 40, 150, 400])))
            normal_mode = int(rng.integers(0, 2))
        else:
            verts, idx, normal_mode = assets[pick].vertices, assets[pick].indices, int(assets[pick].normal_mode)
        W, H = int(rng.integers(1, 150)), int(rng.integers(1, 110))
        if rng.random() < 0.04:   # now and then an image of thousands of tiles (the scheduler's other regime)
            W, H = int(rng.integers(300, 800)), int(rng.integers(200, 500))
        lo, hi = verts[:, :3].min(0), verts[:, :3].max(0)
        mid, ext = 0.5 * (lo + hi), float(np.linalg.norm(hi - lo)) + 1e-3
        where = rng.integers(0, 3)
        if where == 0:
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            eye = mid + d * ext * rng.uniform(0.7, 2.0)
        elif where == 1:
            eye = lo + rng.uniform(0.1, 0.9, 3) * (hi - lo)
        else:
            eye = verts[rng.integers(len(verts)), :3] + rng.normal(size=3) * 0.02 * ext
        target = mid + rng.normal(size=3) * 0.2 * ext
        cam = po.camera(float(rng.uniform(25, 90)), W, H, tuple(eye), tuple(target), (0, 1, 0))
        depth = int(rng.choice([0, 1, 2, 5, 10]))
        ntri = len(idx)
        accel = int(rng.choice([0, 0, 1, 2, 4]))
        if ntri <= 4 and accel in (2, 4):
            accel = 0
        if kind == 0:
            accel = 0
        reuse = bool(rng.random() < 0.7)
        mean = bool(rng.random() < 0.6)
        if rng.random() < 0.3:
            x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
            window = (x0, y0, int(rng.integers(1, W - x0 + 1)), int(rng.integers(1, H - y0 + 1)))
        else:
            window = None
        # a quarter of the mesh cases use the diffuse/emissive mode: random table of up to 10 rows (some emitters), one id per triangle
        materials_mode = bool(kind == 1 and rng.random() < 0.25)
        rows = int(rng.integers(1, 11))
        table = rng.uniform(0.05, 0.95, (rows, 4)).astype(np.float32)
        table[:, 3] = np.where(rng.random(rows) < 0.3, rng.uniform(0.5, 20.0, rows), 0.0).astype(np.float32)
        mat_ids = rng.integers(0, rows, len(idx)).astype(np.int32)
        if materials_mode and accel in (3, 4):
            accel = 0
        # now and then only the row blocks of one of several renderers (image-tile partition)
        interleave = None
        if rng.random() < 0.15:
            cnt = int(rng.integers(2, 6))
            interleave = (int(rng.choice([1, 3, 8, 16])), int(rng.integers(0, cnt)), cnt)
        first0 = int(rng.integers(1, 50))
        stride = 1 if mean else int(rng.integers(1, 9))
        counts = [int(rng.choice([1, 2, 7, 33, 96])) for _ in range(int(rng.integers(1, 4)))]
        builder = int(rng.choice([capi.BVH_BUILDER_BINNED, capi.BVH_BUILDER_BINNED, capi.BVH_BUILDER_PLOC, capi.BVH_BUILDER_MORTON, capi.BVH_BUILDER_SAH]))
        if args.only >= 0 and case != args.only:
            continue
        prog = t.program_create(kind)
        t.use_program(prog)
        t.upload_spheres(sph)
        t.upload_materials(table if materials_mode else mat)
        t.upload_mesh(verts, idx)
        if materials_mode:
            t.upload_material_ids(mat_ids)
        t.set_normal_mode(normal_mode)
        t.set_shading_mode(1 if materials_mode else 0)
        t.set_accum_mode(capi.ACCUM_MEAN if mean else capi.ACCUM_SUM)
        t.set_primary_reuse(reuse)
        t.target_create(W, H)
        t.set_uniforms(cam, ntri, depth)
        if kind == 1:
            t.set_bvh_builder(builder)
            t.build_accel()
        t.set_accel_mode(accel)
        if window:
            t.set_window(*window)
        if interleave:
            t.set_row_interleave(*interleave)
        wx0, wy0, ww, wh = window if window else (0, 0, W, H)
        if interleave:   # the oracle renders the owned row blocks one by one, as windows
            blk, idx_r, cnt = interleave
            owned_rows = [y for y in range(wy0, wy0 + wh) if (y // blk) % cnt == idx_r]
            runs, start = [], None
            for y in range(wy0, wy0 + wh + 1):
                mine = y < wy0 + wh and (y // blk) % cnt == idx_r
                if mine and start is None:
                    start = y
                if not mine and start is not None:
                    runs.append((wx0, start, ww, y - start)); start = None
            oracle_windows = runs
        else:
            oracle_windows = [window]
        scene = po.Scene(verts, idx, sph, table if materials_mode else mat, mat_ids if materials_mode else None)
        u = po.uniforms(cam, W, H, ntri, depth, kind, normal_mode, 1 if materials_mode else 0)
        ids, ts = t.readback_primary()
        oids, ots, _ = po.primary(scene, u, window=window, threads=args.threads)
        t.reset_stats()
        ref = None if mean else np.zeros((H, W, 4), np.float32)
        first, oracle_rays = first0, 0
        plan = []
        for n in counts:
            plan.append((first, n))
            t.dispatch(first, n, stride)
            if mean and ref is None:
                ref = np.zeros((H, W, 4), np.float32)
                ref[..., 3] = 1.0
            for ow in oracle_windows:
                if mean:
                    ref, r = po.render(scene, u, first, n, ref, window=ow, threads=args.threads)
                else:
                    ref, r = po.render_sum(scene, u, first, stride, n, ref, window=ow, threads=args.threads)
                oracle_rays += r
            first += n * stride
        t.finish()
        img = t.readback()
        st = t.stats()
        if window:
            x0, y0, w, h = window
            sl = (slice(y0, y0 + h), slice(x0, x0 + w))
        else:
            sl = (slice(None), slice(None))
        # alpha: 1 in the mean mode, the number of samples so far in the sum mode -- compared through the colour channels only
        a, b = img[sl][..., :3], ref[sl][..., :3]
        both_nan = np.isnan(a) & np.isnan(b)   # x86 and gfx950 generate default NaNs of opposite sign: a NaN is a NaN
        diff = int(((a.view(np.uint32) != b.view(np.uint32)) & ~both_nan).sum())
        rowmask = np.ones(H, bool)
        if interleave:   # the primary read-back fills in the renderer's own rows only
            rowmask[:] = False
            rowmask[owned_rows] = True
        rm = rowmask[sl[0]]
        id_diff = int((ids[sl][rm] != oids[sl][rm]).sum())
        t_diff = int((ts[sl][rm].view(np.uint32) != ots[sl][rm].view(np.uint32)).sum())
        ray_diff = int(st["rays"]) - int(oracle_rays)
        ok = diff == 0 and id_diff == 0 and t_diff == 0 and ray_diff == 0
        bad_cases += 0 if ok else 1
        total_rays += int(st["rays"])
        print(json.dumps({"case": case, "scene": str(pick) if kind else "spheres", "triangles": ntri, "size": [W, H], "window": window,
                          "interleave": interleave, "depth": depth, "shading": "materials" if materials_mode else "reference", "accel": accel, "builder": builder, "reuse": reuse, "mean": mean, "stride": stride, "dispatches": plan,
                          "differing_words": diff, "nan_words": int(both_nan.sum()), "differing_ids": id_diff, "differing_t": t_diff, "ray_count_difference": ray_diff}),
              flush=True)
        if args.only >= 0:
            bad = np.argwhere(((a.view(np.uint32) != b.view(np.uint32)) & ~both_nan).any(-1))
            for y, x in bad[:16]:
                print(json.dumps({"pixel": [int(x), int(y)], "hip": [float(v) for v in img[sl][y, x, :3]], "oracle": [float(v) for v in ref[sl][y, x, :3]],
                                  "primary_id": int(ids[sl][y, x])}))
        if window:
            t.set_window(0, 0, 0, 0)
        if interleave:
            t.set_row_interleave(16, 0, 1)
        t.set_accel_mode(0)
        t.set_shading_mode(0)
        t.program_destroy(prog)
    t.close()
    print(json.dumps({"cases": args.cases, "seed": args.seed, "cases_with_differences": bad_cases, "rays_compared": total_rays}))
    return 1 if bad_cases else 0


if __name__ == "__main__":
    sys.exit(main())
