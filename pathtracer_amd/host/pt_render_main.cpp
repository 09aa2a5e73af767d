// Command-line front end of the render loop: the headless counterpart of the reference executable.
//   pt_render [--obj file.obj] [--mtl-dir dir/] [--size WxH] [--passes N] [--batch N] [--kind world|mesh|cpu]
//             [--device N] [--out image.tga] [--quiet] [--shading reference|materials|materials_unit] [--accel auto|loop]
//             [--translate-z Z] [--camera ex,ey,ez,tx,ty,tz,fov] [--file-normals]
//             [--workload cornell|monkeyhead2|suzanne|torus|world|instanced|torus_distorted] [--assets dir]
//             [--gpus N | --devices a,b,...] [--partition sample|tile] [--exchange auto|rccl|host] [--block-rows R] [--merge-at-end]
// --workload renders one of the benchmark configurations' scenes (its own camera, and its own size unless --size is given).
// --gpus N uses devices 0..N-1 of this node, one rank each, merged over RCCL; --devices may name a device more than once
// (logical ranks sharing a GPU, merged through host memory: a rehearsal of the partition).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <string>
#include <vector>

#include "RenderLoop.h"

int main(int argc, char** argv) {
    pthost::RenderOptions opt;
    pthost::MultiGpuOptions multi;
    bool size_given = false;
    for (int i = 1; i < argc; ++i) {
        auto next = [&](const char* flag) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", flag); exit(2); }
            return argv[++i];
        };
        if (!strcmp(argv[i], "--obj")) opt.obj_path = next("--obj");
        else if (!strcmp(argv[i], "--mtl-dir")) opt.mtl_basedir = next("--mtl-dir");
        else if (!strcmp(argv[i], "--shader-dir")) opt.shader_dir = next("--shader-dir");
        else if (!strcmp(argv[i], "--size")) { if (sscanf(next("--size"), "%dx%d", &opt.width, &opt.height) != 2) { fprintf(stderr, "--size WxH\n"); return 2; } size_given = true; }
        else if (!strcmp(argv[i], "--workload")) opt.workload = next("--workload");
        else if (!strcmp(argv[i], "--assets")) opt.assets_dir = next("--assets");
        else if (!strcmp(argv[i], "--gpus")) { const int n = atoi(next("--gpus")); multi.devices.clear(); for (int d = 0; d < n; ++d) multi.devices.push_back(d); }
        else if (!strcmp(argv[i], "--devices")) {
            multi.devices.clear();
            for (const char* p = next("--devices"); *p;) {
                char* end = nullptr;
                multi.devices.push_back((int)strtol(p, &end, 10));
                if (end == p) { fprintf(stderr, "--devices a,b,...\n"); return 2; }
                p = *end == ',' ? end + 1 : end;
            }
        }
        else if (!strcmp(argv[i], "--partition")) multi.partition = !strcmp(next("--partition"), "tile") ? pthost::MultiGpuOptions::TILE : pthost::MultiGpuOptions::SAMPLE;
        else if (!strcmp(argv[i], "--exchange")) { const char* e = next("--exchange"); multi.exchange = !strcmp(e, "rccl") ? pthost::MultiGpuOptions::EXCHANGE_RCCL : (!strcmp(e, "host") ? pthost::MultiGpuOptions::EXCHANGE_HOST : pthost::MultiGpuOptions::EXCHANGE_AUTO); }
        else if (!strcmp(argv[i], "--block-rows")) multi.block_rows = atoi(next("--block-rows"));
        else if (!strcmp(argv[i], "--merge-at-end")) multi.exchange_every_frame = false;
        else if (!strcmp(argv[i], "--passes")) opt.passes = atoi(next("--passes"));
        else if (!strcmp(argv[i], "--batch")) opt.samples_per_dispatch = atoi(next("--batch"));
        else if (!strcmp(argv[i], "--depth")) opt.depth = atoi(next("--depth"));
        else if (!strcmp(argv[i], "--device")) opt.device = atoi(next("--device"));
        else if (!strcmp(argv[i], "--kind")) { const char* k = next("--kind"); opt.program_kind = !strcmp(k, "mesh") ? 1 : (!strcmp(k, "cpu") ? 2 : 0); }
        else if (!strcmp(argv[i], "--shading")) { const char* m = next("--shading"); opt.lambert_unit_form = !strcmp(m, "materials_unit"); opt.shade_materials = opt.lambert_unit_form || !strcmp(m, "materials"); }
        else if (!strcmp(argv[i], "--accel")) opt.accel = !strcmp(next("--accel"), "loop") ? 0 : 1;
        else if (!strcmp(argv[i], "--file-normals")) opt.file_normals = true;
        else if (!strcmp(argv[i], "--translate-z")) opt.translate_z = (float)atof(next("--translate-z"));
        else if (!strcmp(argv[i], "--camera")) {
            // eye, target, vertical field of view in degrees
            if (sscanf(next("--camera"), "%f,%f,%f,%f,%f,%f,%f", &opt.eye[0], &opt.eye[1], &opt.eye[2], &opt.target[0], &opt.target[1],
                       &opt.target[2], &opt.fov) != 7) { fprintf(stderr, "--camera ex,ey,ez,tx,ty,tz,fov\n"); return 2; }
            opt.custom_camera = true;
        }
        else if (!strcmp(argv[i], "--out")) opt.output_tga = next("--out");
        else if (!strcmp(argv[i], "--quiet")) opt.quiet = true;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (!opt.workload.empty() && !size_given) opt.width = opt.height = 0;   // the workload's own size
    Framebuffer fb;
    pthost::RenderResult res;
    const int rc = multi.devices.empty() ? pthost::RunRenderLoop(opt, fb, res) : pthost::RunRenderLoopMulti(opt, multi, fb, res);
    if (rc != 0) {
        fprintf(stderr, "pt_render: %s\n", res.error.c_str());
    } else {
        printf("%d triangles, %llu samples, %llu rays, %.3f s wall, %.3f ms on device, %.1f Mrays/s\n", res.triangles,
               (unsigned long long)res.samples, (unsigned long long)res.rays, res.seconds, res.device_ms,
               res.rays / res.seconds * 1e-6);
    }
    fb.destroy();
    return rc;
}
