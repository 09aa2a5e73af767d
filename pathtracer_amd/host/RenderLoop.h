// The reference's render loop (PathTracer.cpp:1061-1524, the USE_GLSL branch) with the GL program dispatch
// replaced by calls into libpt_hip.so. Headless: there is no window, so the loop runs for a fixed number of passes
// instead of until ESC, and the present pass is an optional image file.
#pragma once

#include <cstdint>
#include <string>

#include "Framebuffer.h"

namespace pthost {

struct RenderOptions {
    std::string obj_path = "data/monkeyhead2.obj";  // PathTracer.cpp:1170
    std::string mtl_basedir;                         // "" = current directory, like the reference
    std::string shader_dir = "data/shaders";         // where pathtracing.fs would be; optional
    int width = 1280, height = 720;                  // PathTracer.cpp:1064
    int depth = 10;                                  // PathTracer.cpp:1376
    int passes = 16;                                 // frames to render
    int samples_per_dispatch = 1;                    // 1 = one dispatch per frame, exactly like the reference
    int device = 0;
    int program_kind = -1;                           // -1 = decide from the shader file / default; 0 world; 1 mesh; 2 = the reference's CPU
                                                     // tracer (Scene::color), running sums kept as Framebuffer::accumulationBuffer keeps them
    uint32_t seed = 18;                              // CPU tracer only: the reference seeds with srand(2 ^ 17 - 1) = srand(18) (PathTracer.cpp:1216)
    bool quiet = false;                              // suppress the per-frame line
    std::string output_tga;                          // optional: write the display image when done
    // optional camera override (eye, target, fov); the reference hard-codes eye 0, target +z, fov 60
    bool custom_camera = false;
    float eye[3] = {0, 0, 0}, target[3] = {0, 0, 1}, up[3] = {0, 1, 0}, fov = 60.f;
    float translate_z = 3.f;                         // PathTracer.cpp:1179
    bool shade_materials = false;                    // true: PT_SHADE_MATERIALS with the table built from the MTL file
    bool lambert_unit_form = false;                  // with shade_materials: PT_SHADE_MATERIALS_UNIT, the bounce of pathtracing.fs:272
    bool file_normals = false;                       // true: normals through the OBJ's normal indices instead of the reference's pairing
    int accel = 1;                                   // 1 = pt_build_accel (direction grid / LBVH, same image); 0 = the shader's plain loop
};

struct RenderResult {
    uint64_t rays = 0;
    uint64_t samples = 0;
    double seconds = 0.0;       // wall time of the passes (host clock around dispatch + finish, like the reference)
    double device_ms = 0.0;     // HIP-event time of the render kernels
    int triangles = 0;
    std::string error;
};

// Runs the loop; fills `fb` (created inside, caller destroys) with the result in the reference's Framebuffer
// layout. Returns 0 on success.
int RunRenderLoop(const RenderOptions& opt, Framebuffer& fb, RenderResult& result);

// Writes fb.colorBuffer as an uncompressed 32-bit TGA (18-byte header, image type 2), the format of the reference's
// one-shot output (PathTracer.cpp:1457-1471, libs/TGA.h).
bool WriteTga(const char* path, const Framebuffer& fb);

}  // namespace pthost
