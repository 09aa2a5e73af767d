// The reference's render loop (PathTracer.cpp:1061-1524, the USE_GLSL branch) with the GL program dispatch
// replaced by calls into libpt_hip.so. Headless: there is no window, so the loop runs for a fixed number of passes
// instead of until ESC, and the present pass is an optional image file.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "Framebuffer.h"

struct PtContext;

namespace pthost {

struct Workload;

struct RenderOptions {
    std::string obj_path = "data/monkeyhead2.obj";  // PathTracer.cpp:1170
    std::string mtl_basedir;                         // "" = current directory, like the reference
    std::string shader_dir = "data/shaders";         // where pathtracing.fs would be; optional
    std::string workload;                            // "" = obj_path, set up as the reference's main() does; otherwise one of the named
                                                     // workloads of Workloads.h (cornell, monkeyhead2, suzanne, torus, world, instanced,
                                                     // torus_distorted) with its own scene, camera and -- unless width/height are set --
                                                     // size; its OBJ/MTL files are read from assets_dir
    std::string assets_dir = "assets";
    int width = 1280, height = 720;                  // PathTracer.cpp:1064 (with a workload: 0 = the workload's own size)
    int depth = 10;                                  // PathTracer.cpp:1376
    int passes = 16;                                 // frames to render
    int samples_per_dispatch = 1;                    // 1 = one dispatch per frame, exactly like the reference
    int device = 0;
    int program_kind = -1;                           // -1 = decide from the shader file / default; 0 world; 1 mesh; 2 = the reference's CPU
                                                     // tracer (Scene::color), running sums kept as Framebuffer::accumulationBuffer keeps them
    uint32_t seed = 18;                              // CPU tracer only: seed of its counter-based generator (no correspondence to the
                                                     // reference's srand values, PathTracer.cpp:1217, 1297)
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
    uint64_t rays = 0;          // closest-hit queries, one per bounce-loop iteration (PtStats::rays)
    uint64_t rays_traced = 0;   // of those, the ones evaluated (PtStats::rays_traced)
    uint64_t samples = 0;
    double seconds = 0.0;       // wall time of the passes (host clock around dispatch + finish, like the reference)
    double device_ms = 0.0;     // HIP-event time of the render kernels (several GPUs: the slowest one's)
    int triangles = 0;
    int width = 0, height = 0;  // the size rendered (a workload may bring its own)
    std::string exchange;       // several GPUs: how the images were merged ("rccl" or "host")
    std::string error;
};

// Runs the loop; fills `fb` (created inside, caller destroys) with the result in the reference's Framebuffer
// layout. Returns 0 on success.
int RunRenderLoop(const RenderOptions& opt, Framebuffer& fb, RenderResult& result);

// The same loop on several GPUs of one node, which the reference cannot do (one GL context, PathTracer.cpp:1301-1500): one
// context, one host thread and one stream pair per rank inside this process, the images merged over RCCL.
struct MultiGpuOptions {
    std::vector<int> devices;   // one rank per entry. A device may appear more than once: logical ranks sharing a GPU, for
                                // rehearsing the partition on a machine with fewer GPUs (RCCL cannot serve that: host exchange)
    enum Partition {
        SAMPLE,  // every rank renders its residue class of the sample indices (rank r: r + 1, r + 1 + N, ...) of the whole frame
                 // into a running SUM; ncclAllReduce(ncclFloat, ncclSum) of a copy of the sums, then x 1 / passes. Differs from
                 // the sequential running mean by rounding only.
        TILE     // every rank renders all samples of the image rows in its interleaved blocks of `block_rows` with the reference
                 // recurrence, so its pixels are bit-identical to one GPU's; ncclAllGather of the owned rows, no arithmetic.
    } partition = SAMPLE;
    enum Exchange {
        EXCHANGE_AUTO,  // RCCL when every rank has a device of its own, else HOST
        EXCHANGE_RCCL,
        EXCHANGE_HOST   // the same packing and unpacking on the devices, but the buffers meet in host memory (rehearsals; no RCCL needed)
    } exchange = EXCHANGE_AUTO;
    int block_rows = 16;
    bool exchange_every_frame = true;   // merge after every frame (what a host that presents every frame needs, PathTracer.cpp:1473-1498),
                                        // beside the rendering of the next frame; false: once, after the last frame
};
// opt.passes = samples per pixel in total; opt.samples_per_dispatch = samples a rank renders per frame. opt.device is ignored.
int RunRenderLoopMulti(const RenderOptions& opt, const MultiGpuOptions& multi, Framebuffer& fb, RenderResult& result);

// ---- pieces shared by the two loops -----------------------------------------------------------------------------------------
// The scene the options name: a workload of Workloads.h, or opt.obj_path set up as main() does (PathTracer.cpp:1160-1244).
// Fixes opt.width/height to the size that will be rendered.
bool PrepareWorkload(RenderOptions& opt, Workload& w, std::string* err);
// Program, uploads, modes and search structures on `ctx` (PathTracer.cpp:1248-1288). Makes ctx the calling thread's Shader
// context. *program <- the handle to pass to pt_use_program. Returns false with the message in *err.
bool SetupContext(PtContext* ctx, const RenderOptions& opt, const Workload& w, uint32_t* program, std::string* err);
// The per-frame uniform block (PathTracer.cpp:1372-1382) for `w`'s camera.
bool SetFrameUniforms(PtContext* ctx, const RenderOptions& opt, const Workload& w);

// Writes fb.colorBuffer as an uncompressed 32-bit TGA (18-byte header, image type 2), the format of the reference's
// one-shot output (PathTracer.cpp:1457-1471, libs/TGA.h).
bool WriteTga(const char* path, const Framebuffer& fb);

}  // namespace pthost
