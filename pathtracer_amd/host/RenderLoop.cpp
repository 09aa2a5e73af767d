#include "RenderLoop.h"

#include <chrono>
#include <cstdio>
#include <vector>

#include "Scene.h"
#include "Shader.h"
#include "Workloads.h"
#include "pt_hip.h"

namespace pthost {

namespace {
struct ContextGuard {
    PtContext* ctx = nullptr;
    ~ContextGuard() {
        if (ctx) {
            Shader::SetContext(nullptr);
            pt_destroy(ctx);
        }
    }
};
}  // namespace

bool WriteTga(const char* path, const Framebuffer& fb) {
    unsigned char header[18] = {0};
    header[2] = 2;  // uncompressed true colour
    header[12] = static_cast<unsigned char>(fb.width & 0xff);
    header[13] = static_cast<unsigned char>(fb.width >> 8);
    header[14] = static_cast<unsigned char>(fb.height & 0xff);
    header[15] = static_cast<unsigned char>(fb.height >> 8);
    header[16] = 32;
    FILE* f = fopen(path, "wb");
    if (!f) return false;
    const size_t n = static_cast<size_t>(fb.width) * fb.height * 4;
    const bool ok = fwrite(header, sizeof header, 1, f) == 1 && fwrite(fb.colorBuffer, n, 1, f) == 1;
    fclose(f);
    return ok;
}

bool PrepareWorkload(RenderOptions& opt, Workload& w, std::string* err) {
    if (!opt.workload.empty()) {
        if (!buildWorkload(opt.workload, opt.assets_dir, opt.width, opt.height, w, err)) return false;
        opt.width = w.width;
        opt.height = w.height;
        if (opt.program_kind < 0) opt.program_kind = w.program_kind;
    } else {
        // scene, as main() builds it (PathTracer.cpp:1160-1244)
        w = Workload();
        w.name = opt.obj_path;
        w.width = opt.width;
        w.height = opt.height;
        if (opt.width < 1 || opt.height < 1) {
            if (err) *err = "image size " + std::to_string(opt.width) + "x" + std::to_string(opt.height) + " is not positive";
            return false;
        }
        const char* basedir = opt.mtl_basedir.empty() ? nullptr : opt.mtl_basedir.c_str();
        if (!buildReferenceScene(w.scene, opt.obj_path.c_str(), basedir, opt.width, opt.height, err)) return false;
        std::vector<ptobj::material_t> materials;
        if (opt.translate_z != 3.f || opt.custom_camera || opt.shade_materials || opt.file_normals) {
            // non-reference framing or shading: rebuild the mesh with the requested values, keep the MTL data
            ptobj::attrib_t attrib;
            std::vector<ptobj::shape_t> shapes;
            std::string warn;
            ptobj::LoadObj(&attrib, &shapes, &materials, &warn, opt.obj_path.c_str(), basedir);
            w.scene.meshes[0] = opt.file_normals ? buildMeshIndexedNormals(attrib, shapes, 1.0f, float3(0.f, 0.f, opt.translate_z))
                                                 : buildMesh(attrib, shapes, 1.0f, float3(0.f, 0.f, opt.translate_z));
        }
        w.mesh = w.scene.meshes[0];
        w.scene.meshes.clear();
        w.material_table = buildMaterialTable(materials);
        w.material_ids = resolveMaterialIds(w.mesh, w.material_table.size());
    }
    if (opt.custom_camera) {
        w.scene.camera.initialize(opt.fov, 0.1f, 100.f, (float)opt.width, (float)opt.height);
        w.scene.camera.lookAt(float3(opt.eye[0], opt.eye[1], opt.eye[2]), float3(opt.target[0], opt.target[1], opt.target[2]),
                              float3(opt.up[0], opt.up[1], opt.up[2]));
    }
    // (Framebuffer keeps the reference's 16-bit width and height, Framebuffer.h:10-11)
    if (opt.width < 1 || opt.height < 1 || opt.width > 65535 || opt.height > 65535) {
        if (err) *err = "image size " + std::to_string(opt.width) + "x" + std::to_string(opt.height) + " outside 1..65535 (Framebuffer holds 16-bit sizes)";
        return false;
    }
    return true;
}

bool SetupContext(PtContext* ctx, const RenderOptions& opt, const Workload& w, uint32_t* program, std::string* err) {
#define PT_SETUP(call)                          \
    do {                                        \
        if ((call) != PT_OK) {                  \
            if (err) *err = pt_last_error();    \
            return false;                       \
        }                                       \
    } while (0)
    Shader::SetContext(ctx);
    Shader::SetVersionString(450);
    if (opt.program_kind == 0) Shader::SetDefaultKind(Shader::STAGE_PATHTRACE_WORLD);
    if (opt.program_kind == 1) Shader::SetDefaultKind(Shader::STAGE_PATHTRACE_MESH);

    // accumulation targets (PathTracer.cpp:1136-1152)
    PT_SETUP(pt_target_create(ctx, opt.width, opt.height));

    // program (PathTracer.cpp:1248-1251)
    const bool cpu_tracer = opt.program_kind == 2;   // the branch the reference compiles without USE_GLSL (PathTracer.cpp:1317-1368)
    *program = 0;
    if (cpu_tracer) {
        PT_SETUP(pt_program_create(ctx, PT_PROGRAM_CPU_TRACER, program));
        PT_SETUP(pt_set_cpu_tracer(ctx, opt.seed, 100.f - 0.1f));   // far - near (PathTracer.cpp:1161, 745)
        PT_SETUP(pt_set_accum_mode(ctx, PT_ACCUM_SUM));             // Framebuffer::accumulationBuffer starts at zero (Framebuffer.h:51-52)
        PT_SETUP(pt_target_clear(ctx));
    } else {
        Shader pathTracingShader;
        pathTracingShader.loadVertexShader((opt.shader_dir + "/postprocess.vs").c_str());
        if (opt.program_kind >= 0) {
            // explicit choice: do not look at the shader file
            pathTracingShader.loadFragmentShader("pathtracing.fs");
        } else {
            pathTracingShader.loadFragmentShader((opt.shader_dir + "/pathtracing.fs").c_str());
        }
        pathTracingShader.createProgram(Shader::ATTRIBUTE_LESS);
        *program = pathTracingShader.getProgram();   // (the handle lives in the context: pt_program_destroy or pt_destroy releases it)
        if (*program == 0) {
            if (err) *err = "could not create the path-tracing program";
            return false;
        }
    }

    // uploads (PathTracer.cpp:1253-1288)
    const Mesh& mesh = w.mesh;
    const int32_t triangles = (int32_t)(mesh.indices.size() / 3);
    PT_SETUP(pt_upload_spheres(ctx, &w.scene.spheres[0].center.x, (int32_t)w.scene.spheres.size()));
    PT_SETUP(pt_upload_materials(ctx, &w.scene.materials[0].r, (int32_t)w.scene.materials.size()));
    std::vector<uint32_t> indices(mesh.indices.begin(), mesh.indices.end());
    PT_SETUP(pt_upload_mesh(ctx, &mesh.vertices[0].position.x, (int32_t)mesh.vertices.size(), indices.data(), triangles));
    PT_SETUP(pt_set_normal_mode(ctx, mesh.has_vertex_normals ? PT_NORMALS_VERTEX : PT_NORMALS_GEOMETRIC));
    if (opt.shade_materials) {
        PT_SETUP(pt_upload_materials(ctx, &w.material_table[0].r, (int32_t)w.material_table.size()));
        PT_SETUP(pt_upload_material_ids(ctx, w.material_ids.data(), (int32_t)w.material_ids.size()));
        PT_SETUP(pt_set_shading_mode(ctx, opt.lambert_unit_form ? PT_SHADE_MATERIALS_UNIT : PT_SHADE_MATERIALS));
    }
    if (opt.accel == 1) PT_SETUP(pt_build_accel(ctx));
    PT_SETUP(pt_enable_timing(ctx, 1));
    return true;
#undef PT_SETUP
}

bool SetFrameUniforms(PtContext* ctx, const RenderOptions& opt, const Workload& w) {
    PtUniforms u;
    const Camera& cam = w.scene.camera;
    const float3* src[4] = {&cam.position, &cam.lowerLeftCorner, &cam.horizontal, &cam.vertical};
    float* dst[4] = {u.camera_position, u.camera_lower_left_corner, u.camera_horizontal, u.camera_vertical};
    for (int k = 0; k < 4; ++k) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
    u.depth = opt.depth;
    u.num_triangles = (int32_t)(w.mesh.indices.size() / 3);
    return pt_set_uniforms(ctx, &u) == PT_OK;
}

int RunRenderLoop(const RenderOptions& options, Framebuffer& fb, RenderResult& result) {
#define PT_CHECK(call)                              \
    do {                                            \
        if ((call) != PT_OK) {                      \
            result.error = pt_last_error();         \
            return 1;                               \
        }                                           \
    } while (0)

    RenderOptions opt = options;
    Workload w;
    if (!PrepareWorkload(opt, w, &result.error)) return 1;
    const int width = opt.width, height = opt.height;
    result.width = width;
    result.height = height;
    result.triangles = static_cast<int>(w.mesh.indices.size() / 3);
    fb.create(width, height);

    // device context in place of the window + GL context (PathTracer.cpp:1072-1112)
    ContextGuard guard;
    PT_CHECK(pt_create(opt.device, &guard.ctx));
    PtContext* ctx = guard.ctx;
    uint32_t program = 0;
    if (!SetupContext(ctx, opt, w, &program, &result.error)) return 1;
    const bool cpu_tracer = opt.program_kind == 2;

    int samples = 0;
    const auto loop_start = std::chrono::steady_clock::now();
    while (samples < opt.passes) {
        const int batch = opt.samples_per_dispatch < opt.passes - samples ? opt.samples_per_dispatch : opt.passes - samples;
        const int first = samples + 1;
        samples += batch;
        PtStats before;
        PT_CHECK(pt_get_stats(ctx, &before));
        const auto t1 = std::chrono::steady_clock::now();

        // PathTracer.cpp:1370-1403
        PT_CHECK(pt_use_program(ctx, program));
        if (!SetFrameUniforms(ctx, opt, w)) {
            result.error = pt_last_error();
            return 1;
        }
        PT_CHECK(pt_dispatch(ctx, first, batch, 1));
        PT_CHECK(pt_finish(ctx));

        const auto t2 = std::chrono::steady_clock::now();
        const double seconds = std::chrono::duration<double>(t2 - t1).count();
        PtStats after;
        PT_CHECK(pt_get_stats(ctx, &after));
        const uint64_t totalRayCount = after.rays - before.rays;
        if (!opt.quiet) {
            // the reference's line (PathTracer.cpp:1419), with a real ray count
            printf("sample [%d] %.2fms (%.1f FPS) %.1fMrays/sec %.2fMRays/frame\n", samples, seconds * 1000.f, 1.0f / seconds,
                   totalRayCount / seconds * 1.0e-6f, totalRayCount * 1.0e-6f);
        }
    }
    result.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - loop_start).count();
    PtStats st;
    PT_CHECK(pt_get_stats(ctx, &st));
    result.rays = st.rays;
    result.rays_traced = st.rays_traced;
    result.samples = st.samples;
    result.device_ms = st.total_dispatch_ms;

    std::vector<float> rgba(static_cast<size_t>(width) * height * 4);
    PT_CHECK(pt_readback_rgba32f(ctx, rgba.data()));
    if (cpu_tracer) fb.storeSumsTileOrder(rgba.data(), samples);
    else fb.storeMean(rgba.data(), samples);
    if (!opt.output_tga.empty() && !WriteTga(opt.output_tga.c_str(), fb)) {
        result.error = "cannot write " + opt.output_tga;
        return 1;
    }
    PT_CHECK(pt_program_destroy(ctx, program));
    return 0;
#undef PT_CHECK
}

}  // namespace pthost
