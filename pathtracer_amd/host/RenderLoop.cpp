#include "RenderLoop.h"

#include <chrono>
#include <cstdio>
#include <vector>

#include "Scene.h"
#include "Shader.h"
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

int RunRenderLoop(const RenderOptions& opt, Framebuffer& fb, RenderResult& result) {
#define PT_CHECK(call)                              \
    do {                                            \
        if ((call) != PT_OK) {                      \
            result.error = pt_last_error();         \
            return 1;                               \
        }                                           \
    } while (0)

    const int width = opt.width, height = opt.height;
    // (Framebuffer keeps the reference's 16-bit width and height, Framebuffer.h:10-11)
    if (width < 1 || height < 1 || width > 65535 || height > 65535) {
        result.error = "image size " + std::to_string(width) + "x" + std::to_string(height) + " outside 1..65535 (Framebuffer holds 16-bit sizes)";
        return 1;
    }
    fb.create(width, height);

    // scene, as main() builds it (PathTracer.cpp:1160-1244)
    Scene scene;
    std::string err;
    if (!buildReferenceScene(scene, opt.obj_path.c_str(), opt.mtl_basedir.empty() ? nullptr : opt.mtl_basedir.c_str(), width,
                             height, &err)) {
        result.error = err;
        return 1;
    }
    std::vector<ptobj::material_t> materials;
    if (opt.translate_z != 3.f || opt.custom_camera || opt.shade_materials || opt.file_normals) {
        // non-reference framing or shading: rebuild mesh / camera with the requested values, keep the MTL data
        ptobj::attrib_t attrib;
        std::vector<ptobj::shape_t> shapes;
        ptobj::LoadObj(&attrib, &shapes, &materials, &err, opt.obj_path.c_str(),
                       opt.mtl_basedir.empty() ? nullptr : opt.mtl_basedir.c_str());
        scene.meshes[0] = opt.file_normals ? buildMeshIndexedNormals(attrib, shapes, 1.0f, float3(0.f, 0.f, opt.translate_z))
                                           : buildMesh(attrib, shapes, 1.0f, float3(0.f, 0.f, opt.translate_z));
        if (opt.custom_camera) {
            scene.camera.initialize(opt.fov, 0.1f, 100.f, (float)width, (float)height);
            scene.camera.lookAt(float3(opt.eye[0], opt.eye[1], opt.eye[2]), float3(opt.target[0], opt.target[1], opt.target[2]),
                                float3(opt.up[0], opt.up[1], opt.up[2]));
        }
    }
    const Mesh& mesh = scene.meshes[0];
    result.triangles = static_cast<int>(mesh.indices.size() / 3);

    // device context in place of the window + GL context (PathTracer.cpp:1072-1112)
    ContextGuard guard;
    PT_CHECK(pt_create(opt.device, &guard.ctx));
    PtContext* ctx = guard.ctx;
    Shader::SetContext(ctx);
    Shader::SetVersionString(450);
    if (opt.program_kind == 0) Shader::SetDefaultKind(Shader::STAGE_PATHTRACE_WORLD);
    if (opt.program_kind == 1) Shader::SetDefaultKind(Shader::STAGE_PATHTRACE_MESH);

    // accumulation targets (PathTracer.cpp:1136-1152)
    PT_CHECK(pt_target_create(ctx, width, height));

    // program (PathTracer.cpp:1248-1251)
    Shader pathTracingShader;
    const bool cpu_tracer = opt.program_kind == 2;   // the branch the reference compiles without USE_GLSL (PathTracer.cpp:1317-1368)
    uint32_t cpu_program = 0;
    if (cpu_tracer) {
        PT_CHECK(pt_program_create(ctx, PT_PROGRAM_CPU_TRACER, &cpu_program));
        PT_CHECK(pt_set_cpu_tracer(ctx, opt.seed, 100.f - 0.1f));   // srand (PathTracer.cpp:1216); far - near (:1161, 745)
        PT_CHECK(pt_set_accum_mode(ctx, PT_ACCUM_SUM));             // Framebuffer::accumulationBuffer starts at zero (Framebuffer.h:51-52)
        PT_CHECK(pt_target_clear(ctx));
    } else {
    pathTracingShader.loadVertexShader((opt.shader_dir + "/postprocess.vs").c_str());
    if (opt.program_kind >= 0) {
        // explicit choice: do not look at the shader file
        pathTracingShader.loadFragmentShader("pathtracing.fs");
    } else {
        pathTracingShader.loadFragmentShader((opt.shader_dir + "/pathtracing.fs").c_str());
    }
    pathTracingShader.createProgram(Shader::ATTRIBUTE_LESS);
    if (pathTracingShader.getProgram() == 0) {
        result.error = "could not create the path-tracing program";
        return 1;
    }
    }
    const uint32_t program = cpu_tracer ? cpu_program : pathTracingShader.getProgram();

    // uploads (PathTracer.cpp:1253-1288)
    PT_CHECK(pt_upload_spheres(ctx, &scene.spheres[0].center.x, (int32_t)scene.spheres.size()));
    PT_CHECK(pt_upload_materials(ctx, &scene.materials[0].r, (int32_t)scene.materials.size()));
    std::vector<uint32_t> indices(mesh.indices.begin(), mesh.indices.end());
    PT_CHECK(pt_upload_mesh(ctx, &mesh.vertices[0].position.x, (int32_t)mesh.vertices.size(), indices.data(), result.triangles));
    PT_CHECK(pt_set_normal_mode(ctx, mesh.has_vertex_normals ? PT_NORMALS_VERTEX : PT_NORMALS_GEOMETRIC));
    if (opt.shade_materials) {
        const std::vector<Material> table = buildMaterialTable(materials);
        const std::vector<int32_t> ids = resolveMaterialIds(mesh, table.size());
        PT_CHECK(pt_upload_materials(ctx, &table[0].r, (int32_t)table.size()));
        PT_CHECK(pt_upload_material_ids(ctx, ids.data(), (int32_t)ids.size()));
        PT_CHECK(pt_set_shading_mode(ctx, opt.lambert_unit_form ? PT_SHADE_MATERIALS_UNIT : PT_SHADE_MATERIALS));
    }
    if (opt.accel == 1) PT_CHECK(pt_build_accel(ctx));
    PT_CHECK(pt_enable_timing(ctx, 1));

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
        PtUniforms u;
        const Camera& cam = scene.camera;
        const float3* src[4] = {&cam.position, &cam.lowerLeftCorner, &cam.horizontal, &cam.vertical};
        float* dst[4] = {u.camera_position, u.camera_lower_left_corner, u.camera_horizontal, u.camera_vertical};
        for (int k = 0; k < 4; ++k) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
        u.depth = opt.depth;
        u.num_triangles = result.triangles;
        PT_CHECK(pt_set_uniforms(ctx, &u));
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
    if (cpu_tracer) PT_CHECK(pt_program_destroy(ctx, cpu_program));
    else pathTracingShader.destroyProgram();
    return 0;
#undef PT_CHECK
}

}  // namespace pthost
