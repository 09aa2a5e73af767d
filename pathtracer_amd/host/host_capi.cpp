// C exports of the host library for tests, benchmarks and other-language callers. These are conveniences over the
// C++ host classes (ObjLoader, Scene, Shader, RenderLoop); the reference-replacing device ABI is include/pt_hip.h.
#include <cstring>
#include <string>
#include <vector>

#include "ObjLoader.h"
#include "RenderLoop.h"
#include "Scene.h"
#include "Shader.h"
#include "Workloads.h"

#define PTH_API extern "C" __attribute__((visibility("default")))

struct PthObj {
    ptobj::attrib_t attrib;
    std::vector<ptobj::shape_t> shapes;
    std::vector<ptobj::material_t> materials;
    std::string err;
    size_t triangles = 0;
};

static void copy_text(char* dst, int cap, const std::string& s) {
    if (!dst || cap <= 0) return;
    const size_t n = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
    memcpy(dst, s.data(), n);
    dst[n] = '\0';
}

PTH_API PthObj* pth_obj_load(const char* path, const char* mtl_basedir, char* err, int err_cap) {
    PthObj* o = new PthObj();
    const bool ok = ptobj::LoadObj(&o->attrib, &o->shapes, &o->materials, &o->err, path, mtl_basedir);
    copy_text(err, err_cap, o->err);
    if (!ok) {
        delete o;
        return nullptr;
    }
    for (auto& s : o->shapes) o->triangles += s.mesh.indices.size() / 3;
    return o;
}

PTH_API void pth_obj_free(PthObj* o) { delete o; }

// counts: positions, normals, triangles, shapes, materials
PTH_API void pth_obj_counts(const PthObj* o, int* counts5) {
    counts5[0] = (int)(o->attrib.vertices.size() / 3);
    counts5[1] = (int)(o->attrib.normals.size() / 3);
    counts5[2] = (int)o->triangles;
    counts5[3] = (int)o->shapes.size();
    counts5[4] = (int)o->materials.size();
}

// Any destination may be NULL. Index arrays hold 3 entries per triangle, all shapes concatenated in shape order.
PTH_API void pth_obj_copy(const PthObj* o, float* v, float* vn, int* idx_v, int* idx_vn, int* mat_ids, int* shape_tris) {
    if (v && !o->attrib.vertices.empty()) memcpy(v, o->attrib.vertices.data(), o->attrib.vertices.size() * 4);
    if (vn && !o->attrib.normals.empty()) memcpy(vn, o->attrib.normals.data(), o->attrib.normals.size() * 4);
    size_t k = 0, t = 0, si = 0;
    for (auto& s : o->shapes) {
        for (auto& i : s.mesh.indices) {
            if (idx_v) idx_v[k] = i.vertex_index;
            if (idx_vn) idx_vn[k] = i.normal_index;
            ++k;
        }
        for (int m : s.mesh.material_ids) {
            if (mat_ids) mat_ids[t] = m;
            ++t;
        }
        if (shape_tris) shape_tris[si] = (int)(s.mesh.indices.size() / 3);
        ++si;
    }
}

PTH_API void pth_obj_shape_name(const PthObj* o, int i, char* name, int cap) { copy_text(name, cap, o->shapes[i].name); }

// values19: Ka Kd Ks Kt Ke (3 each), Ns, Ni, d, illum
PTH_API void pth_obj_material(const PthObj* o, int i, char* name, int cap, float* values19) {
    const ptobj::material_t& m = o->materials[i];
    copy_text(name, cap, m.name);
    const float* groups[5] = {m.ambient, m.diffuse, m.specular, m.transmittance, m.emission};
    for (int g = 0; g < 5; ++g)
        for (int c = 0; c < 3; ++c) values19[3 * g + c] = groups[g][c];
    values19[15] = m.shininess;
    values19[16] = m.ior;
    values19[17] = m.dissolve;
    values19[18] = (float)m.illum;
}

// (materials + 1) rows of (r, g, b, e): the table of the diffuse/emissive shading mode built from the MTL statements.
PTH_API void pth_obj_material_table(const PthObj* o, float* rows4) {
    const std::vector<pthost::Material> t = pthost::buildMaterialTable(o->materials);
    memcpy(rows4, t.data(), t.size() * sizeof(pthost::Material));
}

// Mesh as main() builds it: vertices6 = positions x {pos, normal}, indices = triangles x 3, material_ids per triangle.
// Returns 1 when the OBJ carries a normal for every position (vertex-normal mode usable), else 0.
PTH_API int pth_obj_build_mesh(const PthObj* o, float scale, float translate_z, float* vertices6, unsigned* indices, int* material_ids) {
    pthost::Mesh m = pthost::buildMesh(o->attrib, o->shapes, scale, pthost::float3(0.f, 0.f, translate_z));
    if (vertices6 && !m.vertices.empty()) memcpy(vertices6, m.vertices.data(), m.vertices.size() * sizeof(pthost::Vertex));
    if (indices)
        for (size_t i = 0; i < m.indices.size(); ++i) indices[i] = (unsigned)m.indices[i];
    if (material_ids && !m.material_ids.empty()) memcpy(material_ids, m.material_ids.data(), m.material_ids.size() * sizeof(int));
    return m.has_vertex_normals ? 1 : 0;
}

// The mesh with the normals the file assigns (normal_index): call with NULL arrays for the vertex count, then again to copy.
// Triangle order equals pth_obj_build_mesh's. Returns the vertex count; *has_normals = every corner carries a normal index.
PTH_API int pth_obj_build_mesh_indexed_normals(const PthObj* o, float scale, float translate_z, float* vertices6, unsigned* indices,
                                               int* has_normals) {
    pthost::Mesh m = pthost::buildMeshIndexedNormals(o->attrib, o->shapes, scale, pthost::float3(0.f, 0.f, translate_z));
    if (vertices6 && !m.vertices.empty()) memcpy(vertices6, m.vertices.data(), m.vertices.size() * sizeof(pthost::Vertex));
    if (indices)
        for (size_t i = 0; i < m.indices.size(); ++i) indices[i] = (unsigned)m.indices[i];
    if (has_normals) *has_normals = m.has_vertex_normals ? 1 : 0;
    return (int)m.vertices.size();
}

// out12 = position, lowerLeftCorner, horizontal, vertical
PTH_API void pth_camera(float fov, float width, float height, const float* eye, const float* target, const float* up, float* out12) {
    pthost::Camera c;
    c.initialize(fov, 0.1f, 100.f, width, height);
    c.lookAt(pthost::float3(eye[0], eye[1], eye[2]), pthost::float3(target[0], target[1], target[2]),
             pthost::float3(up[0], up[1], up[2]));
    const pthost::float3* v[4] = {&c.position, &c.lowerLeftCorner, &c.horizontal, &c.vertical};
    for (int k = 0; k < 4; ++k) { out12[3 * k] = v[k]->x; out12[3 * k + 1] = v[k]->y; out12[3 * k + 2] = v[k]->z; }
}

PTH_API void pth_reference_spheres(float* spheres40, float* materials40) {
    pthost::Scene s;
    pthost::buildReferenceSpheres(s);
    memcpy(spheres40, s.spheres.data(), 160);
    memcpy(materials40, s.materials.data(), 160);
}

// ---- named workloads (Workloads.h) ----------------------------------------------------------------------------------------
struct PthWorkload {
    pthost::Workload w;
};

PTH_API PthWorkload* pth_workload_build(const char* name, const char* assets_dir, int width, int height, char* err, int err_cap) {
    PthWorkload* h = new PthWorkload();
    std::string msg;
    if (!pthost::buildWorkload(name ? name : "", assets_dir ? assets_dir : "", width, height, h->w, &msg)) {
        copy_text(err, err_cap, msg.empty() ? std::string("cannot build workload") : msg);
        delete h;
        return nullptr;
    }
    copy_text(err, err_cap, "");
    return h;
}

PTH_API void pth_workload_free(PthWorkload* h) { delete h; }

// info7: program kind, width, height, vertices, triangles, material table rows, 1 = vertex normals usable
PTH_API void pth_workload_info(const PthWorkload* h, int* info7) {
    const pthost::Workload& w = h->w;
    info7[0] = w.program_kind; info7[1] = w.width; info7[2] = w.height;
    info7[3] = (int)w.mesh.vertices.size(); info7[4] = (int)(w.mesh.indices.size() / 3);
    info7[5] = (int)w.material_table.size(); info7[6] = w.mesh.has_vertex_normals ? 1 : 0;
}

// Any destination may be NULL. camera12 = position, lowerLeftCorner, horizontal, vertical.
PTH_API void pth_workload_copy(const PthWorkload* h, float* vertices6, unsigned* indices, int* material_ids, float* table4, float* camera12,
                               float* spheres40, float* materials40) {
    const pthost::Workload& w = h->w;
    if (vertices6 && !w.mesh.vertices.empty()) memcpy(vertices6, w.mesh.vertices.data(), w.mesh.vertices.size() * sizeof(pthost::Vertex));
    if (indices)
        for (size_t i = 0; i < w.mesh.indices.size(); ++i) indices[i] = (unsigned)w.mesh.indices[i];
    if (material_ids && !w.material_ids.empty()) memcpy(material_ids, w.material_ids.data(), w.material_ids.size() * sizeof(int32_t));
    if (table4 && !w.material_table.empty()) memcpy(table4, w.material_table.data(), w.material_table.size() * sizeof(pthost::Material));
    if (camera12) {
        const pthost::Camera& c = w.scene.camera;
        const pthost::float3* v[4] = {&c.position, &c.lowerLeftCorner, &c.horizontal, &c.vertical};
        for (int k = 0; k < 4; ++k) { camera12[3 * k] = v[k]->x; camera12[3 * k + 1] = v[k]->y; camera12[3 * k + 2] = v[k]->z; }
    }
    if (spheres40) memcpy(spheres40, w.scene.spheres.data(), 160);
    if (materials40) memcpy(materials40, w.scene.materials.data(), 160);
}

static pthost::Mesh mesh_from_arrays(const float* vertices6, int nv, const unsigned* indices, int nt) {
    pthost::Mesh m;
    m.vertices.resize((size_t)nv);
    if (nv > 0) memcpy(m.vertices.data(), vertices6, (size_t)nv * sizeof(pthost::Vertex));
    m.indices.resize((size_t)nt * 3);
    for (size_t i = 0; i < m.indices.size(); ++i) m.indices[i] = (int)indices[i];
    return m;
}

// instances copies of a mesh (pthost::instancedMesh): out_vertices6 holds instances * nv vertices, out_indices instances * nt * 3
PTH_API void pth_instanced_mesh(const float* vertices6, int nv, const unsigned* indices, int nt, int instances, int columns, double spacing,
                                double unit, unsigned seed, float* out_vertices6, unsigned* out_indices) {
    const pthost::Mesh m = pthost::instancedMesh(mesh_from_arrays(vertices6, nv, indices, nt), instances, columns, spacing, unit, seed);
    memcpy(out_vertices6, m.vertices.data(), m.vertices.size() * sizeof(pthost::Vertex));
    for (size_t i = 0; i < m.indices.size(); ++i) out_indices[i] = (unsigned)m.indices[i];
}

// the distorted copy of a mesh's vertices (pthost::distortedCopy); out_vertices6 holds nv vertices
PTH_API void pth_distorted_copy(const float* vertices6, int nv, double amplitude, const double* offset3, unsigned seed, float* out_vertices6) {
    const pthost::Mesh m = pthost::distortedCopy(mesh_from_arrays(vertices6, nv, nullptr, 0), amplitude, offset3, seed);
    memcpy(out_vertices6, m.vertices.data(), m.vertices.size() * sizeof(pthost::Vertex));
}

PTH_API float pth_parse_float(const char* text, double fallback) { return ptobj::ParseFloatText(text, text + strlen(text), fallback); }

// 0 none, 1 world, 2 mesh, 3 present
PTH_API int pth_shader_classify_text(const char* code) { return (int)Shader::ClassifySource(code); }

// Loads a fragment shader by file name through the Shader class and reports the stage it selected.
PTH_API int pth_shader_stage_for_file(const char* filename) {
    Shader s;
    s.loadFragmentShader(filename);
    return (int)s.getStage();
}

// Plain-C mirror of pthost::RenderOptions for callers that need more than pth_render offers.
struct PthRenderOptions {
    const char* obj_path;
    const char* mtl_basedir;   // may be NULL
    const char* output_tga;    // may be NULL
    int width, height, passes, samples_per_dispatch, program_kind, device, depth;
    int shade_materials;       // 1 = PT_SHADE_MATERIALS with the MTL-derived table
    int accel;                 // 1 = pt_build_accel (direction grid / LBVH), 0 = the shader's plain loop
    int custom_camera;         // 1 = use eye/target/up/fov below instead of the reference's camera
    float eye[3], target[3], up[3], fov;
    float translate_z;
};

static int run_loop(const pthost::RenderOptions& opt, float* rgb_sum, unsigned char* rgba8, double* stats3, char* err, int err_cap) {
    Framebuffer fb;
    pthost::RenderResult res;
    const int rc = pthost::RunRenderLoop(opt, fb, res);
    if (rc == 0) {
        const size_t n = (size_t)opt.width * opt.height;
        if (rgb_sum) memcpy(rgb_sum, fb.accumulationBuffer, n * 3 * sizeof(float));
        if (rgba8) memcpy(rgba8, fb.colorBuffer, n * 4);
        if (stats3) { stats3[0] = (double)res.rays; stats3[1] = (double)res.samples; stats3[2] = res.device_ms; }
    }
    copy_text(err, err_cap, res.error);
    fb.destroy();
    return rc;
}

PTH_API int pth_render_ex(const PthRenderOptions* o, float* rgb_sum, unsigned char* rgba8, double* stats3, char* err, int err_cap) {
    pthost::RenderOptions opt;
    opt.obj_path = o->obj_path;
    opt.mtl_basedir = o->mtl_basedir ? o->mtl_basedir : "";
    opt.output_tga = o->output_tga ? o->output_tga : "";
    opt.width = o->width; opt.height = o->height; opt.passes = o->passes;
    opt.samples_per_dispatch = o->samples_per_dispatch;
    opt.program_kind = o->program_kind;
    opt.device = o->device;
    opt.depth = o->depth;
    opt.shade_materials = o->shade_materials != 0;
    opt.accel = o->accel;
    opt.custom_camera = o->custom_camera != 0;
    for (int k = 0; k < 3; ++k) { opt.eye[k] = o->eye[k]; opt.target[k] = o->target[k]; opt.up[k] = o->up[k]; }
    opt.fov = o->fov;
    opt.translate_z = o->translate_z;
    opt.quiet = true;
    return run_loop(opt, rgb_sum, rgba8, stats3, err, err_cap);
}

// A named workload (Workloads.h) through the render loop, on one GPU (num_devices = 0: device 0) or on several (one rank per
// entry of `devices`, which may repeat a device; partition 0 = sample, 1 = tile; exchange 0 = auto, 1 = RCCL, 2 = host).
// shading 0 = reference, 1 = materials. rgb_sum / rgba8 / stats3 as for pth_render; stats3[2] = 1 when RCCL merged the images.
PTH_API int pth_render_workload(const char* workload, const char* assets_dir, int width, int height, int passes, int samples_per_dispatch,
                                int shading, const int* devices, int num_devices, int partition, int exchange, int block_rows, int every_frame,
                                float* rgb_sum, unsigned char* rgba8, double* stats3, char* err, int err_cap) {
    pthost::RenderOptions opt;
    opt.workload = workload ? workload : "";
    opt.assets_dir = assets_dir ? assets_dir : "assets";
    opt.width = width;
    opt.height = height;
    opt.passes = passes;
    opt.samples_per_dispatch = samples_per_dispatch;
    opt.shade_materials = shading != 0;
    opt.quiet = true;
    Framebuffer fb;
    pthost::RenderResult res;
    int rc;
    if (num_devices <= 0) {
        rc = pthost::RunRenderLoop(opt, fb, res);
    } else {
        pthost::MultiGpuOptions multi;
        multi.devices.assign(devices, devices + num_devices);
        multi.partition = partition ? pthost::MultiGpuOptions::TILE : pthost::MultiGpuOptions::SAMPLE;
        multi.exchange = exchange == 1 ? pthost::MultiGpuOptions::EXCHANGE_RCCL : (exchange == 2 ? pthost::MultiGpuOptions::EXCHANGE_HOST : pthost::MultiGpuOptions::EXCHANGE_AUTO);
        multi.block_rows = block_rows > 0 ? block_rows : 16;
        multi.exchange_every_frame = every_frame != 0;
        rc = pthost::RunRenderLoopMulti(opt, multi, fb, res);
    }
    if (rc == 0) {
        const size_t n = (size_t)res.width * res.height;
        if (rgb_sum) memcpy(rgb_sum, fb.accumulationBuffer, n * 3 * sizeof(float));
        if (rgba8) memcpy(rgba8, fb.colorBuffer, n * 4);
        if (stats3) { stats3[0] = (double)res.rays; stats3[1] = (double)res.samples; stats3[2] = res.exchange == "rccl" ? 1.0 : 0.0; }
    }
    copy_text(err, err_cap, res.error);
    fb.destroy();
    return rc;
}

// The whole render loop. rgb_sum (may be NULL) receives Framebuffer::accumulationBuffer (w*h*3), rgba8 (may be NULL)
// Framebuffer::colorBuffer (w*h*4). stats3 = rays, samples, device milliseconds. Returns 0 on success.
PTH_API int pth_render(const char* obj_path, const char* mtl_basedir, int width, int height, int passes, int samples_per_dispatch,
                       int program_kind, int device, const char* output_tga, float* rgb_sum, unsigned char* rgba8, double* stats3,
                       char* err, int err_cap) {
    pthost::RenderOptions opt;
    opt.obj_path = obj_path;
    opt.mtl_basedir = mtl_basedir ? mtl_basedir : "";
    opt.width = width;
    opt.height = height;
    opt.passes = passes;
    opt.samples_per_dispatch = samples_per_dispatch;
    opt.program_kind = program_kind;
    opt.device = device;
    opt.quiet = true;
    opt.output_tga = output_tga ? output_tga : "";
    return run_loop(opt, rgb_sum, rgba8, stats3, err, err_cap);
}
