#include "Workloads.h"

#include <cmath>

namespace pthost {

std::vector<double> lcgUniform(size_t count, uint32_t seed) {
    std::vector<double> out(count);
    uint32_t state = seed;
    for (size_t i = 0; i < count; ++i) {
        state = 1664525u * state + 1013904223u;
        out[i] = (double)(state >> 8) / 16777216.0;
    }
    return out;
}

Mesh instancedMesh(const Mesh& base, int instances, int columns, double spacing, double unit, uint32_t seed) {
    Mesh m;
    m.has_vertex_normals = false;
    m.bounds_min = float3(3.0e38f);
    m.bounds_max = float3(-3.0e38f);
    const size_t nv = base.vertices.size(), nt = base.indices.size() / 3;
    const std::vector<double> rnd = lcgUniform((size_t)instances * 4, seed);
    const int rows = (instances + columns - 1) / columns;
    m.vertices.resize((size_t)instances * nv);
    m.indices.resize((size_t)instances * nt * 3);
    m.material_ids.assign((size_t)instances * nt, -1);
    for (int k = 0; k < instances; ++k) {
        const double* r = &rnd[4 * (size_t)k];
        const double cx = ((double)(k % columns) - (double)(columns - 1) / 2.0) * spacing + (r[0] - 0.5) * 0.8;
        const double cy = ((double)(k / columns) - (double)(rows - 1) / 2.0) * spacing + (r[1] - 0.5) * 0.8;
        const double cz = (r[2] - 0.5) * 6.0;
        const double yaw = (r[3] - 0.5) * 1.0;
        const double c = cos(yaw), s = sin(yaw);
        for (size_t i = 0; i < nv; ++i) {
            const double vx = base.vertices[i].position.x, vy = base.vertices[i].position.y, vz = base.vertices[i].position.z;
            const double x = vx * c + vz * s;
            const double z = -vx * s + vz * c;
            Vertex& out = m.vertices[(size_t)k * nv + i];
            out.position = float3((float)((x + cx) * unit), (float)((vy + cy) * unit), (float)((z + cz) * unit));
            m.bounds_min = float3(std::fmin(m.bounds_min.x, out.position.x), std::fmin(m.bounds_min.y, out.position.y), std::fmin(m.bounds_min.z, out.position.z));
            m.bounds_max = float3(std::fmax(m.bounds_max.x, out.position.x), std::fmax(m.bounds_max.y, out.position.y), std::fmax(m.bounds_max.z, out.position.z));
        }
        for (size_t j = 0; j < nt * 3; ++j) m.indices[(size_t)k * nt * 3 + j] = base.indices[j] + (int)((size_t)k * nv);
    }
    return m;
}

Mesh distortedCopy(const Mesh& base, double amplitude, const double offset[3], uint32_t seed) {
    Mesh m;
    m.has_vertex_normals = false;
    m.indices = base.indices;
    m.material_ids = base.material_ids;
    const size_t nv = base.vertices.size();
    const std::vector<double> rnd = lcgUniform(nv * 3, seed);
    m.vertices.resize(nv);
    m.bounds_min = float3(3.0e38f);
    m.bounds_max = float3(-3.0e38f);
    for (size_t i = 0; i < nv; ++i) {
        const float3& p = base.vertices[i].position;
        const double in[3] = {p.x, p.y, p.z};
        float out[3];
        for (int a = 0; a < 3; ++a) out[a] = (float)((in[a] + (rnd[3 * i + a] - 0.5) * (2.0 * amplitude)) + offset[a]);
        m.vertices[i].position = float3(out[0], out[1], out[2]);
        m.bounds_min = float3(std::fmin(m.bounds_min.x, out[0]), std::fmin(m.bounds_min.y, out[1]), std::fmin(m.bounds_min.z, out[2]));
        m.bounds_max = float3(std::fmax(m.bounds_max.x, out[0]), std::fmax(m.bounds_max.y, out[1]), std::fmax(m.bounds_max.z, out[2]));
    }
    return m;
}

Mesh torusWithDistortedCopy(const Mesh& base, double separation) {
    const double offset[3] = {separation, 0.0, 0.0};
    const Mesh right = distortedCopy(base, 0.12, offset);
    Mesh m;
    m.has_vertex_normals = false;
    const size_t nv = base.vertices.size();
    m.vertices.reserve(2 * nv);
    for (const Vertex& v : base.vertices) {
        Vertex left;   // (normals stay zero: the file has none)
        left.position = float3((float)((double)v.position.x - separation), v.position.y, v.position.z);
        m.vertices.push_back(left);
    }
    m.vertices.insert(m.vertices.end(), right.vertices.begin(), right.vertices.end());
    m.indices = base.indices;
    for (int i : base.indices) m.indices.push_back(i + (int)nv);
    m.material_ids.assign(m.indices.size() / 3, -1);
    m.bounds_min = float3(3.0e38f);
    m.bounds_max = float3(-3.0e38f);
    for (const Vertex& v : m.vertices) {
        m.bounds_min = float3(std::fmin(m.bounds_min.x, v.position.x), std::fmin(m.bounds_min.y, v.position.y), std::fmin(m.bounds_min.z, v.position.z));
        m.bounds_max = float3(std::fmax(m.bounds_max.x, v.position.x), std::fmax(m.bounds_max.y, v.position.y), std::fmax(m.bounds_max.z, v.position.z));
    }
    return m;
}

namespace {

struct View {
    float fov;
    float3 eye, target, up;
    float translate_z;
};
// The reference hard-codes eye (0,0,0) looking down +z with a 60 degree vertical field of view and moves the mesh to z + 3
// (PathTracer.cpp:1160-1162, 1178-1183). That frames the monkey heads; the Cornell box (0..556 units) needs a camera of its
// own, which is this build's choice: the classic Cornell view.
const View kReferenceView = {60.0f, float3(0.f, 0.f, 0.f), float3(0.f, 0.f, 1.f), float3(0.f, 1.f, 0.f), 3.0f};
const View kCornellView = {39.3f, float3(278.f, 273.f, -800.f), float3(278.f, 273.f, 0.f), float3(0.f, 1.f, 0.f), 0.0f};

void setCamera(Workload& w, const View& v) {
    w.scene.camera.initialize(v.fov, 0.1f, 100.f, (float)w.width, (float)w.height);
    w.scene.camera.lookAt(v.eye, v.target, v.up);
}

bool loadMesh(const std::string& path, const std::string& basedir, float translate_z, Mesh& mesh, std::vector<ptobj::material_t>& mtl, std::string* err) {
    ptobj::attrib_t attrib;
    std::vector<ptobj::shape_t> shapes;
    if (!ptobj::LoadObj(&attrib, &shapes, &mtl, err, path.c_str(), basedir.c_str())) return false;
    mesh = buildMesh(attrib, shapes, 1.0f, float3(0.f, 0.f, translate_z));
    return true;
}

Material row(float r, float g, float b, float e) {
    Material m;
    m.r = r; m.g = g; m.b = b; m.type = e;
    return m;
}

}  // namespace

bool buildWorkload(const std::string& name, const std::string& assets_dir, int width, int height, Workload& w, std::string* err) {
    const std::string dir = assets_dir.empty() || assets_dir.back() == '/' ? assets_dir : assets_dir + "/";
    w = Workload();
    w.name = name;
    buildReferenceSpheres(w.scene);
    std::vector<ptobj::material_t> mtl;
    auto size = [&](int w0, int h0) {
        w.width = width > 0 ? width : w0;
        w.height = height > 0 ? height : h0;
    };
    if (name == "instanced") {
        // ~1M triangles: 254 copies of suzanne.obj (3,936 triangles each = 999,744), geometric normals, the camera on the grid's
        // axis far enough to see all of it; one colour per instance, every eighth instance a lamp
        size(3840, 2160);
        Mesh base;
        if (!loadMesh(dir + "suzanne.obj", dir, 0.0f, base, mtl, err)) return false;
        const double unit = 30.0;
        w.mesh = instancedMesh(base, 254, 16, 2.8, unit);
        const size_t per = base.indices.size() / 3;
        w.material_ids.resize(w.mesh.indices.size() / 3);
        for (size_t t = 0; t < w.material_ids.size(); ++t) w.material_ids[t] = (int32_t)((t / per) % 8);
        w.material_table = {row(0.8f, 0.8f, 0.8f, 0.f), row(0.8f, 0.3f, 0.3f, 0.f), row(0.3f, 0.8f, 0.3f, 0.f), row(0.3f, 0.3f, 0.8f, 0.f),
                            row(0.8f, 0.8f, 0.3f, 0.f), row(0.3f, 0.8f, 0.8f, 0.f), row(0.8f, 0.3f, 0.8f, 0.f), row(1.0f, 0.9f, 0.7f, 6.0f)};
        const View v = {60.0f, float3(0.f, 0.f, (float)(-44.0 * unit)), float3(0.f, 0.f, 0.f), float3(0.f, 1.f, 0.f), 0.0f};
        setCamera(w, v);
        return true;
    }
    if (name == "torus_distorted") {
        // the benchmark's fourth configuration: catmark_torus_creases0.obj next to a distorted copy of itself, the reference's
        // camera and its z + 3 translation, geometric normals (the file has no vn)
        size(3840, 2160);
        Mesh base;
        if (!loadMesh(dir + "catmark_torus_creases0.obj", dir, 3.0f, base, mtl, err)) return false;
        w.mesh = torusWithDistortedCopy(base);
        const size_t per = base.indices.size() / 3;
        w.material_ids.resize(w.mesh.indices.size() / 3);
        for (size_t t = 0; t < w.material_ids.size(); ++t) w.material_ids[t] = t >= per ? 1 : 0;
        w.material_table = {row(0.75f, 0.75f, 0.75f, 0.f), row(0.3f, 0.6f, 0.8f, 0.f)};
        setCamera(w, kReferenceView);
        return true;
    }
    struct Entry { const char* name; const char* obj; const View* view; int program, w0, h0; };
    static const Entry table[] = {
        {"cornell", "cornell_box.obj", &kCornellView, 1, 1920, 1080},
        {"monkeyhead2", "monkeyhead2.obj", &kReferenceView, 1, 1920, 1080},
        {"suzanne", "suzanne.obj", &kReferenceView, 1, 1920, 1080},
        {"torus", "catmark_torus_creases0.obj", &kReferenceView, 1, 3840, 2160},
        {"world", "monkeyhead2.obj", &kReferenceView, 0, 1280, 720},
    };
    for (const Entry& e : table) {
        if (name != e.name) continue;
        size(e.w0, e.h0);
        w.program_kind = e.program;
        if (!loadMesh(dir + e.obj, dir, e.view->translate_z, w.mesh, mtl, err)) return false;
        const std::vector<Material> t = buildMaterialTable(mtl);
        w.material_table = t;
        w.material_ids = resolveMaterialIds(w.mesh, t.size());
        setCamera(w, *e.view);
        return true;
    }
    if (err) *err = "unknown workload '" + name + "' (cornell, monkeyhead2, suzanne, torus, world, instanced, torus_distorted)";
    return false;
}

}  // namespace pthost
