#include "Scene.h"

#include <algorithm>
#include <map>
#include <utility>

#include <cfloat>
#include <cmath>
#include <cstdlib>

namespace pthost {

float length(const float3& v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
float3 normalize(const float3& v) { return v * (1.f / length(v)); }
float3 cross(const float3& a, const float3& b) {
    return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

void Camera::initialize(float fov_, float near_, float far_, float width_, float height_) {
    width = width_;
    height = height_;
    fov = fov_;
    nearPlane = near_;
    farPlane = far_;
    aspectRatio = width / height;
}

void Camera::lookAt(const float3& eye, const float3& target, const float3& up) {
    position = eye;
    const float3 w = normalize(target - eye);
    const float3 u = normalize(cross(up, w));
    const float3 v = cross(w, u);
    // half-height of the image plane at distance 1; the angle is evaluated in double, as the reference's
    // tan(fov * M_PI / 360.0f) is
    const float theta = static_cast<float>(tan(static_cast<double>(fov) * M_PI / static_cast<double>(360.0f)));
    const float hh = theta;
    const float hw = theta * aspectRatio;
    const float focusDist = 1.f;
    lowerLeftCorner = position - float3(hw) * u * focusDist - float3(hh) * v * focusDist + w * focusDist;
    horizontal = float3(2.f * hw * focusDist) * u;
    vertical = float3(2.f * hh * focusDist) * v;
}

Mesh buildMesh(const ptobj::attrib_t& attrib, const std::vector<ptobj::shape_t>& shapes, float scale, const float3& translation) {
    Mesh m;
    m.bounds_min = float3(FLT_MAX);
    m.bounds_max = float3(-FLT_MAX);
    const size_t nfloats = attrib.vertices.size();
    m.has_vertex_normals = attrib.normals.size() >= nfloats && nfloats > 0;
    m.vertices.reserve(nfloats / 3);
    for (size_t i = 0; i + 2 < nfloats; i += 3) {
        Vertex vx;
        vx.position = float3(attrib.vertices[i] * scale, attrib.vertices[i + 1] * scale, attrib.vertices[i + 2] * scale + translation.z);
        if (m.has_vertex_normals) vx.normal = float3(attrib.normals[i], attrib.normals[i + 1], attrib.normals[i + 2]);
        const float3& p = vx.position;
        if (p.x < m.bounds_min.x) m.bounds_min.x = p.x;
        if (p.y < m.bounds_min.y) m.bounds_min.y = p.y;
        if (p.z < m.bounds_min.z) m.bounds_min.z = p.z;
        if (p.x > m.bounds_max.x) m.bounds_max.x = p.x;
        if (p.y > m.bounds_max.y) m.bounds_max.y = p.y;
        if (p.z > m.bounds_max.z) m.bounds_max.z = p.z;
        m.vertices.push_back(vx);
    }
    for (const ptobj::shape_t& s : shapes) {
        const size_t n = s.mesh.indices.size();
        for (size_t k = 0; k + 2 < n; k += 3) {
            m.indices.push_back(s.mesh.indices[k].vertex_index);
            m.indices.push_back(s.mesh.indices[k + 1].vertex_index);
            m.indices.push_back(s.mesh.indices[k + 2].vertex_index);
            m.material_ids.push_back(k / 3 < s.mesh.material_ids.size() ? s.mesh.material_ids[k / 3] : -1);
        }
    }
    return m;
}

float nextFloat(float min, float max) { return min + ((float)rand() / (float)RAND_MAX) * (max - min); }

void buildReferenceSpheres(Scene& scene) {
    srand(2 ^ 17 - 1);  // sic: XOR, equals srand(18)
    scene.spheres.clear();
    scene.spheres.reserve(10);
    for (int i = 0; i < 10; i++) {
        Sphere s;
        s.center.x = nextFloat(-3.f, +3.f);
        s.center.y = nextFloat(-3.f, +3.f);
        s.center.z = nextFloat(0.5f, +10.f);
        s.radius = nextFloat(0.1f, 1.f);
        scene.spheres.push_back(s);
    }
    scene.materials.clear();
    scene.materials.reserve(10);
    for (int i = 0; i < 10; i++) {
        int id = Material::LAMBERT;
        if (i == 7) id = Material::DIELECTRIC;
        else if (i > 7) id = Material::METAL;
        Material mt;
        mt.r = nextFloat(0.0f, 1.f);
        mt.g = nextFloat(0.0f, 1.f);
        mt.b = nextFloat(0.f, 1.f);
        mt.type = (float)id;
        scene.materials.push_back(mt);
    }
}

Mesh buildMeshIndexedNormals(const ptobj::attrib_t& attrib, const std::vector<ptobj::shape_t>& shapes, float scale,
                             const float3& translation) {
    Mesh m;
    m.bounds_min = float3(3.0e38f);
    m.bounds_max = float3(-3.0e38f);
    m.has_vertex_normals = true;
    std::map<std::pair<int, int>, int> seen;
    const size_t num_normals = attrib.normals.size() / 3, num_positions = attrib.vertices.size() / 3;
    for (const ptobj::shape_t& s : shapes) {
        for (size_t k = 0; k < s.mesh.indices.size(); ++k) {
            const ptobj::index_t& ix = s.mesh.indices[k];
            const bool has_normal = ix.normal_index >= 0 && (size_t)ix.normal_index < num_normals;
            if (!has_normal) m.has_vertex_normals = false;
            const std::pair<int, int> key(ix.vertex_index, has_normal ? ix.normal_index : -1);
            auto it = seen.find(key);
            if (it == seen.end()) {
                Vertex v;
                if (ix.vertex_index >= 0 && (size_t)ix.vertex_index < num_positions) {
                    const float* p = &attrib.vertices[3 * (size_t)ix.vertex_index];
                    v.position = float3(p[0] * scale, p[1] * scale, p[2] * scale + translation.z);  // same rule as buildMesh
                }
                if (has_normal) {
                    const float* n = &attrib.normals[3 * (size_t)ix.normal_index];
                    v.normal = float3(n[0], n[1], n[2]);
                }
                m.bounds_min = float3(std::min(m.bounds_min.x, v.position.x), std::min(m.bounds_min.y, v.position.y), std::min(m.bounds_min.z, v.position.z));
                m.bounds_max = float3(std::max(m.bounds_max.x, v.position.x), std::max(m.bounds_max.y, v.position.y), std::max(m.bounds_max.z, v.position.z));
                it = seen.emplace(key, (int)m.vertices.size()).first;
                m.vertices.push_back(v);
            }
            m.indices.push_back(it->second);
            if (k % 3 == 0) m.material_ids.push_back(k / 3 < s.mesh.material_ids.size() ? s.mesh.material_ids[k / 3] : -1);
        }
    }
    return m;
}

std::vector<Material> buildMaterialTable(const std::vector<ptobj::material_t>& mtl, const float3& unassigned) {
    std::vector<Material> table;
    table.reserve(mtl.size() + 1);
    auto peak_of = [](const float* c) { return c[0] > c[1] ? (c[0] > c[2] ? c[0] : c[2]) : (c[1] > c[2] ? c[1] : c[2]); };
    for (const ptobj::material_t& m : mtl) {
        const float* emission = nullptr;
        if (peak_of(m.emission) > 0.f) emission = m.emission;
        else if (peak_of(m.ambient) > 1.0f) emission = m.ambient;
        Material row;
        if (emission) {
            const float peak = peak_of(emission);
            row.r = emission[0] / peak; row.g = emission[1] / peak; row.b = emission[2] / peak;
            row.type = peak;
        } else {
            row.r = m.diffuse[0]; row.g = m.diffuse[1]; row.b = m.diffuse[2];
            row.type = 0.f;
        }
        table.push_back(row);
    }
    Material grey;
    grey.r = unassigned.x; grey.g = unassigned.y; grey.b = unassigned.z;
    grey.type = 0.f;
    table.push_back(grey);
    return table;
}

std::vector<int32_t> resolveMaterialIds(const Mesh& mesh, size_t table_rows) {
    std::vector<int32_t> ids(mesh.indices.size() / 3, (int32_t)table_rows - 1);
    for (size_t t = 0; t < ids.size() && t < mesh.material_ids.size(); ++t) {
        const int m = mesh.material_ids[t];
        if (m >= 0 && (size_t)m + 1 < table_rows) ids[t] = m;
    }
    return ids;
}

bool buildReferenceScene(Scene& scene, const char* obj_path, const char* mtl_basedir, int width, int height, std::string* err) {
    scene.camera.position = float3(0.f, 0.f, 0.f);
    scene.camera.initialize(60.f, 0.1f, 100.f, (float)width, (float)height);
    scene.camera.lookAt(float3(0.0f, 0.0f, 0.0f), float3(0.0f, 0.0f, 1.f), float3(0.0f, 1.0f, 0.0f));
    ptobj::attrib_t attrib;
    std::vector<ptobj::shape_t> shapes;
    std::vector<ptobj::material_t> materials;
    if (!ptobj::LoadObj(&attrib, &shapes, &materials, err, obj_path, mtl_basedir)) return false;
    scene.meshes.clear();
    scene.meshes.push_back(buildMesh(attrib, shapes));
    buildReferenceSpheres(scene);
    return true;
}

}  // namespace pthost
