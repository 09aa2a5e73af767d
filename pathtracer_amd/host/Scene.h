// Host-side scene types and construction, mirroring what the reference's main() builds before its render loop
// (PathTracer.cpp:1160-1244): camera, the OBJ mesh as interleaved {position, normal} vertices + an index list,
// and the ten random spheres / materials. Layouts equal the reference's structs because they are what gets
// uploaded (struct Vertex PathTracer.cpp:71-75, Sphere :589-592, Material :87-93).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "ObjLoader.h"

namespace pthost {

struct float3 {
    float x, y, z;
    float3() : x(0.f), y(0.f), z(0.f) {}
    float3(float s) : x(s), y(s), z(s) {}
    float3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
inline float3 operator+(const float3& a, const float3& b) { return float3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float3 operator-(const float3& a, const float3& b) { return float3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float3 operator*(const float3& a, const float3& b) { return float3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float3 operator*(const float3& a, float s) { return float3(a.x * s, a.y * s, a.z * s); }
float length(const float3& v);               // sqrtf(x*x + y*y + z*z), summed left to right
float3 normalize(const float3& v);           // v * (1.f / length(v))
float3 cross(const float3& a, const float3& b);

struct Vertex {
    float3 position;
    float3 normal;
};
static_assert(sizeof(Vertex) == 24, "Vertex is uploaded verbatim: 6 floats");

struct Sphere {
    float3 center;
    float radius;
};
static_assert(sizeof(Sphere) == 16, "Sphere is uploaded verbatim: 4 floats");

struct Material {
    enum { LAMBERT, METAL, DIELECTRIC };
    float r, g, b;
    float type;
};
static_assert(sizeof(Material) == 16, "Material is uploaded verbatim: 4 floats");

// Pinhole camera expressed as the four vectors the program receives as uniforms.
struct Camera {
    float3 position;
    float3 lowerLeftCorner;
    float3 horizontal;
    float3 vertical;
    float width = 0.f, height = 0.f, nearPlane = 0.f, farPlane = 0.f, fov = 0.f, aspectRatio = 1.f;

    // PathTracer.cpp:709-716 (fov = full vertical field of view in degrees)
    void initialize(float fov_, float near_, float far_, float width_, float height_);
    // PathTracer.cpp:691-707
    void lookAt(const float3& eye, const float3& target, const float3& up);
};

struct Mesh {
    std::vector<Vertex> vertices;
    std::vector<int> indices;        // 3 per triangle; triangle id = position / 3
    std::vector<int> material_ids;   // per triangle, from `usemtl` (-1 = none); the reference loads but never uses them
    float3 bounds_min, bounds_max;
    bool has_vertex_normals = false; // false: the OBJ has fewer `vn` than `v`, use geometric normals
};

struct Scene {
    std::vector<Sphere> spheres;
    std::vector<Mesh> meshes;
    std::vector<Material> materials;
    Camera camera;
};

// PathTracer.cpp:1178-1209: positions scaled and translated in z only, normals taken from attrib.normals with the
// POSITION index (the reference assumes `vn` lines parallel to `v` lines), triangles = vertex_index of every corner,
// shapes concatenated in file order. Where the file has fewer normals than positions the reference reads past the
// end of the array; here the normals are zero and has_vertex_normals is false.
Mesh buildMesh(const ptobj::attrib_t& attrib, const std::vector<ptobj::shape_t>& shapes, float scale = 1.0f,
               const float3& translation = float3(0.0f, 0.0f, 3.f));

// The same mesh with the normals the OBJ file actually assigns: every corner takes attrib.normals[normal_index] (the
// reference takes attrib.normals[vertex_index], which is only right for files whose `vn` lines run parallel to their `v`
// lines). Vertices are the distinct (vertex_index, normal_index) pairs in order of first use, so triangle order -- and
// with it every triangle id -- is unchanged. Corners without a normal index get a zero normal; has_vertex_normals is
// true only when every corner has one.
Mesh buildMeshIndexedNormals(const ptobj::attrib_t& attrib, const std::vector<ptobj::shape_t>& shapes, float scale = 1.0f,
                             const float3& translation = float3(0.0f, 0.0f, 3.f));

// Table for the diffuse/emissive shading mode (PT_SHADE_MATERIALS), one row per MTL material plus a last row for
// triangles without `usemtl`. The reference parses MTL files (tiny_obj_loader.h:956-1324) and never looks at the result,
// so the rule is this build's: emission = Ke, or Ka when Ke is zero and a Ka component exceeds 1 (cornell_box.mtl marks
// its light with `Ka 20 20 20` and has no Ke). An emitter is stored as (colour / peak, peak) with peak = the largest
// component; everything else as (Kd, 0). Material::type holds the fourth value.
std::vector<Material> buildMaterialTable(const std::vector<ptobj::material_t>& mtl, const float3& unassigned = float3(0.5f));
// Per-triangle row numbers for that table: the mesh's material ids with -1 (and anything outside the table) mapped to
// the last row.
std::vector<int32_t> resolveMaterialIds(const Mesh& mesh, size_t table_rows);

// PathTracer.cpp:765-768
float nextFloat(float min, float max);
// PathTracer.cpp:1216-1244: srand(2 ^ 17 - 1) -- an XOR, i.e. srand(18) -- then ten spheres and ten materials
// from one libc rand() stream. Results depend on the C library (glibc here).
void buildReferenceSpheres(Scene& scene);

// Everything main() sets up before the loop, for `obj_path` at `width` x `height`. Returns false (and a message)
// when the OBJ cannot be read.
bool buildReferenceScene(Scene& scene, const char* obj_path, const char* mtl_basedir, int width, int height,
                         std::string* err);

}  // namespace pthost
