// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or shipped with the product.
//
// Runs the reference's own OBJ/MTL parser (the tiny_obj_loader.h vendored in the reference checkout,
// compiled from where it lies -- it is NOT copied into this repository) on one OBJ file and dumps
// what it returns, so that the product's independent loader can be checked against it.
//
// Reference call being reproduced: /root/reference/PathTracer.cpp:1166-1170
//   tinyobj::LoadObj(&attrib, &shapes, &materials, &err, "<file>.obj")   (mtl_basedir defaults to NULL)
// An optional second argument supplies mtl_basedir (the reference never passes one, which is why its
// MTL files are only found when they sit in the working directory).
//
// Output directory layout (all little-endian raw arrays):
//   v.f32        attrib.vertices            (3 floats per position)
//   vn.f32       attrib.normals             (3 floats per normal)
//   idx_v.i32    vertex_index of every triangulated corner, all shapes concatenated in shape order
//   idx_vn.i32   normal_index of every corner (-1 when the face has none)
//   mat_ids.i32  material id per triangle (-1 = none)
//   shapes.txt   one line per shape: <triangle count> <name>
//   materials.txt one line per material: name, then Ka Kd Ks Kt Ke (3 hex floats each), Ns Ni d illum
//   err.txt      loader warning/error text
//
// Build: see Makefile (target _ref/tinyobj_dump).

#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"  // resolved with -I/root/reference/libs

#include <cstdio>
#include <string>
#include <vector>

template <class T>
static void dump(const std::string& path, const std::vector<T>& v) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(2); }
    if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: tinyobj_dump <file.obj> <outdir> [mtl_basedir]\n");
        return 2;
    }
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string err;
    const char* basedir = argc > 3 ? argv[3] : nullptr;
    bool ok = tinyobj::LoadObj(&attrib, &shapes, &materials, &err, argv[1], basedir);
    std::string out = argv[2];

    std::vector<int> idx_v, idx_vn, mat_ids;
    FILE* fs = fopen((out + "/shapes.txt").c_str(), "w");
    for (auto& s : shapes) {
        for (auto& i : s.mesh.indices) {
            idx_v.push_back(i.vertex_index);
            idx_vn.push_back(i.normal_index);
        }
        for (int m : s.mesh.material_ids) mat_ids.push_back(m);
        fprintf(fs, "%zu %s\n", s.mesh.indices.size() / 3, s.name.c_str());
    }
    fclose(fs);
    dump(out + "/v.f32", attrib.vertices);
    dump(out + "/vn.f32", attrib.normals);
    dump(out + "/idx_v.i32", idx_v);
    dump(out + "/idx_vn.i32", idx_vn);
    dump(out + "/mat_ids.i32", mat_ids);

    FILE* fm = fopen((out + "/materials.txt").c_str(), "w");
    for (auto& m : materials) {
        fprintf(fm, "%s", m.name.empty() ? "<unnamed>" : m.name.c_str());
        const float* groups[5] = {m.ambient, m.diffuse, m.specular, m.transmittance, m.emission};
        for (auto g : groups) fprintf(fm, " %a %a %a", g[0], g[1], g[2]);
        fprintf(fm, " %a %a %a %d\n", m.shininess, m.ior, m.dissolve, m.illum);
    }
    fclose(fm);
    FILE* fe = fopen((out + "/err.txt").c_str(), "w");
    fprintf(fe, "%s", err.c_str());
    fclose(fe);
    printf("{\"ok\": %s, \"positions\": %zu, \"normals\": %zu, \"triangles\": %zu, \"shapes\": %zu, \"materials\": %zu}\n",
           ok ? "true" : "false", attrib.vertices.size() / 3, attrib.normals.size() / 3, idx_v.size() / 3,
           shapes.size(), materials.size());
    return ok ? 0 : 1;
}
