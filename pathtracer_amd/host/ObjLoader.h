// Wavefront OBJ / MTL reader for the host side.
//
// The reference host loads its scene with the third-party tiny_obj_loader v1.0.5 that it vendors
// (libs/tiny_obj_loader.h, called at PathTracer.cpp:1166-1170). This is an independent reader that returns
// the same data for the same file: same field names, same triangle order (which is what triangle ids mean),
// same float values. The behaviours that decide those results are kept on purpose:
//   * polygons become triangle fans (f0, f[k-1], f[k])              libs/tiny_obj_loader.h:899-922
//   * 1-based, 0 and negative (relative) indices                    libs/tiny_obj_loader.h:416-420
//   * decimal text -> double by digit accumulation -> float         libs/tiny_obj_loader.h:465-588
//   * shape / material-group bookkeeping on g, o, usemtl            libs/tiny_obj_loader.h:1518-1650, 1701-1709
//   * MTL statements and defaults                                   libs/tiny_obj_loader.h:844-879, 956-1324
// It is checked against dumps of the reference reader's output for every OBJ asset (tests/test_obj_loader.py).
#pragma once

#include <string>
#include <vector>

namespace ptobj {

struct index_t {
    int vertex_index;
    int normal_index;
    int texcoord_index;
};

struct mesh_t {
    std::vector<index_t> indices;                  // 3 per triangle after triangulation
    std::vector<unsigned char> num_face_vertices;  // 3 for every triangulated face
    std::vector<int> material_ids;                 // per face, -1 = none
};

struct shape_t {
    std::string name;
    mesh_t mesh;
};

struct attrib_t {
    std::vector<float> vertices;   // xyz
    std::vector<float> normals;    // xyz
    std::vector<float> texcoords;  // uv
};

struct material_t {
    std::string name;
    float ambient[3];
    float diffuse[3];
    float specular[3];
    float transmittance[3];
    float emission[3];
    float shininess;
    float ior;
    float dissolve;
    int illum;
    float roughness, metallic, sheen, clearcoat_thickness, clearcoat_roughness, anisotropy, anisotropy_rotation;
    std::string ambient_texname, diffuse_texname, specular_texname, specular_highlight_texname, bump_texname,
        displacement_texname, alpha_texname, emissive_texname, normal_texname;
};

// Same call shape as the reference's loader. `mtl_basedir` is prepended to the name given by `mtllib`; the
// reference passes none, so its MTL files are only found in the working directory. Returns false only when
// the OBJ file cannot be opened. Warnings (e.g. a missing MTL file) are appended to *err.
bool LoadObj(attrib_t* attrib, std::vector<shape_t>* shapes, std::vector<material_t>* materials, std::string* err,
             const char* filename, const char* mtl_basedir = nullptr, bool triangulate = true);

// Parses one decimal number the way the OBJ reader does (exposed for tests). Returns `fallback` when the text is
// not a number in the accepted grammar: [sign] digits [. digits] [(e|E) [sign] digits].
float ParseFloatText(const char* begin, const char* end, double fallback = 0.0);

}  // namespace ptobj
