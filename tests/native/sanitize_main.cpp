// TEST INFRASTRUCTURE: the CPU side of the build under AddressSanitizer + UndefinedBehaviorSanitizer (host compiler only; no GPU
// code is involved). One executable, linked statically against the sanitizer runtimes, that
//   1. loads every OBJ/MTL asset with the host's reader and runs the scene construction on it (mesh in both normal pairings,
//      material table, per-triangle rows, camera, sphere tables, the synthetic workloads, Framebuffer stores);
//   2. feeds the reader mutated OBJ/MTL text -- the reader parses untrusted files; the reference's own scene set-up has
//      undefined behaviour on meshes without normals (PathTracer.cpp:1191-1193 reads attrib.normals with position indices);
//   3. runs the parity oracle (oracle/pt_oracle.c, compiled into this executable with the same sanitizers) on a small frame in
//      every mode and prints a checksum of each result, which tests/test_sanitizers.py compares with the regular build's.
// Usage: sanitize_check <assets dir> <scratch dir> <fuzz cases>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../pathtracer_amd/host/Framebuffer.h"
#include "../../pathtracer_amd/host/ObjLoader.h"
#include "../../pathtracer_amd/host/Scene.h"
#include "../../pathtracer_amd/host/Workloads.h"

extern "C" {
struct PtoScene {
    int32_t num_triangles, num_vertices;
    const uint32_t* indices;
    const float* vertices;
    const float* spheres;
    const float* materials;
    const int32_t* material_ids;
};
struct PtoUniforms {
    float cam_pos[3], cam_llc[3], cam_hor[3], cam_ver[3];
    int32_t num_triangles, depth, width, height, variant, normal_mode, shading_mode;
};
uint64_t pto_render(const PtoScene*, const PtoUniforms*, int first_sample, int num_samples, float* rgba, int x0, int y0, int w, int h);
uint64_t pto_render_sum(const PtoScene*, const PtoUniforms*, int first_sample, int stride, int count, float* rgba_sum, int x0, int y0, int w, int h);
uint64_t pto_cpu_tracer(const PtoScene*, const PtoUniforms*, float max_t, uint32_t seed, int first_sample, int stride, int count, float* rgba_sum,
                        int x0, int y0, int w, int h);
void pto_primary(const PtoScene*, const PtoUniforms*, int32_t* ids, float* ts, float* rds, int x0, int y0, int w, int h);
void pto_present(const float* rgba, int num_pixels, uint8_t* out_rgba8);
void pto_rng_stream(int px, int py, float samples, int n, float* out);
void pto_lp_math(int kind, int n, const float* a, const float* b, float* out);
}

namespace {

uint64_t fnv1a(const void* data, size_t bytes, uint64_t h = 1469598103934665603ull) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < bytes; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

struct Rng {
    uint64_t s;
    uint32_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); }
    uint32_t below(uint32_t n) { return n ? next() % n : 0; }
};

std::string read_file(const std::string& path) {
    std::string out;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return out;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return out;
}

bool write_file(const std::string& path, const std::string& text) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = text.empty() || fwrite(text.data(), text.size(), 1, f) == 1;
    fclose(f);
    return ok;
}

// everything the scene construction makes of a parsed file; returns a checksum so that nothing is optimised away
uint64_t build_everything(const ptobj::attrib_t& attrib, const std::vector<ptobj::shape_t>& shapes, const std::vector<ptobj::material_t>& mtl) {
    uint64_t h = 0;
    const pthost::Mesh a = pthost::buildMesh(attrib, shapes);
    const pthost::Mesh b = pthost::buildMeshIndexedNormals(attrib, shapes, 0.5f, pthost::float3(0.f, 0.f, 1.f));
    for (const pthost::Mesh* m : {&a, &b}) {
        if (!m->vertices.empty()) h = fnv1a(m->vertices.data(), m->vertices.size() * sizeof(pthost::Vertex), h);
        if (!m->indices.empty()) h = fnv1a(m->indices.data(), m->indices.size() * sizeof(int), h);
        if (!m->material_ids.empty()) h = fnv1a(m->material_ids.data(), m->material_ids.size() * sizeof(int), h);
    }
    const std::vector<pthost::Material> table = pthost::buildMaterialTable(mtl);
    h = fnv1a(table.data(), table.size() * sizeof(pthost::Material), h);
    const std::vector<int32_t> ids = pthost::resolveMaterialIds(a, table.size());
    if (!ids.empty()) h = fnv1a(ids.data(), ids.size() * sizeof(int32_t), h);
    for (const ptobj::material_t& m : mtl) h = fnv1a(m.name.data(), m.name.size(), h);
    for (const ptobj::shape_t& s : shapes) h = fnv1a(s.name.data(), s.name.size(), h);
    return h;
}

const char* kTokens[] = {"f ", "v ", "vn ", "vt ", "usemtl ", "mtllib ", "g ", "o ", "s ", "-1", "0", "99999999999999999999", "1/2/3", "1//3", "//", "/",
                         "1e400", "-1e-400", "nan", "inf", "-", "+", ".", "e", "E+", "\r\n", "\n", "\t", " ", "#", "newmtl ", "Kd ", "Ka ", "Ke ", "Ns ",
                         "map_Kd ", "illum ", "d ", "Tr ", "-2147483648", "2147483647", "4294967296", "0x10", "1.5.5", "--3", "1e", "f 1 2", "f 1 2 3 4 5 6 7 8 9"};

std::string mutate(const std::string& base, Rng& rng) {
    std::string t = base;
    const int ops = 1 + (int)rng.below(6);
    for (int k = 0; k < ops; ++k) {
        const size_t n = t.size();
        switch (rng.below(7)) {
            case 0: if (n) t[rng.below((uint32_t)n)] = (char)rng.below(256); break;
            case 1: if (n) { const size_t a = rng.below((uint32_t)n), len = rng.below(64); t.erase(a, len); } break;
            case 2: if (n) { const size_t a = rng.below((uint32_t)n), len = rng.below(200); t.insert(rng.below((uint32_t)n), t.substr(a, len)); } break;
            case 3: t.insert(n ? rng.below((uint32_t)n) : 0, kTokens[rng.below(sizeof kTokens / sizeof kTokens[0])]); break;
            case 4: if (n) t.resize(rng.below((uint32_t)n)); break;
            case 5: if (n) { const size_t a = rng.below((uint32_t)n); t.insert(a, std::string(1 + rng.below(3), (char)('0' + rng.below(10)))); } break;
            default: if (n) { const size_t a = rng.below((uint32_t)n); t.insert(a, 1, '\0'); } break;
        }
    }
    return t;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <assets dir> <scratch dir> <fuzz cases>\n", argv[0]);
        return 2;
    }
    const std::string assets = std::string(argv[1]) + "/", scratch = std::string(argv[2]) + "/";
    const int fuzz_cases = atoi(argv[3]);

    // ---- 1. assets through the reader and the scene construction
    const char* names[] = {"cornell_box", "monkeyhead2", "suzanne", "catmark_torus_creases0", "cube", "no_material", "monkeyhead"};
    std::vector<std::string> obj_texts, mtl_texts;
    for (const char* n : names) {
        ptobj::attrib_t attrib;
        std::vector<ptobj::shape_t> shapes;
        std::vector<ptobj::material_t> mtl;
        std::string err;
        if (!ptobj::LoadObj(&attrib, &shapes, &mtl, &err, (assets + n + ".obj").c_str(), assets.c_str())) {
            fprintf(stderr, "cannot load %s: %s\n", n, err.c_str());
            return 1;
        }
        size_t tris = 0;
        for (const ptobj::shape_t& s : shapes) tris += s.mesh.indices.size() / 3;
        printf("asset %s %zu %zu %zu %016llx\n", n, attrib.vertices.size() / 3, tris, mtl.size(), (unsigned long long)build_everything(attrib, shapes, mtl));
        obj_texts.push_back(read_file(assets + n + ".obj"));
        if (obj_texts.back().size() > 6000) obj_texts.back().resize(6000);   // the head of a large file is enough to mutate
    }
    for (const char* n : {"cornell_box.mtl", "cube.mtl", "monkeyhead2.mtl"}) mtl_texts.push_back(read_file(assets + n));
    for (const char* n : {"cornell", "monkeyhead2", "world", "torus", "torus_distorted"}) {
        pthost::Workload w;
        std::string err;
        if (!pthost::buildWorkload(n, assets, 64, 40, w, &err)) {
            fprintf(stderr, "workload %s: %s\n", n, err.c_str());
            return 1;
        }
        uint64_t h = fnv1a(w.mesh.vertices.data(), w.mesh.vertices.size() * sizeof(pthost::Vertex));
        h = fnv1a(&w.scene.camera.lowerLeftCorner, sizeof(pthost::float3), h);
        printf("workload %s %zu %016llx\n", n, w.mesh.indices.size() / 3, (unsigned long long)h);
    }
    {
        pthost::Workload w;
        std::string err;
        if (pthost::buildWorkload("teapot", assets, 0, 0, w, &err) || err.empty()) return 1;
        if (pthost::buildWorkload("cornell", scratch + "nowhere", 0, 0, w, &err)) return 1;
    }
    struct Size { int s[2]; };
    for (const Size& sz : {Size{{17, 9}}, Size{{32, 32}}, Size{{1, 1}}, Size{{48, 20}}}) {
        const int* size = sz.s;
        Framebuffer fb;
        fb.create(size[0], size[1]);
        std::vector<float> img((size_t)size[0] * size[1] * 4);
        for (size_t i = 0; i < img.size(); ++i) img[i] = (float)(i % 97) / 31.0f - 0.5f;   // includes negatives and values above 1
        fb.storeMean(img.data(), 7);
        fb.storeSumsTileOrder(img.data(), 3);
        printf("framebuffer %dx%d %016llx\n", size[0], size[1],
               (unsigned long long)fnv1a(fb.colorBuffer, (size_t)size[0] * size[1] * 4, fnv1a(fb.accumulationBuffer, (size_t)size[0] * size[1] * 12)));
        fb.destroy();
    }

    // ---- 2. mutated text through the reader
    Rng rng{0x9E3779B97F4A7C15ull};
    uint64_t fuzz_hash = 0;
    int loaded = 0;
    const std::string obj_path = scratch + "fuzz.obj", mtl_path = scratch + "fuzz.mtl";
    for (int c = 0; c < fuzz_cases; ++c) {
        std::string obj = mutate(obj_texts[rng.below((uint32_t)obj_texts.size())], rng);
        if (rng.below(3) == 0) obj = "mtllib fuzz.mtl\n" + obj;
        const std::string mtl = mutate(mtl_texts[rng.below((uint32_t)mtl_texts.size())], rng);
        if (!write_file(obj_path, obj) || !write_file(mtl_path, mtl)) {
            fprintf(stderr, "cannot write into %s\n", scratch.c_str());
            return 1;
        }
        ptobj::attrib_t attrib;
        std::vector<ptobj::shape_t> shapes;
        std::vector<ptobj::material_t> mats;
        std::string err;
        if (ptobj::LoadObj(&attrib, &shapes, &mats, &err, obj_path.c_str(), scratch.c_str(), rng.below(8) != 0)) {
            ++loaded;
            fuzz_hash = fnv1a(&fuzz_hash, sizeof fuzz_hash, build_everything(attrib, shapes, mats));
        }
        const std::string num = mutate("-12.5e+3", rng);
        const float f = ptobj::ParseFloatText(num.data(), num.data() + num.size(), 1.0);
        fuzz_hash = fnv1a(&f, sizeof f, fuzz_hash);
    }
    remove(obj_path.c_str());
    remove(mtl_path.c_str());
    printf("fuzz %d %d %016llx\n", fuzz_cases, loaded, (unsigned long long)fuzz_hash);

    // ---- 3. the oracle under the sanitizers
    pthost::Workload w;
    std::string err;
    if (!pthost::buildWorkload("monkeyhead2", assets, 40, 24, w, &err)) return 1;
    std::vector<uint32_t> indices(w.mesh.indices.begin(), w.mesh.indices.end());
    std::vector<float> table(40, 0.f);
    const float rows[4][4] = {{0.5f, 0.5f, 0.5f, 0.f}, {0.85f, 0.3f, 0.2f, 0.f}, {0.25f, 0.7f, 0.35f, 0.f}, {1.0f, 0.9f, 0.7f, 3.0f}};
    memcpy(table.data(), rows, sizeof rows);
    std::vector<int32_t> mids(indices.size() / 3);
    for (size_t t = 0; t < mids.size(); ++t) mids[t] = t % 11 == 5 ? 3 : (int32_t)(t % 3);
    PtoScene sc;
    sc.num_triangles = (int32_t)(indices.size() / 3);
    sc.num_vertices = (int32_t)w.mesh.vertices.size();
    sc.indices = indices.data();
    sc.vertices = &w.mesh.vertices[0].position.x;
    sc.spheres = &w.scene.spheres[0].center.x;
    sc.materials = &w.scene.materials[0].r;
    sc.material_ids = mids.data();
    PtoUniforms u;
    const pthost::Camera& cam = w.scene.camera;
    const pthost::float3* src[4] = {&cam.position, &cam.lowerLeftCorner, &cam.horizontal, &cam.vertical};
    float* dst[4] = {u.cam_pos, u.cam_llc, u.cam_hor, u.cam_ver};
    for (int k = 0; k < 4; ++k) { dst[k][0] = src[k]->x; dst[k][1] = src[k]->y; dst[k][2] = src[k]->z; }
    u.num_triangles = sc.num_triangles;
    u.depth = 10;
    u.width = 40;
    u.height = 24;
    const size_t px = 40 * 24;
    struct Mode { const char* name; int variant, normal_mode, shading; };
    for (const Mode& m : {Mode{"mesh", 1, 0, 0}, Mode{"mesh_geo", 1, 1, 0}, Mode{"world", 0, 0, 0}, Mode{"materials", 1, 0, 1}, Mode{"materials_unit", 1, 1, 2}}) {
        u.variant = m.variant;
        u.normal_mode = m.normal_mode;
        u.shading_mode = m.shading;
        sc.materials = m.shading ? table.data() : &w.scene.materials[0].r;
        std::vector<float> rgba(px * 4, 0.f);
        for (size_t i = 0; i < px; ++i) rgba[4 * i + 3] = 1.0f;
        uint64_t rays = pto_render(&sc, &u, 1, 2, rgba.data(), 0, 0, 40, 24);
        rays += pto_render(&sc, &u, 3, 1, rgba.data(), 5, 3, 20, 11);   // a window, continuing the image
        std::vector<float> sums(px * 4, 0.f);
        rays += pto_render_sum(&sc, &u, 2, 3, 2, sums.data(), 0, 0, 40, 24);
        std::vector<int32_t> ids(px);
        std::vector<float> ts(px), rds(px * 3);
        pto_primary(&sc, &u, ids.data(), ts.data(), rds.data(), 0, 0, 40, 24);
        std::vector<uint8_t> rgba8(px * 4);
        pto_present(rgba.data(), (int)px, rgba8.data());
        uint64_t h = fnv1a(rgba.data(), rgba.size() * 4);
        h = fnv1a(sums.data(), sums.size() * 4, h);
        h = fnv1a(ids.data(), ids.size() * 4, h);
        h = fnv1a(ts.data(), ts.size() * 4, h);
        h = fnv1a(rgba8.data(), rgba8.size(), h);
        printf("oracle %s %llu %016llx\n", m.name, (unsigned long long)rays, (unsigned long long)h);
    }
    {
        u.variant = 1; u.normal_mode = 0; u.shading_mode = 0;
        sc.materials = &w.scene.materials[0].r;
        std::vector<float> sums(px * 4, 0.f);
        const uint64_t rays = pto_cpu_tracer(&sc, &u, 99.9f, 18u, 1, 1, 2, sums.data(), 0, 0, 40, 24);
        printf("oracle cpu_tracer %llu %016llx\n", (unsigned long long)rays, (unsigned long long)fnv1a(sums.data(), sums.size() * 4));
        float draws[16];
        pto_rng_stream(3, 5, 7.0f, 16, draws);
        const float a[8] = {0.0f, 1.0f, -40.0f, 6.2831f, 1e-30f, 0.5f, 130.0f, -130.0f}, b[8] = {0.5f, 1.0f / 1025.0f, 2.0f, 0.0f, 4.0f, 0.25f, 1.0f, 3.0f};
        float out[4][8];
        for (int kind = 0; kind < 4; ++kind) pto_lp_math(kind, 8, a, b, out[kind]);
        printf("oracle math %016llx\n", (unsigned long long)fnv1a(out, sizeof out, fnv1a(draws, sizeof draws)));
    }
    printf("done\n");
    return 0;
}
