// TEST INFRASTRUCTURE ONLY. Drives the REFERENCE's own CPU tracer -- float3 / Ray / HitRecord::hit / AABB / Mesh / Sphere /
// Camera / nextFloat / randomUnitVector / scatter / Scene::color and the 16x16-tile frame loop, compiled from the lines
// where they lie in /root/reference/PathTracer.cpp (44-641, 680-1052 and 1319-1368; oracle/Makefile passes them in through
// PT_EXTRACT and PT_FRAME, nothing of them is kept) -- on a scene given as binary arrays, and writes per-pixel statistics:
// the running sum the reference keeps in Framebuffer::accumulationBuffer, and the sum of squares of the per-frame radiance,
// both in the reference's tile-traversal order. The reference code is compiled out in its own build (#define USE_GLSL,
// PathTracer.cpp:1246); its random numbers are one serial libc rand() stream, so runs are reproducible (srand) but no
// parallel implementation can be compared with it sample by sample: tests compare per-pixel means and variances.
//
//   cpu_tracer job.txt     job lines: key value...; keys width height passes seed fov near far eye3 target3 up3
//                          spheres FILE (10x4 f32) materials FILE (10x4 f32) [vertices FILE (nx6 f32) indices FILE (mx3 i32)]
//                          out_sum FILE out_sumsq FILE out_color FILE
#include <algorithm>
#include <array>
#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define _USE_MATH_DEFINES
#include <math.h>
#include <map>
#include <sstream>
#include <fstream>
#include <string>
#include <vector>

#include "Framebuffer.h"  // the reference's (-I$(REF))
#include PT_EXTRACT

template <typename T>
static std::vector<T> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    std::vector<char> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<T> out(bytes.size() / sizeof(T));
    memcpy(out.data(), bytes.data(), out.size() * sizeof(T));
    return out;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: cpu_tracer job.txt\n"); return 2; }
    std::map<std::string, std::vector<std::string>> job;
    std::ifstream jf(argv[1]);
    for (std::string line; std::getline(jf, line);) {
        std::istringstream ss(line);
        std::string key, v;
        ss >> key;
        while (ss >> v) job[key].push_back(v);
    }
    auto num = [&](const char* k, int i = 0) { return atof(job.at(k).at(i).c_str()); };
    const int width = (int)num("width"), height = (int)num("height"), passes = (int)num("passes");

    Framebuffer fb;
    fb.create(width, height);
    Scene scene;
    // PathTracer.cpp:1160-1162
    scene.camera.position = float3((float)num("eye", 0), (float)num("eye", 1), (float)num("eye", 2));
    scene.camera.initialize((float)num("fov"), (float)num("near"), (float)num("far"), fb.width, fb.height);
    scene.camera.lookAt(scene.camera.position, float3((float)num("target", 0), (float)num("target", 1), (float)num("target", 2)),
                        float3((float)num("up", 0), (float)num("up", 1), (float)num("up", 2)));
    if (job.count("vertices")) {
        // the mesh as PathTracer.cpp:1166-1211 leaves it: vertices with their normals, indices, bounds over the positions
        const std::vector<float> v = read_file<float>(job["vertices"][0]);
        const std::vector<int> idx = read_file<int>(job["indices"][0]);
        Mesh m;
        m.bounds.min = vec3(FLT_MAX);
        m.bounds.max = vec3(-FLT_MAX);
        for (size_t i = 0; i + 5 < v.size(); i += 6) {
            m.vertices.emplace_back(Vertex{{v[i], v[i + 1], v[i + 2]}, {v[i + 3], v[i + 4], v[i + 5]}});
            m.bounds.insert(vec3(v[i], v[i + 1], v[i + 2]));
        }
        m.indices.assign(idx.begin(), idx.end());
        scene.meshes.reserve(1);
        scene.meshes.emplace_back(std::move(m));
    }
    const std::vector<float> sph = read_file<float>(job.at("spheres")[0]), mat = read_file<float>(job.at("materials")[0]);
    for (size_t i = 0; i + 3 < sph.size(); i += 4) scene.spheres.emplace_back(Sphere{{sph[i], sph[i + 1], sph[i + 2]}, sph[i + 3]});
    for (size_t i = 0; i + 3 < mat.size(); i += 4) scene.materials.emplace_back(Material{mat[i], mat[i + 1], mat[i + 2], mat[i + 3]});

    srand((unsigned)num("seed"));
    const size_t n3 = (size_t)width * height * 3;
    std::vector<double> sumsq(n3, 0.0);
    std::vector<float> before(n3, 0.f);
    uint64_t rays = 0;
    for (int samples = 1; samples <= passes; ++samples) {
        uint64_t totalRayCount = 0;
        memcpy(before.data(), fb.accumulationBuffer, n3 * sizeof(float));
        {
#include PT_FRAME
        }
        rays += totalRayCount;
        for (size_t k = 0; k < n3; ++k) {
            const double s = (double)fb.accumulationBuffer[k] - (double)before[k];   // this frame's radiance
            sumsq[k] += s * s;
        }
    }
    std::vector<float> sq(n3);
    for (size_t k = 0; k < n3; ++k) sq[k] = (float)sumsq[k];
    FILE* f;
    if ((f = fopen(job.at("out_sum")[0].c_str(), "wb"))) { fwrite(fb.accumulationBuffer, sizeof(float), n3, f); fclose(f); }
    if ((f = fopen(job.at("out_sumsq")[0].c_str(), "wb"))) { fwrite(sq.data(), sizeof(float), n3, f); fclose(f); }
    if ((f = fopen(job.at("out_color")[0].c_str(), "wb"))) { fwrite(fb.colorBuffer, 1, (size_t)width * height * 4, f); fclose(f); }
    printf("{\"width\": %d, \"height\": %d, \"passes\": %d, \"rays\": %llu}\n", width, height, passes, (unsigned long long)rays);
    return 0;
}
