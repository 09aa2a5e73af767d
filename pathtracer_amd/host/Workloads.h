// Named workloads of the benchmark configurations, built on the host exactly once and in one place: which OBJ, which camera,
// which program, which material table -- and the two synthetic scenes (the ~1M-triangle instanced scene, the torus next to
// its distorted copy). The reference has one hard-coded scene (PathTracer.cpp:1160-1212: monkeyhead2.obj at z + 3, camera at
// the origin looking down +z, 60 degrees); everything else here is this build's choice, stated where it is made.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "Scene.h"

namespace pthost {

// Deterministic uniform [0, 1) stream: 32-bit LCG (multiplier 1664525, increment 1013904223), top 24 bits of each state.
// Integer arithmetic, so the same on every machine.
std::vector<double> lcgUniform(size_t count, uint32_t seed = 20260101u);

// `instances` copies of `base` on a jittered grid `columns` wide facing the camera, each with its own yaw, jitter and depth
// from the LCG stream, everything scaled by `unit` (Cornell-like units, so that the shader's absolute `det >= 1e-4` cull,
// pathtracing.fs:142, keeps the small triangles). Triangle ids of copy k follow those of copy k - 1. Normals are zero
// (geometric-normal mode).
Mesh instancedMesh(const Mesh& base, int instances = 254, int columns = 16, double spacing = 2.8, double unit = 30.0,
                   uint32_t seed = 20260101u);

// A deterministically distorted copy of a mesh: every vertex moved by up to `amplitude` along each axis with offsets from the
// LCG stream, then translated by `offset`. Stands in for the reference's data/distordedmesh/*.bin, a container that nothing in
// the reference loads.
Mesh distortedCopy(const Mesh& base, double amplitude, const double offset[3], uint32_t seed = 20260101u);

// `base` moved `separation` to the left and its distorted copy the same distance to the right: one mesh of twice the
// triangles (ids of the copy follow those of the original).
Mesh torusWithDistortedCopy(const Mesh& base, double separation = 1.4);

struct Workload {
    std::string name;
    int program_kind = 1;            // 0 = IntersectWorld (10 spheres), 1 = IntersectMesh
    int width = 0, height = 0;
    Mesh mesh;
    Scene scene;                     // camera, the ten spheres and their materials (meshes unused: see `mesh`)
    std::vector<Material> material_table;   // rows (r, g, b, e) of the diffuse/emissive mode
    std::vector<int32_t> material_ids;      // per triangle, rows of that table
};

// name: cornell | monkeyhead2 | suzanne | torus | world | instanced | torus_distorted. width/height <= 0: the workload's own
// size (1920x1080; 3840x2160 for torus, torus_distorted and instanced; 1280x720 for world). OBJ and MTL files are read from
// `assets_dir`. Returns false with a message when a file cannot be read or the name is unknown.
bool buildWorkload(const std::string& name, const std::string& assets_dir, int width, int height, Workload& out, std::string* err);

}  // namespace pthost
