// Bounding volume hierarchy over the mesh -- built on the device (by default top-down with the binned surface-area heuristic;
// or clusters merged bottom-up; or Morton codes -> radix sort -> Karras hierarchy) or top-down on the host with the same
// surface-area rules; boxes bottom-up on the device either way -- emitted as 64-byte "two children per record" nodes for
// wave-coherent traversal and as four-child nodes with 16-bit boxes for the per-lane walk. The reference has no acceleration
// structure (its
// shader loops over every triangle, pathtracing.fs:129); this one exists so that scenes far beyond a few thousand
// triangles are traceable, and it must not change a single result: traversal (pt_kernels.hip) only ever SKIPS triangles
// that the exact test could not accept, and resolves equal distances by the lowest triangle index, which is what the
// shader's ascending loop with a strict '<' does.
#include "pt_internal.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

namespace {

// ---- order-preserving float <-> uint mapping for atomic min/max -------------------------------------------------
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// Bounds kept by atomic min / max only ever move one way, so a plain look first (however stale) tells when the atomic cannot
// change anything: nearly always, once a few waves have been through, and then thousands of waves no longer queue on one address.
__device__ __forceinline__ void atomic_min_if_lower(uint32_t* at, uint32_t v) {
    if (v < *(volatile const uint32_t*)at) atomicMin(at, v);
}
__device__ __forceinline__ void atomic_max_if_higher(uint32_t* at, uint32_t v) {
    if (v > *(volatile const uint32_t*)at) atomicMax(at, v);
}

__device__ __forceinline__ unsigned long long spread21(uint32_t v) {  // 21 bits -> every third bit of 63
    unsigned long long x = v & 0x1fffffu;
    x = (x | x << 32) & 0x001f00000000ffffull;
    x = (x | x << 16) & 0x001f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(256) void bvh_init_scene_box(uint32_t* scene_box) {
    if (threadIdx.x < 3) scene_box[threadIdx.x] = 0xffffffffu;       // min
    else if (threadIdx.x < 6) scene_box[threadIdx.x] = 0u;           // max
}

// per triangle: absolute box (6 floats); scene box of the centroids by atomics
__global__ __launch_bounds__(256) void bvh_tri_boxes(const float* __restrict__ vertices, const uint32_t* __restrict__ indices, int n,
                                                     float* __restrict__ tri_box, uint32_t* __restrict__ scene_box) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = vertices + 6 * (size_t)indices[3 * i];
    const float* b = vertices + 6 * (size_t)indices[3 * i + 1];
    const float* c = vertices + 6 * (size_t)indices[3 * i + 2];
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) {
        lo[k] = fminf(a[k], fminf(b[k], c[k]));
        hi[k] = fmaxf(a[k], fmaxf(b[k], c[k]));
        tri_box[6 * (size_t)i + k] = lo[k];
        tri_box[6 * (size_t)i + 3 + k] = hi[k];
        const float ctr = 0.5f * lo[k] + 0.5f * hi[k];
        atomic_min_if_lower(&scene_box[k], f2ord(ctr));
        atomic_max_if_higher(&scene_box[3 + k], f2ord(ctr));
    }
}

// key = 63-bit Morton code of the centroid (21 bits per axis: fine enough that the triangles of one small object in a
// large scene still get distinct cells); value = triangle index. Equal keys are told apart by their sorted position.
__global__ __launch_bounds__(256) void bvh_morton_keys(const float* __restrict__ tri_box, const uint32_t* __restrict__ scene_box, int n,
                                                       unsigned long long* __restrict__ keys, uint32_t* __restrict__ order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = ord2f(scene_box[k]), hi = ord2f(scene_box[3 + k]);
        const float ctr = 0.5f * tri_box[6 * (size_t)i + k] + 0.5f * tri_box[6 * (size_t)i + 3 + k];
        const float ext = hi - lo;
        float u = ext > 0.f ? (ctr - lo) / ext : 0.f;
        u = fminf(fmaxf(u * 2097152.f, 0.f), 2097151.f);
        q[k] = (uint32_t)u;
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    order[i] = (uint32_t)i;
}

__device__ __forceinline__ int key_delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long x = keys[i] ^ keys[j];
    // equal codes: fall back to the sorted positions, which are distinct (Karras 2012, section 4)
    return x ? __builtin_clzll(x) : 64 + __builtin_clz((uint32_t)i ^ (uint32_t)j);
}

// Karras 2012: internal node i of n-1. child refs: >= 0 internal node index; < 0 leaf at sorted position ~ref.
__global__ __launch_bounds__(256) void bvh_hierarchy(const unsigned long long* __restrict__ keys, int n, int* __restrict__ child,
                                                     int* __restrict__ range) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = key_delta(keys, n, i, i + 1) > key_delta(keys, n, i, i - 1) ? 1 : -1;
    const int dmin = key_delta(keys, n, i, i - d);
    int lmax = 2;
    while (key_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (key_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = key_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (key_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    child[2 * i] = first == gamma ? ~gamma : gamma;
    child[2 * i + 1] = last == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    range[2 * i] = first;
    range[2 * i + 1] = last;
}

// One bottom-up sweep: computes the box of every node whose children were finished by an EARLIER launch (done[c] in
// [1, pass)). Kernel boundaries provide the visibility between sweeps; nothing is handed over inside a launch.
__global__ __launch_bounds__(256) void bvh_refit_pass(const uint32_t* __restrict__ order, const int* __restrict__ child,
                                                      const float* __restrict__ tri_box, int n, int pass, float* __restrict__ node_box,
                                                      int* __restrict__ done, int* __restrict__ remaining) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || done[i] != 0) return;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int c = 0; c < 2; ++c) {
        const int ref = child[2 * i + c];
        const float* box;
        if (ref < 0) {
            box = tri_box + 6 * (size_t)order[~ref];
        } else {
            const int dn = done[ref];
            if (dn == 0 || dn >= pass) return;
            box = node_box + 6 * (size_t)ref;
        }
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], box[k]);
            hi[k] = fmaxf(hi[k], box[3 + k]);
        }
    }
    for (int k = 0; k < 3; ++k) {
        node_box[6 * (size_t)i + k] = lo[k];
        node_box[6 * (size_t)i + 3 + k] = hi[k];
    }
    done[i] = pass;
    atomicSub(remaining, 1);
}

// Reference to a child as the traversal sees it: subtrees of at most PT_BVH_LEAF_MAX triangles are leaves
// (0x80000000 | first << 3 | count-1, sorted positions), everything else an internal node index.
__device__ __forceinline__ uint32_t traversal_ref(int ref, const int* __restrict__ range, int leaf_max = PT_BVH_LEAF_MAX) {
    if (ref < 0) return 0x80000000u | ((uint32_t)(~ref) << 3);
    const int first = range[2 * ref], count = range[2 * ref + 1] - first + 1;
    if (count <= leaf_max) return 0x80000000u | ((uint32_t)first << 3) | (uint32_t)(count - 1);
    return (uint32_t)ref;
}

// Traversal record of internal node i: both children's boxes as centre + half extents, relative to the ray origin (all
// rays share it) and padded so that a ray the exact fp32 triangle test accepts can never fail the box test through
// rounding: the slab arithmetic errs by ~1e-7 relative to the distance from the origin; the pad is 1e-5 of it.
__global__ __launch_bounds__(256) void bvh_emit_nodes(const uint32_t* __restrict__ order, const int* __restrict__ child,
                                                      const int* __restrict__ range, const float* __restrict__ tri_box,
                                                      const float* __restrict__ node_box, int n, float rox, float roy, float roz,
                                                      float extra_pad, float* __restrict__ nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const float ro[3] = {rox, roy, roz};
    float* out = nodes + (size_t)i * PT_BVH_NODE_FLOATS;
    for (int c = 0; c < 2; ++c) {
        const int ref = child[2 * i + c];
        const float* box = ref < 0 ? tri_box + 6 * (size_t)order[~ref] : node_box + 6 * (size_t)ref;
        float rel[6], far = 0.f;
        for (int k = 0; k < 3; ++k) {
            rel[k] = box[k] - ro[k];
            rel[3 + k] = box[3 + k] - ro[k];
            far = fmaxf(far, fmaxf(fabsf(rel[k]), fabsf(rel[3 + k])));
        }
        const float pad = 1.0e-5f * far + extra_pad + 1.0e-30f;
        for (int k = 0; k < 3; ++k) {
            const float lo = rel[k] - pad, hi = rel[3 + k] + pad;
            out[6 * c + k] = 0.5f * lo + 0.5f * hi;
            // the centre is rounded (half an ulp of `far` at most): the half extent takes that up again
            out[6 * c + 3 + k] = (0.5f * hi - 0.5f * lo) + 1.0e-6f * far;
        }
        out[12 + c] = __uint_as_float(traversal_ref(ref, range));
    }
    out[14] = 0.f;
    out[15] = 0.f;
}

// Triangle records and ids in sorted (leaf) order; the tail is padded with all-zero records like the unsorted array.
// The id also goes into the record's last (unused) word, so that a leaf needs no second fetch.
// drop_unreachable: records whose e2r (first word of the last quarter) is not positive can never be hit by a ray from
// the shared origin (t = e2r / det with det > 0); they are stored as zeros, which fail the test at its first comparison.
__global__ __launch_bounds__(256) void bvh_gather_records(const uint32_t* __restrict__ order, const float4* __restrict__ tris, int n,
                                                          float4* __restrict__ sorted_tris, int* __restrict__ sorted_ids, bool drop_unreachable) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n + PT_TRI_PAD) return;
    if (k < n) {
        const uint32_t id = order[k];
        for (int q = 0; q < 3; ++q) sorted_tris[4 * (size_t)k + q] = tris[4 * (size_t)id + q];
        float4 last = tris[4 * (size_t)id + 3];
        last.w = __uint_as_float(id);
        sorted_tris[4 * (size_t)k + 3] = last;
        if (drop_unreachable && !(last.x > 0.0f))
            for (int q = 0; q < 3; ++q) sorted_tris[4 * (size_t)k + q] = make_float4(0.f, 0.f, 0.f, 0.f);
        sorted_ids[k] = (int)id;
    } else {
        for (int q = 0; q < 4; ++q) sorted_tris[4 * (size_t)k + q] = make_float4(0.f, 0.f, 0.f, 0.f);
        sorted_ids[k] = 0x7fffffff;
    }
}

// Frame of the 16-bit boxes: around the root's box, wider than any padded child box can reach (a pad is at most 1.5e-5 of the
// largest coordinate), 65533 steps across so that the outward rounding below never leaves 0..65535.
__global__ void bvh_emit_qframe(const float* __restrict__ node_box, float scene_scale, float* __restrict__ qframe) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int k = 0; k < 3; ++k) {
        const float margin = 2.0e-5f * scene_scale + 1.0e-20f;
        const float base = node_box[k] - margin;
        const float extent = (node_box[3 + k] + margin) - base;
        qframe[k] = base;
        qframe[3 + k] = extent / 65533.0f;
    }
    qframe[6] = 0.f;
    qframe[7] = 0.f;
}

// ---- wide nodes --------------------------------------------------------------------------------------------------------------
// Wide node of binary node i from its list of children (binary refs). Boxes are the ones bvh_emit_nodes pads (same pads,
// absolute coordinates), rounded outwards to the frame's steps and one step further. A walk decodes a bound as
// fma(q, step / d, (base - o) / d): its rounding is some 1e-7 of the scene's size in position, a step is 1.5e-5 of it.
__global__ __launch_bounds__(256) void bvh_emit_wide(const uint32_t* __restrict__ order, const int* __restrict__ wide_child,
                                                     const int* __restrict__ range, const float* __restrict__ tri_box,
                                                     const float* __restrict__ node_box, const int* __restrict__ wide_index, int n,
                                                     float extra_pad, const float* __restrict__ qframe, uint32_t* __restrict__ nodes_w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || wide_index[i] < 0) return;
    uint32_t* out = nodes_w + (size_t)wide_index[i] * PT_BVH_WNODE_WORDS;
    const int* mine = wide_child + 4 * (size_t)wide_index[i];
    for (int slot = 0; slot < 4; ++slot) {
        const int ref = mine[slot];
        if (ref == PT_BVH_NO_WIDE_CHILD) {
            out[4 * slot] = 0u; out[4 * slot + 1] = 0u; out[4 * slot + 2] = 0u;
            out[4 * slot + 3] = PT_BVH_NO_CHILD;
            continue;
        }
        const float* box = ref < 0 ? tri_box + 6 * (size_t)order[~ref] : node_box + 6 * (size_t)ref;
        float far = 0.f;
        for (int k = 0; k < 6; ++k) far = fmaxf(far, fabsf(box[k]));
        const float pad = 1.1e-5f * far + extra_pad + 1.0e-30f;
        for (int k = 0; k < 3; ++k) {
            const float lo = (box[k] - pad) - qframe[k], hi = (box[3 + k] + pad) - qframe[k];
            const float qlo = floorf(lo / qframe[3 + k]) - 1.0f, qhi = ceilf(hi / qframe[3 + k]) + 1.0f;
            const uint32_t ulo = (uint32_t)fminf(fmaxf(qlo, 0.0f), 65535.0f), uhi = (uint32_t)fminf(fmaxf(qhi, 0.0f), 65535.0f);
            out[4 * slot + k] = ulo | (uhi << 16);
        }
        const uint32_t t = traversal_ref(ref, range, PT_BVH_WIDE_LEAF_MAX);
        out[4 * slot + 3] = (t & 0x80000000u) ? t : (uint32_t)wide_index[ref];
    }
}

// a, e1, e2 and the id of the triangle at each sorted position, 48 bytes; zero records after the last
__global__ __launch_bounds__(256) void bvh_gather_qrecords(const uint32_t* __restrict__ order, const float4* __restrict__ tris_abs, int n,
                                                           float4* __restrict__ tris_q) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n + PT_TRI_PAD) return;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
    if (k < n) {
        const uint32_t id = order[k];
        q0 = tris_abs[4 * (size_t)id];
        q1 = tris_abs[4 * (size_t)id + 1];
        q2 = tris_abs[4 * (size_t)id + 2];
        q2.y = __uint_as_float(id);
        q2.z = 0.f;
        q2.w = 0.f;
    }
    tris_q[3 * (size_t)k] = q0;
    tris_q[3 * (size_t)k + 1] = q1;
    tris_q[3 * (size_t)k + 2] = q2;
}

inline dim3 grid_for(int n) { return dim3((unsigned)((n + 255) / 256)); }

// ---- host-side builder: top-down, binned surface-area heuristic --------------------------------------------------------------
// Splits a range of triangles where the summed (area x count) of the two sides is smallest among 3 x 15 candidate planes through
// the centroid bounds; writes the same arrays the Morton-code builder writes (order, child, range; node 0 the root; the subtree
// over c triangles owns c - 1 consecutive node indices, its left subtree first). Takes far longer than the device build and makes
// hierarchies whose walks fetch fewer records.
struct HostBox {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; ++k) { lo[k] = 3.0e38f; hi[k] = -3.0e38f; } }
    void add(const float* b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b[k]); hi[k] = std::max(hi[k], b[3 + k]); } }
    void add(const HostBox& o) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], o.lo[k]); hi[k] = std::max(hi[k], o.hi[k]); } }
    float half_area() const {
        const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x < 0.f ? 0.f : (x * y + y * z) + z * x;
    }
};

struct SahBuilder {
    static constexpr int kBins = 16;
    // ranges deeper than this are halved instead of split by cost, so the tree's height stays below kMaxCostDepth + log2(n) + 1:
    // the walks keep their stacks in fixed arrays (PT_BVH_STACK entries per wave), and a mesh whose centroids cluster
    // geometrically (or whose boxes have no area, so that every plane costs the same) would otherwise give one level per triangle
    static constexpr int kMaxCostDepth = 40;
    const float* tri_box;   // n x 6

    // bin of a centroid coordinate: clamped while still a float, so a NaN or a coordinate far outside the sampled bounds lands in
    // bin 0 or the last bin on every compiler (the conversion of an out-of-range float to int is undefined); the binning pass and
    // the partition must agree, so both use this
    static int bin_of(float c, float lo, float scale) {
        const float f = (c - lo) * scale;
        if (!(f > 0.f)) return 0;
        return f >= (float)kBins ? kBins - 1 : (int)f;
    }
    uint32_t* order;
    int* child;
    int* range;

    // where the range [first, last] is cut: the first position of its right part
    int split(int first, int last, int depth) const {
        const int count = last - first + 1;
        // a large range picks its plane from every stride-th triangle (16,384 or more of them); all of them are then partitioned
        const int stride = count >= (1 << 17) ? count >> 14 : 1;
        float clo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, chi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (int p = first; p <= last; p += stride) {
            const float* b = tri_box + 6 * (size_t)order[p];
            for (int k = 0; k < 3; ++k) {
                const float c = 0.5f * b[k] + 0.5f * b[3 + k];
                clo[k] = std::min(clo[k], c);
                chi[k] = std::max(chi[k], c);
            }
        }
        int best_axis = -1, best_bin = 0;
        float best_cost = 3.0e38f, scale[3];
        if (count > 2 && depth < kMaxCostDepth) {
            HostBox bins[3][kBins];
            int counts[3][kBins];
            for (int k = 0; k < 3; ++k) {
                scale[k] = chi[k] > clo[k] ? (float)kBins / (chi[k] - clo[k]) : 0.f;
                for (int j = 0; j < kBins; ++j) { bins[k][j].reset(); counts[k][j] = 0; }
            }
            int sampled = 0;
            for (int p = first; p <= last; p += stride, ++sampled) {
                const float* b = tri_box + 6 * (size_t)order[p];
                for (int k = 0; k < 3; ++k) {
                    const float c = 0.5f * b[k] + 0.5f * b[3 + k];
                    const int j = bin_of(c, clo[k], scale[k]);
                    bins[k][j].add(b);
                    ++counts[k][j];
                }
            }
            for (int k = 0; k < 3; ++k) {
                if (scale[k] == 0.f) continue;
                float right_area[kBins];
                HostBox acc;
                acc.reset();
                for (int j = kBins - 1; j > 0; --j) {
                    if (counts[k][j]) acc.add(bins[k][j]);
                    right_area[j] = acc.half_area();
                }
                acc.reset();
                int left = 0;
                for (int j = 0; j + 1 < kBins; ++j) {
                    if (counts[k][j]) acc.add(bins[k][j]);
                    left += counts[k][j];
                    if (left == 0 || left == sampled) continue;
                    const float cost = acc.half_area() * (float)left + right_area[j + 1] * (float)(sampled - left);
                    if (cost < best_cost) { best_cost = cost; best_axis = k; best_bin = j; }
                }
            }
        }
        if (best_axis >= 0) {
            const int k = best_axis;
            uint32_t* mid = std::partition(order + first, order + last + 1, [&](uint32_t id) {
                const float* b = tri_box + 6 * (size_t)id;
                const float c = 0.5f * b[k] + 0.5f * b[3 + k];
                return bin_of(c, clo[k], scale[k]) <= best_bin;
            });
            const int at = (int)(mid - order);
            if (at > first && at <= last) return at;
        }
        // all centroids in one place (or two triangles): halves, by the widest axis of the centroids where there is one
        int k = 0;
        for (int a = 1; a < 3; ++a) if (chi[a] - clo[a] > chi[k] - clo[k]) k = a;
        const int at = first + count / 2;
        if (chi[k] > clo[k]) {
            // (compared through their bit patterns, which are totally ordered whatever the values: a mesh may hold NaNs)
            auto key = [&](uint32_t id) {
                const float* b = tri_box + 6 * (size_t)id;
                const float c = 0.5f * b[k] + 0.5f * b[3 + k];
                uint32_t u;
                memcpy(&u, &c, sizeof u);
                return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            };
            std::nth_element(order + first, order + at, order + last + 1, [&](uint32_t x, uint32_t y) {
                const uint32_t kx = key(x), ky = key(y);
                return kx < ky || (kx == ky && x < y);
            });
        }
        return at;
    }

    // the subtree over [first, last] (two triangles or more), its root at index `node`
    void build(int node, int first, int last, int depth0, std::vector<std::array<int, 4>>* defer, int defer_below) {
        struct Item { int node, first, last, depth; };
        std::vector<Item> stack;
        stack.push_back({node, first, last, depth0});
        while (!stack.empty()) {
            const Item it = stack.back();
            stack.pop_back();
            if (defer && it.last - it.first + 1 <= defer_below) {   // left to a worker thread
                defer->push_back({it.node, it.first, it.last, it.depth});
                continue;
            }
            const int at = split(it.first, it.last, it.depth);
            const int left_count = at - it.first, right_count = it.last - at + 1;
            range[2 * it.node] = it.first;
            range[2 * it.node + 1] = it.last;
            const int left_node = it.node + 1, right_node = it.node + left_count;   // (a subtree over c triangles has c - 1 nodes)
            child[2 * it.node] = left_count == 1 ? ~it.first : left_node;
            child[2 * it.node + 1] = right_count == 1 ? ~at : right_node;
            if (left_count > 1) stack.push_back({left_node, it.first, at - 1, it.depth + 1});
            if (right_count > 1) stack.push_back({right_node, at, it.last, it.depth + 1});
        }
    }
};

// Boxes bottom-up, for whatever made order / child / range (node 0 is the root).
hipError_t finish_topology(const PtBvhBuild& b, hipStream_t stream, int32_t* out_passes) {
    const int n = b.num_triangles;
    hipError_t e = hipMemsetAsync(b.done, 0, (size_t)(n - 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    int remaining = n - 1;
    e = hipMemcpyAsync(b.remaining, &remaining, sizeof(int), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    // One sweep per level of the tree. The Morton builder's trees have at most 63 + 32 levels (63-bit codes, then 32-bit
    // positions), the host builder's at most SahBuilder::kMaxCostDepth + log2(n) + 1; sweeps run in groups of 8 between checks.
    int pass = 1;
    while (remaining > 0) {
        if (pass > 4096) return hipErrorUnknown;  // cannot happen for a well-formed hierarchy; never spin forever
        for (int g = 0; g < 8; ++g, ++pass)
            hipLaunchKernelGGL(bvh_refit_pass, grid_for(n - 1), dim3(256), 0, stream, b.order, b.child, b.tri_box, n, pass + 1, b.node_box,
                               b.done, b.remaining);
        e = hipMemcpyAsync(&remaining, b.remaining, sizeof(int), hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
    }
    if (out_passes) *out_passes = pass - 1;
    return hipGetLastError();
}

}  // namespace

hipError_t pt_bvh_build_topology_sah(const PtBvhBuild& b, hipStream_t stream, int32_t* out_passes) {
    const int n = b.num_triangles;
    if (out_passes) *out_passes = 0;
    if (n < 2) return hipSuccess;
    hipLaunchKernelGGL(bvh_init_scene_box, dim3(1), dim3(64), 0, stream, b.scene_box);
    hipLaunchKernelGGL(bvh_tri_boxes, grid_for(n), dim3(256), 0, stream, b.vertices, b.indices, n, b.tri_box, b.scene_box);
    std::vector<float> tri_box;
    std::vector<uint32_t> order;
    std::vector<int> child, range;
    try {
        tri_box.resize((size_t)n * 6);
        order.resize((size_t)n);
        child.resize((size_t)(n - 1) * 2);
        range.resize((size_t)(n - 1) * 2);
    } catch (const std::bad_alloc&) {
        return hipErrorOutOfMemory;
    }
    hipError_t e = hipMemcpyAsync(tri_box.data(), b.tri_box, tri_box.size() * sizeof(float), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    for (int i = 0; i < n; ++i) order[(size_t)i] = (uint32_t)i;
    const bool report = getenv("PT_DEBUG_SCHED") != nullptr;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    const auto t_start = now();
    double ms_top = 0.0, ms_subtrees = 0.0, ms_wide = 0.0;
    SahBuilder sb{tri_box.data(), order.data(), child.data(), range.data()};
    // Work that falls into independent pieces is shared out among this thread and up to 15 more. Threads that cannot be started
    // are done without. Returns false when a piece ran out of memory.
    const int threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    auto parallel_for = [threads](size_t pieces, const std::function<void(size_t)>& piece) {
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
        auto work = [&]() {
            try {
                for (size_t k = next.fetch_add(1); k < pieces; k = next.fetch_add(1)) piece(k);
            } catch (const std::exception&) {
                failed = true;
            }
        };
        std::vector<std::thread> pool;
        try {
            pool.reserve((size_t)threads);
            for (int t = 1; t < threads && (size_t)t < pieces; ++t) pool.emplace_back(work);
        } catch (const std::exception&) {
        }
        work();
        for (auto& th : pool) th.join();
        return !failed;
    };
    // The top of the tree on this thread, down to subtrees of at most n / 64 triangles; then the subtrees, which work on disjoint
    // parts of every array (a subtree's node indices follow from its position, so the result does not depend on which thread
    // built what).
    std::vector<std::array<int, 4>> parts;   // node, first, last, depth
    try {
        sb.build(0, 0, n - 1, 0, threads > 1 && n >= 4096 ? &parts : nullptr, std::max(2, n / 64));
    } catch (const std::exception&) {
        return hipErrorOutOfMemory;
    }
    ms_top = ms_since(t_start);
    if (!parallel_for(parts.size(), [&](size_t k) { sb.build(parts[k][0], parts[k][1], parts[k][2], parts[k][3], nullptr, 0); })) return hipErrorOutOfMemory;
    ms_subtrees = ms_since(t_start) - ms_top;
    const auto t_wide = now();
    // Wide nodes: a node's two children, then the child with the largest box replaced by its own two children until there are
    // four (children that are leaves of the wide form stay). Fewer, tighter nodes than dropping every second level of the binary
    // tree: the surface-area builder's subtrees are not of equal depth. Numbered depth first: the top of the tree, then one block
    // per wide node found inside a subtree of the list above (those blocks are made side by side).
    std::vector<int> wide_index, wide_child;
    int wide_count = 0;
    try {
        const size_t ni = (size_t)(n - 1);
        std::vector<HostBox> box(ni);
        std::vector<uint8_t> in_part(ni, 0);
        auto node_box = [&](int i) {   // (children have larger indices than their parent)
            HostBox both;
            both.reset();
            for (int c = 0; c < 2; ++c) {
                const int ref = child[2 * (size_t)i + c];
                if (ref >= 0) both.add(box[(size_t)ref]);
                else both.add(tri_box.data() + 6 * (size_t)order[(size_t)~ref]);
            }
            box[(size_t)i] = both;
        };
        if (!parallel_for(parts.size(), [&](size_t k) {
                const int lo = parts[k][0], hi = parts[k][0] + (parts[k][2] - parts[k][1]) - 1;   // a subtree over c triangles: c - 1 nodes
                for (int i = hi; i >= lo; --i) { node_box(i); in_part[(size_t)i] = 1; }
            }))
            return hipErrorOutOfMemory;
        for (int i = n - 2; i >= 0; --i)
            if (!in_part[(size_t)i]) node_box(i);
        auto is_leaf = [&](int ref) { return ref < 0 || range[2 * (size_t)ref + 1] - range[2 * (size_t)ref] + 1 <= PT_BVH_WIDE_LEAF_MAX; };
        // children of the wide node rooted at binary node i -> slots; returns how many
        auto expand = [&](int i, int* slots) {
            slots[0] = child[2 * (size_t)i]; slots[1] = child[2 * (size_t)i + 1];
            slots[2] = slots[3] = PT_BVH_NO_WIDE_CHILD;
            int used = 2;
            while (used < 4) {
                int pick = -1;
                float pick_area = -1.0f;
                for (int k = 0; k < used; ++k)
                    if (!is_leaf(slots[k])) {
                        const float a = box[(size_t)slots[k]].half_area();
                        if (a > pick_area) { pick_area = a; pick = k; }
                    }
                if (pick < 0) break;
                const int ref = slots[pick];
                slots[pick] = child[2 * (size_t)ref];
                slots[used++] = child[2 * (size_t)ref + 1];
            }
            return used;
        };
        wide_index.assign(ni, -1);
        wide_child.assign(ni * 4, PT_BVH_NO_WIDE_CHILD);
        std::vector<int> roots, deferred;
        if (!is_leaf(0)) roots.push_back(0);
        while (!roots.empty()) {
            const int i = roots.back();
            roots.pop_back();
            if (in_part[(size_t)i]) { deferred.push_back(i); continue; }
            int slots[4];
            const int used = expand(i, slots);
            wide_index[(size_t)i] = wide_count;
            for (int k = 0; k < 4; ++k) wide_child[4 * (size_t)wide_count + k] = slots[k];
            ++wide_count;
            for (int k = 0; k < used; ++k)
                if (!is_leaf(slots[k])) roots.push_back(slots[k]);
        }
        struct Block { std::vector<int> nodes, children; };
        std::vector<Block> blocks(deferred.size());
        if (!parallel_for(deferred.size(), [&](size_t d) {
                Block& blk = blocks[d];
                std::vector<int> todo{deferred[d]};
                while (!todo.empty()) {
                    const int i = todo.back();
                    todo.pop_back();
                    int slots[4];
                    const int used = expand(i, slots);
                    blk.nodes.push_back(i);
                    blk.children.insert(blk.children.end(), slots, slots + 4);
                    for (int k = 0; k < used; ++k)
                        if (!is_leaf(slots[k])) todo.push_back(slots[k]);
                }
            }))
            return hipErrorOutOfMemory;
        std::vector<int> base(deferred.size() + 1, wide_count);
        for (size_t d = 0; d < deferred.size(); ++d) base[d + 1] = base[d] + (int)blocks[d].nodes.size();
        wide_count = base[deferred.size()];
        if (!parallel_for(deferred.size(), [&](size_t d) {
                const Block& blk = blocks[d];
                for (size_t q = 0; q < blk.nodes.size(); ++q) {
                    wide_index[(size_t)blk.nodes[q]] = base[d] + (int)q;
                    for (int k = 0; k < 4; ++k) wide_child[4 * ((size_t)base[d] + q) + k] = blk.children[4 * q + k];
                }
            }))
            return hipErrorOutOfMemory;
    } catch (const std::exception&) {
        return hipErrorOutOfMemory;
    }
    ms_wide = ms_since(t_wide);
    const auto t_upload = now();
    e = hipMemcpyAsync(b.order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(b.child, child.data(), child.size() * sizeof(int), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(b.range, range.data(), range.size() * sizeof(int), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(b.wide_index, wide_index.data(), wide_index.size() * sizeof(int), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(b.wide_child, wide_child.data(), wide_child.size() * sizeof(int), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);   // the host arrays go out of scope
    if (e != hipSuccess) return e;
    if (report)
        fprintf(stderr, "[pt bvh] host builder, %d triangles: top of the tree %.1f ms, subtrees %.1f ms, wide nodes %.1f ms, upload %.1f ms\n", n,
                ms_top, ms_subtrees, ms_wide, ms_since(t_upload));
    return finish_topology(b, stream, out_passes);
}

size_t pt_bvh_sort_temp_bytes(int32_t n) {
    size_t bytes = 0, scan = 0, scan64 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                             (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 63, nullptr);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan, (const int*)nullptr, (int*)nullptr, n, nullptr);  // numbering of the wide nodes
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan64, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, n, nullptr);  // cluster merging
    return std::max(bytes, std::max(scan, scan64));
}

namespace {

hipError_t morton_sorted_order(const PtBvhBuild& b, hipStream_t stream) {
    const int n = b.num_triangles;
    hipLaunchKernelGGL(bvh_init_scene_box, dim3(1), dim3(64), 0, stream, b.scene_box);
    hipLaunchKernelGGL(bvh_tri_boxes, grid_for(n), dim3(256), 0, stream, b.vertices, b.indices, n, b.tri_box, b.scene_box);
    hipLaunchKernelGGL(bvh_morton_keys, grid_for(n), dim3(256), 0, stream, b.tri_box, b.scene_box, n, b.keys_tmp, b.order_tmp);
    size_t bytes = b.sort_tmp_bytes;
    return hipcub::DeviceRadixSort::SortPairs(b.sort_tmp, bytes, b.keys_tmp, b.keys, b.order_tmp, b.order, n, 0, 63, stream);
}

// ---- wide nodes by box size, on the device -----------------------------------------------------------------------------------
// A wide node's children: the two children of its binary node, then, while fewer than four, the child with the largest box that
// is not a leaf of the wide form replaced by its own two children (the rule the host builder applies). Level by level from the
// root: the nodes of one level are expanded side by side, their internal children numbered by a prefix sum and queued as the
// next level, so the numbering (breadth first) does not depend on the order threads run in.
__device__ __forceinline__ bool wide_leaf(int ref, const int* __restrict__ range) {
    return ref < 0 || range[2 * ref + 1] - range[2 * ref] + 1 <= PT_BVH_WIDE_LEAF_MAX;
}
__device__ __forceinline__ float box_half_area(const float* __restrict__ box) {
    const float x = box[3] - box[0], y = box[4] - box[1], z = box[5] - box[2];
    return (x * y + y * z) + z * x;
}

__global__ __launch_bounds__(256) void bvh_wide_expand(const int* __restrict__ queue, int count, int first_index, const int* __restrict__ child,
                                                       const int* __restrict__ range, const float* __restrict__ node_box, int* __restrict__ wide_index,
                                                       int* __restrict__ wide_child, int* __restrict__ internal_children) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= count) return;
    const int i = queue[q], w = first_index + q;
    int slots[4] = {child[2 * i], child[2 * i + 1], PT_BVH_NO_WIDE_CHILD, PT_BVH_NO_WIDE_CHILD};
    int used = 2;
    while (used < 4) {
        int pick = -1;
        float pick_area = -1.0f;
        for (int k = 0; k < used; ++k)
            if (!wide_leaf(slots[k], range)) {
                const float a = box_half_area(node_box + 6 * (size_t)slots[k]);
                if (a > pick_area || pick < 0) { pick_area = a; pick = k; }   // (a NaN area never wins over an earlier pick)
            }
        if (pick < 0) break;
        const int ref = slots[pick];
        slots[pick] = child[2 * ref];
        slots[used++] = child[2 * ref + 1];
    }
    wide_index[i] = w;
    int internal = 0;
    for (int k = 0; k < 4; ++k) {
        wide_child[4 * (size_t)w + k] = slots[k];
        internal += (k < used && !wide_leaf(slots[k], range)) ? 1 : 0;
    }
    internal_children[q] = internal;
}

__global__ __launch_bounds__(256) void bvh_wide_queue_next(int count, int first_index, const int* __restrict__ range, const int* __restrict__ wide_child,
                                                           const int* __restrict__ offsets, int* __restrict__ next_queue) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= count) return;
    int at = offsets[q];
    for (int k = 0; k < 4; ++k) {
        const int ref = wide_child[4 * (size_t)(first_index + q) + k];
        if (ref != PT_BVH_NO_WIDE_CHILD && !wide_leaf(ref, range)) next_queue[at++] = ref;
    }
}

// queue_a / queue_b / counts / offsets: n - 1 ints each (the builder's scratch arrays)
hipError_t collapse_wide_on_device(const PtBvhBuild& b, int* queue_a, int* queue_b, int* counts, int* offsets, hipStream_t stream) {
    const int n = b.num_triangles;
    hipError_t e = hipMemsetAsync(b.wide_index, 0xff, (size_t)(n - 1) * sizeof(int), stream);
    if (e != hipSuccess) return e;
    const int root = 0;
    e = hipMemcpyAsync(queue_a, &root, sizeof(int), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);   // (`root` leaves scope)
    if (e != hipSuccess) return e;
    int count = 1, first_index = 0;
    for (int level = 0; count > 0; ++level) {
        if (level > 4096) return hipErrorUnknown;
        hipLaunchKernelGGL(bvh_wide_expand, grid_for(count), dim3(256), 0, stream, queue_a, count, first_index, b.child, b.range, b.node_box, b.wide_index,
                           b.wide_child, counts);
        size_t bytes = b.sort_tmp_bytes;
        e = hipcub::DeviceScan::ExclusiveSum(b.sort_tmp, bytes, counts, offsets, count, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bvh_wide_queue_next, grid_for(count), dim3(256), 0, stream, count, first_index, b.range, b.wide_child, offsets, queue_b);
        int last[2] = {0, 0};
        e = hipMemcpyAsync(&last[0], offsets + (count - 1), sizeof(int), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&last[1], counts + (count - 1), sizeof(int), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        first_index += count;
        count = last[0] + last[1];
        std::swap(queue_a, queue_b);
    }
    return hipGetLastError();
}

// ---- device builder: clusters merged with their nearest neighbours along the Morton curve ------------------------------------
// (Parallel locally-ordered clustering, Meister and Bittner 2018.) The triangles start as one cluster each, in Morton order. In
// every round a cluster looks `kPlocRadius` places to either side for the neighbour with which it would have the smallest box;
// two clusters that choose each other merge into a node; the clusters are compacted, order kept. Merging what is close in SPACE,
// whatever the Morton code says about it, gives hierarchies close to a surface-area-heuristic build's (measured in materials
// mode: the walks fetch -1 % to +9 % more records than over the host builder's, against 8 % to 21 % more over the Morton-code
// tree), at the price of subtrees whose leaves are no longer neighbours in Morton order: when the tree stands, the triangles
// are re-ordered depth first, so that every subtree is a contiguous range again. A wider search (32 to 128 places) changes the
// records fetched by under 1 % and costs build time, so the radius stays small.
constexpr int kPlocRadius = 16;
constexpr int kPlocSearchAllBelow = 256;

struct PlocCluster {
    float lo[3];
    int ref;      // >= 0: node (creation order); < 0: ~(sorted position of a triangle)
    float hi[3];
    int count;    // triangles below
};
static_assert(sizeof(PlocCluster) == 32, "two 16-byte halves");

__global__ __launch_bounds__(256) void ploc_init(const uint32_t* __restrict__ order, const float* __restrict__ tri_box, int n, PlocCluster* __restrict__ c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* box = tri_box + 6 * (size_t)order[i];
    PlocCluster k;
    for (int a = 0; a < 3; ++a) { k.lo[a] = box[a]; k.hi[a] = box[3 + a]; }
    k.ref = ~i;
    k.count = 1;
    c[i] = k;
}

__device__ __forceinline__ float merged_half_area(const PlocCluster& a, const PlocCluster& b) {
    const float x = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]), y = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]),
                z = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
    return (x * y + y * z) + z * x;
}

__global__ __launch_bounds__(256) void ploc_nearest(const PlocCluster* __restrict__ c, int count, int radius, int* __restrict__ nn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const PlocCluster me = c[i];
    int best = i + 1 < count ? i + 1 : i - 1;   // (a cluster whose every candidate has a NaN box still names a neighbour)
    float best_area = 3.0e38f;
    const float mine = merged_half_area(me, me);
    const int lo = max(0, i - radius), hi = min(count - 1, i + radius);
    for (int j = lo; j <= hi; ++j) {
        if (j == i) continue;
        // what the merge adds: the merged box's area less three quarters of the two it replaces. (Measured on four meshes against
        // the plain merged area, the full difference and the difference to the larger one: records fetched per query 3 % fewer
        // than with the plain area on average; the full difference builds chains, 25 % to 90 % more.)
        const PlocCluster other = c[j];
        const float a = merged_half_area(me, other) - 0.75f * (mine + merged_half_area(other, other));
        if (a < best_area) { best_area = a; best = j; }   // (equal boxes: the lower index)
    }
    nn[i] = best;
}

// low half: 1 = the cluster stays in the list (alone, or as the merged one); high half: 1 = it merges with nn[i] (the lower of the two)
__global__ __launch_bounds__(256) void ploc_flags(const int* __restrict__ nn, int count, unsigned long long* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int j = nn[i];
    const bool mutual = nn[j] == i;
    flags[i] = (unsigned long long)((mutual && j < i) ? 0u : 1u) | ((unsigned long long)((mutual && i < j) ? 1u : 0u) << 32);
}

__global__ __launch_bounds__(256) void ploc_apply(const PlocCluster* __restrict__ in, const int* __restrict__ nn, const unsigned long long* __restrict__ flags,
                                                  const unsigned long long* __restrict__ scan, int count, int first_node, PlocCluster* __restrict__ out,
                                                  int* __restrict__ child, int* __restrict__ left_count, int* __restrict__ node_count,
                                                  int* __restrict__ parent, int* __restrict__ leaf_parent, int* __restrict__ totals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const unsigned long long f = flags[i], s = scan[i];
    if (i == count - 1) {
        totals[0] = (int)(uint32_t)(s + f);            // clusters that stay
        totals[1] = (int)(uint32_t)((s + f) >> 32);    // nodes made
    }
    if (!(uint32_t)f) return;
    PlocCluster k = in[i];
    if ((uint32_t)(f >> 32)) {
        const PlocCluster o = in[nn[i]];
        const int node = first_node + (int)(uint32_t)(s >> 32);
        child[2 * node] = k.ref;
        child[2 * node + 1] = o.ref;
        left_count[node] = k.count;
        node_count[node] = k.count + o.count;
        // parent and side of the two children: node << 1 | (1 = right child)
        if (k.ref < 0) leaf_parent[~k.ref] = node << 1; else parent[k.ref] = node << 1;
        if (o.ref < 0) leaf_parent[~o.ref] = node << 1 | 1; else parent[o.ref] = node << 1 | 1;
        for (int a = 0; a < 3; ++a) { k.lo[a] = fminf(k.lo[a], o.lo[a]); k.hi[a] = fmaxf(k.hi[a], o.hi[a]); }
        k.ref = node;
        k.count += o.count;
    }
    out[(uint32_t)s] = k;
}

// where the subtree below `up` (parent << 1 | side of the thing we start from) begins in depth-first leaf order
__device__ __forceinline__ int ploc_first_leaf(int up, const int* __restrict__ parent, const int* __restrict__ left_count, int root) {
    int first = 0;
    for (;;) {
        const int node = up >> 1;
        if (up & 1) first += left_count[node];
        if (node == root) return first;
        up = parent[node];
    }
}

__global__ __launch_bounds__(256) void ploc_leaf_positions(int n, const int* __restrict__ leaf_parent, const int* __restrict__ parent,
                                                           const int* __restrict__ left_count, const uint32_t* __restrict__ order,
                                                           int* __restrict__ new_pos, uint32_t* __restrict__ new_order) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int at = ploc_first_leaf(leaf_parent[p], parent, left_count, n - 2);
    new_pos[p] = at;
    new_order[at] = order[p];
}

// nodes in their final numbering (the root, made last, becomes node 0; children get larger indices than their parent)
__global__ __launch_bounds__(256) void ploc_write_nodes(int n, const int* __restrict__ child_c, const int* __restrict__ parent, const int* __restrict__ left_count,
                                                        const int* __restrict__ node_count, const int* __restrict__ new_pos, int* __restrict__ child,
                                                        int* __restrict__ range) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n - 1) return;
    const int id = n - 2 - k;
    for (int c = 0; c < 2; ++c) {
        const int ref = child_c[2 * k + c];
        child[2 * id + c] = ref < 0 ? ~new_pos[~ref] : n - 2 - ref;
    }
    const int first = k == n - 2 ? 0 : ploc_first_leaf(parent[k], parent, left_count, n - 2);
    range[2 * id] = first;
    range[2 * id + 1] = first + node_count[k] - 1;
}

struct DeviceScratch {   // freed when the builder returns, whatever happened
    std::vector<void*> ptrs;
    ~DeviceScratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T>
    hipError_t get(T** out, size_t count) {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = static_cast<T*>(p);
        return e;
    }
};

}  // namespace

hipError_t pt_bvh_build_topology(const PtBvhBuild& b, hipStream_t stream, int32_t* out_passes) {
    const int n = b.num_triangles;
    if (out_passes) *out_passes = 0;
    if (n < 2) return hipSuccess;
    hipError_t e = morton_sorted_order(b, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bvh_hierarchy, grid_for(n - 1), dim3(256), 0, stream, b.keys, n, b.child, b.range);
    e = finish_topology(b, stream, out_passes);
    if (e != hipSuccess) return e;
    DeviceScratch scratch;
    int* q[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < 4; ++k)
        if ((e = scratch.get(&q[k], (size_t)(n - 1))) != hipSuccess) return e;
    e = collapse_wide_on_device(b, q[0], q[1], q[2], q[3], stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(stream);
}

hipError_t pt_bvh_build_topology_ploc(const PtBvhBuild& b, hipStream_t stream, int32_t* out_passes) {
    const int n = b.num_triangles;
    if (out_passes) *out_passes = 0;
    if (n < 2) return hipSuccess;
    hipError_t e = morton_sorted_order(b, stream);
    if (e != hipSuccess) return e;
    const bool report = getenv("PT_DEBUG_SCHED") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    DeviceScratch scratch;
    PlocCluster *ca = nullptr, *cb = nullptr;
    int *nn = nullptr, *child_c = nullptr, *left_count = nullptr, *node_count = nullptr, *parent = nullptr, *leaf_parent = nullptr, *new_pos = nullptr, *totals = nullptr;
    unsigned long long *flags = nullptr, *scan = nullptr;
    const size_t nn_ = (size_t)n, ni = (size_t)(n - 1);
    if ((e = scratch.get(&ca, nn_)) != hipSuccess || (e = scratch.get(&cb, nn_)) != hipSuccess || (e = scratch.get(&nn, nn_)) != hipSuccess ||
        (e = scratch.get(&flags, nn_)) != hipSuccess || (e = scratch.get(&scan, nn_)) != hipSuccess || (e = scratch.get(&child_c, 2 * ni)) != hipSuccess ||
        (e = scratch.get(&left_count, ni)) != hipSuccess || (e = scratch.get(&node_count, ni)) != hipSuccess || (e = scratch.get(&parent, ni)) != hipSuccess ||
        (e = scratch.get(&leaf_parent, nn_)) != hipSuccess || (e = scratch.get(&new_pos, nn_)) != hipSuccess || (e = scratch.get(&totals, 2)) != hipSuccess)
        return e;
    hipLaunchKernelGGL(ploc_init, grid_for(n), dim3(256), 0, stream, b.order, b.tri_box, n, ca);
    int count = n, nodes = 0, rounds = 0;
    bool gave_up = false;
    while (count > 1) {
        // few clusters left: every cluster looks at all the others (the sorted order no longer limits the choice, and the
        // search costs count * count, which is small here)
        hipLaunchKernelGGL(ploc_nearest, grid_for(count), dim3(256), 0, stream, ca, count, count <= kPlocSearchAllBelow ? count : kPlocRadius, nn);
        hipLaunchKernelGGL(ploc_flags, grid_for(count), dim3(256), 0, stream, nn, count, flags);
        size_t bytes = b.sort_tmp_bytes;
        e = hipcub::DeviceScan::ExclusiveSum(b.sort_tmp, bytes, flags, scan, count, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(ploc_apply, grid_for(count), dim3(256), 0, stream, ca, nn, flags, scan, count, nodes, cb, child_c, left_count, node_count, parent,
                           leaf_parent, totals);
        int h[2] = {0, 0};
        e = hipMemcpyAsync(h, totals, sizeof h, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        // Every round merges at least the closest pair. Layouts on which little more than that happens (cluster sizes in
        // geometric progression, boxes that are not numbers) would take as many rounds as there are triangles and end in a chain
        // deeper than the walks' stacks: those meshes get the Morton-code hierarchy, whose height is bounded.
        if (h[1] < 1 || (count > 4096 && h[1] < count / 512) || ++rounds > 256) { gave_up = true; break; }
        nodes += h[1];
        count = h[0];
        std::swap(ca, cb);
    }
    if (!gave_up && nodes != n - 1) return hipErrorUnknown;
    if (!gave_up) {
        hipLaunchKernelGGL(ploc_leaf_positions, grid_for(n), dim3(256), 0, stream, n, leaf_parent, parent, left_count, b.order, new_pos, b.order_tmp);
        hipLaunchKernelGGL(ploc_write_nodes, grid_for(n - 1), dim3(256), 0, stream, n, child_c, parent, left_count, node_count, new_pos, b.child, b.range);
        e = hipMemcpyAsync(b.order, b.order_tmp, nn_ * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
        e = finish_topology(b, stream, out_passes);
        if (e != hipSuccess) return e;
        // (the per-wave walk keeps PT_BVH_STACK entries: a hierarchy that could fill them is not used)
        if (out_passes && *out_passes > PT_BVH_STACK - 16) gave_up = true;
    }
    if (gave_up) {
        if (report) fprintf(stderr, "[pt bvh] cluster merging given up after %d rounds (%d clusters left): Morton-code hierarchy instead\n", rounds, count);
        e = morton_sorted_order(b, stream);   // (the depth-first order may already have replaced the Morton order)
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bvh_hierarchy, grid_for(n - 1), dim3(256), 0, stream, b.keys, n, b.child, b.range);
        e = finish_topology(b, stream, out_passes);
        if (e != hipSuccess) return e;
    }
    // wide nodes (scratch: four arrays of n - 1 ints, taken from what the merging no longer needs)
    e = collapse_wide_on_device(b, nn, new_pos, left_count, node_count, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (report)
        fprintf(stderr, "[pt bvh] device builder, %d triangles: %d merging rounds, %d sweeps, %.2f ms with the wide nodes\n", n, rounds,
                out_passes ? *out_passes : 0, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    return e;
}

namespace {

// ---- device builder: top-down, binned surface-area heuristic -------------------------------------------------------------------
// SahBuilder::split's rules, level by level on the device. Ranges of more than kDevSmall triangles are split one level per round
// of kernels over all positions: bounds of the centroids and the 3 x 16 bins of every range by atomics (a wave first gathers what
// its 64 positions add to a range in registers), the plane per range, a flag per triangle, a prefix sum of the flags, a stable
// partition into the other order array. Ranges of at most kDevSmall triangles are then finished by one wave each, which keeps the
// triangles in LDS and splits until single triangles are left. Node indices are the host builder's (the subtree over c triangles
// owns c - 1 consecutive indices, left subtree first), so parents can name their children before those exist.
constexpr int kDevBins = SahBuilder::kBins;
constexpr int kDevSmall = 64;
constexpr int kDevPairs = 3 * kDevBins;   // (axis, bin) pairs of a range
static_assert(kDevBins == 16 && kDevPairs <= 64, "a wave's lanes take one (axis, bin) pair each");

__device__ __forceinline__ int dev_bin_of(float c, float lo, float scale) {   // SahBuilder::bin_of
    const float f = (c - lo) * scale;
    if (!(f > 0.f)) return 0;
    return f >= (float)kDevBins ? kDevBins - 1 : (int)f;
}
__device__ __forceinline__ float dev_half_area(const float* lo, const float* hi) {   // HostBox::half_area
    const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return x < 0.f ? 0.f : (x * y + y * z) + z * x;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    for (int off = 32; off; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    for (int off = 32; off; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    for (int off = 32; off; off >>= 1) { const unsigned long long o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
    return v;
}
// how many triangles of a range take part in choosing its plane: every stride-th (SahBuilder::split)
__device__ __forceinline__ int dev_sample_stride(int count) { return count >= (1 << 17) ? count >> 14 : 1; }

__global__ __launch_bounds__(256) void sahd_iota_ranges(int n, int* __restrict__ seg_of, int value) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) seg_of[p] = value;
}

__global__ __launch_bounds__(256) void sahd_reset(int ranges, uint32_t* __restrict__ cb, uint32_t* __restrict__ cnt, uint32_t* __restrict__ box) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)ranges * 6) cb[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
    if (i < (size_t)ranges * kDevPairs) cnt[i] = 0u;
    if (i < (size_t)ranges * kDevPairs * 6) box[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
}

// what position p adds to its range's statistics: the range, whether p is one of its samples, the triangle's box and centroid
struct DevSample {
    int range;
    bool on;
    float box[6], c[3];
};
__device__ __forceinline__ DevSample sahd_sample(int p, int n, const uint32_t* __restrict__ order, const float* __restrict__ tri_box,
                                                 const int* __restrict__ seg_of, const int* __restrict__ seg_first, const int* __restrict__ seg_last) {
    DevSample s;
    s.range = p < n ? seg_of[p] : -1;
    s.on = false;
    for (int k = 0; k < 6; ++k) s.box[k] = 0.f;
    for (int k = 0; k < 3; ++k) s.c[k] = 0.f;
    if (s.range >= 0) {
        const int first = seg_first[s.range], count = seg_last[s.range] - first + 1;
        s.on = (p - first) % dev_sample_stride(count) == 0;
        if (s.on) {
            const float* b = tri_box + 6 * (size_t)order[p];
            for (int k = 0; k < 6; ++k) s.box[k] = b[k];
            for (int k = 0; k < 3; ++k) s.c[k] = 0.5f * b[k] + 0.5f * b[3 + k];
        }
    }
    return s;
}

// bounds of the sampled centroids of every range (NaN coordinates take no part, as with std::min / std::max on the host)
__global__ __launch_bounds__(256) void sahd_bounds(const uint32_t* __restrict__ order, const float* __restrict__ tri_box, const int* __restrict__ seg_of,
                                                   const int* __restrict__ seg_first, const int* __restrict__ seg_last, int n, uint32_t* __restrict__ cb) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const DevSample s = sahd_sample(p, n, order, tri_box, seg_of, seg_first, seg_last);
    // one range at a time (the 64 consecutive positions of a wave touch at most two ranges of more than 64 triangles)
    unsigned long long todo = __builtin_amdgcn_ballot_w64(s.on);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int r0 = __shfl(s.range, leader, 64);
        const bool mine = s.on && s.range == r0;
        for (int k = 0; k < 3; ++k) {
            const bool number = mine && s.c[k] == s.c[k];
            const uint32_t lo = wave_min_u32(number ? f2ord(s.c[k]) : 0xffffffffu), hi = wave_max_u32(number ? f2ord(s.c[k]) : 0u);
            if (lane == leader) {
                atomic_min_if_lower(&cb[6 * (size_t)r0 + k], lo);
                atomic_max_if_higher(&cb[6 * (size_t)r0 + 3 + k], hi);
            }
        }
        todo &= ~__builtin_amdgcn_ballot_w64(mine);
    }
}

// the 3 x 16 bins of every range: count and box of the sampled triangles whose centroid falls into the bin. Lane t < 48 of a
// wave gathers pair t = (axis, bin) over the wave's 64 positions in registers and adds it to the range's bins when the range
// changes and at the end.
__global__ __launch_bounds__(256) void sahd_bins(const uint32_t* __restrict__ order, const float* __restrict__ tri_box, const int* __restrict__ seg_of,
                                                 const int* __restrict__ seg_first, const int* __restrict__ seg_last, int n,
                                                 const uint32_t* __restrict__ cb, uint32_t* __restrict__ cnt, uint32_t* __restrict__ box) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const DevSample s = sahd_sample(p, n, order, tri_box, seg_of, seg_first, seg_last);
    int bin[3] = {0, 0, 0};
    if (s.on)
        for (int k = 0; k < 3; ++k) {
            const float clo = ord2f(cb[6 * (size_t)s.range + k]), chi = ord2f(cb[6 * (size_t)s.range + 3 + k]);
            const float scale = chi > clo ? (float)kDevBins / (chi - clo) : 0.f;
            bin[k] = dev_bin_of(s.c[k], clo, scale);
        }
    const unsigned long long on = __builtin_amdgcn_ballot_w64(s.on);
    if (!on) return;
    const int my_axis = lane >> 4, my_bin = lane & 15;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    uint32_t count = 0;
    int current = -1;
    auto flush = [&]() {
        if (current >= 0 && lane < kDevPairs && count) {
            const size_t at = (size_t)current * kDevPairs + lane;
            atomicAdd(&cnt[at], count);
            for (int k = 0; k < 3; ++k) {
                atomic_min_if_lower(&box[6 * at + k], f2ord(lo[k]));
                atomic_max_if_higher(&box[6 * at + 3 + k], f2ord(hi[k]));
            }
        }
        for (int k = 0; k < 3; ++k) { lo[k] = 3.0e38f; hi[k] = -3.0e38f; }
        count = 0;
    };
    for (int i = 0; i < 64; ++i) {
        if (!((on >> i) & 1ull)) continue;   // (uniform)
        const int r = __builtin_amdgcn_readlane(s.range, i);
        if (r != current) { flush(); current = r; }
        const int b0 = __builtin_amdgcn_readlane(bin[0], i), b1 = __builtin_amdgcn_readlane(bin[1], i), b2 = __builtin_amdgcn_readlane(bin[2], i);
        float tb[6];
        for (int k = 0; k < 6; ++k) tb[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.box[k]), i));
        const int its = my_axis == 0 ? b0 : (my_axis == 1 ? b1 : b2);
        if (lane < kDevPairs && its == my_bin) {
            for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], tb[k]); hi[k] = fmaxf(hi[k], tb[3 + k]); }   // (a NaN bound takes no part)
            ++count;
        }
    }
    flush();
}

// cost of the plane after bin `bin` of `axis` from the 16 bins of that axis (count, lo, hi each); +inf where one side is empty
__device__ __forceinline__ float sahd_plane_cost(const int* __restrict__ count, const float (*lo)[3], const float (*hi)[3], int axis, int bin) {
    float llo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, lhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, rlo[3] = {3.0e38f, 3.0e38f, 3.0e38f},
          rhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    int left = 0, right = 0;
    for (int j = 0; j < kDevBins; ++j) {
        const int t = axis * kDevBins + j, c = count[t];
        if (!c) continue;
        if (j <= bin) {
            left += c;
            for (int k = 0; k < 3; ++k) { llo[k] = fminf(llo[k], lo[t][k]); lhi[k] = fmaxf(lhi[k], hi[t][k]); }
        } else {
            right += c;
            for (int k = 0; k < 3; ++k) { rlo[k] = fminf(rlo[k], lo[t][k]); rhi[k] = fmaxf(rhi[k], hi[t][k]); }
        }
    }
    if (left == 0 || right == 0) return __builtin_inff();
    return dev_half_area(llo, lhi) * (float)left + dev_half_area(rlo, rhi) * (float)right;
}
// the cheapest of a wave's candidate planes (lane t: cost of pair t, +inf for none): the lowest t among equal costs, as the host's
// loops with their strict '<'; -1 when no lane has a finite cost
__device__ __forceinline__ int sahd_cheapest(float cost, int t) {
    const uint32_t bits = __float_as_uint(cost);   // costs are sums of non-negative products: their bit patterns order like the values
    const unsigned long long key = bits < 0x7f800000u ? ((unsigned long long)bits << 32) | (uint32_t)t : ~0ull;
    const unsigned long long best = wave_min_u64(key);
    return best == ~0ull ? -1 : (int)(uint32_t)best;
}

// one wave per range: the plane
__global__ __launch_bounds__(64) void sahd_choose(const uint32_t* __restrict__ cb, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ box,
                                                  int* __restrict__ axis, int* __restrict__ bin, float* __restrict__ plane_lo, float* __restrict__ plane_scale) {
    __shared__ int count[kDevPairs];
    __shared__ float lo[kDevPairs][3], hi[kDevPairs][3];
    const int r = blockIdx.x, t = threadIdx.x;
    if (t < kDevPairs) {
        const size_t at = (size_t)r * kDevPairs + t;
        count[t] = (int)cnt[at];
        for (int k = 0; k < 3; ++k) { lo[t][k] = ord2f(box[6 * at + k]); hi[t][k] = ord2f(box[6 * at + 3 + k]); }
    }
    __syncthreads();
    float cost = __builtin_inff(), clo = 0.f, scale = 0.f;
    if (t < kDevPairs) {
        const int k = t >> 4;
        clo = ord2f(cb[6 * (size_t)r + k]);
        const float chi = ord2f(cb[6 * (size_t)r + 3 + k]);
        scale = chi > clo ? (float)kDevBins / (chi - clo) : 0.f;
        if (scale != 0.f && (t & 15) < kDevBins - 1) cost = sahd_plane_cost(count, lo, hi, k, t & 15);
    }
    const int best = sahd_cheapest(cost, t);
    if (best < 0) {
        if (t == 0) axis[r] = -1;
    } else if (t == best) {
        axis[r] = t >> 4;
        bin[r] = t & 15;
        plane_lo[r] = clo;
        plane_scale[r] = scale;
    }
}

__global__ __launch_bounds__(256) void sahd_flags(const uint32_t* __restrict__ order, const float* __restrict__ tri_box, const int* __restrict__ seg_of, int n,
                                                  const int* __restrict__ axis, const int* __restrict__ bin, const float* __restrict__ plane_lo,
                                                  const float* __restrict__ plane_scale, int cost_driven, int* __restrict__ flags) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int r = seg_of[p];
    int left = 0;
    if (r >= 0 && cost_driven && axis[r] >= 0) {
        const int k = axis[r];
        const float* b = tri_box + 6 * (size_t)order[p];
        left = dev_bin_of(0.5f * b[k] + 0.5f * b[3 + k], plane_lo[r], plane_scale[r]) <= bin[r] ? 1 : 0;
    }
    flags[p] = left;
}

// per range: where it is cut, its node, and its two parts as ranges of the next level (more than kDevSmall triangles), as work for
// sahd_small (2 to kDevSmall) or as a leaf. counters[0] / [1]: ranges of the next level / small ranges so far.
__global__ __launch_bounds__(256) void sahd_split(int ranges, const int* __restrict__ seg_first, const int* __restrict__ seg_last, const int* __restrict__ seg_node,
                                                  const int* __restrict__ flags, const int* __restrict__ scan, int cost_driven, int depth, int* __restrict__ axis,
                                                  int* __restrict__ cut, int* __restrict__ part_range, int* __restrict__ child, int* __restrict__ range,
                                                  int* __restrict__ next_first, int* __restrict__ next_last, int* __restrict__ next_node, int* __restrict__ small_first,
                                                  int* __restrict__ small_last, int* __restrict__ small_node, int* __restrict__ small_depth, int* __restrict__ counters) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ranges) return;
    const int first = seg_first[r], last = seg_last[r], node = seg_node[r], count = last - first + 1;
    int a = cost_driven ? axis[r] : -1, at = 0;
    if (a >= 0) {
        const int left = scan[last] + flags[last] - scan[first];
        at = first + left;
        if (left <= 0 || at > last) a = -1;
    }
    if (a < 0) at = first + count / 2;   // halves, in the order the triangles stand in (Morton order within the range)
    axis[r] = a;
    cut[r] = at;
    const int part_first[2] = {first, at}, part_last[2] = {at - 1, last}, part_node[2] = {node + 1, node + (at - first)};
    range[2 * (size_t)node] = first;
    range[2 * (size_t)node + 1] = last;
    for (int side = 0; side < 2; ++side) {
        const int c = part_last[side] - part_first[side] + 1;
        child[2 * (size_t)node + side] = c == 1 ? ~part_first[side] : part_node[side];
        int next = -1;
        if (c > kDevSmall) {
            next = atomicAdd(&counters[0], 1);
            next_first[next] = part_first[side]; next_last[next] = part_last[side]; next_node[next] = part_node[side];
        } else if (c >= 2) {
            const int q = atomicAdd(&counters[1], 1);
            small_first[q] = part_first[side]; small_last[q] = part_last[side]; small_node[q] = part_node[side]; small_depth[q] = depth + 1;
        }
        part_range[2 * (size_t)r + side] = next;
    }
}

// the stable partition: every triangle to its place in the other order array, and the range of the next level it is in (-1: none)
__global__ __launch_bounds__(256) void sahd_scatter(const uint32_t* __restrict__ order_in, const int* __restrict__ seg_of, int n, const int* __restrict__ seg_first,
                                                    const int* __restrict__ axis, const int* __restrict__ cut, const int* __restrict__ part_range,
                                                    const int* __restrict__ flags, const int* __restrict__ scan, uint32_t* __restrict__ order_out,
                                                    int* __restrict__ seg_of_out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int r = seg_of[p];
    const uint32_t id = order_in[p];
    if (r < 0) {
        order_out[p] = id;
        seg_of_out[p] = -1;
        return;
    }
    const int first = seg_first[r], at = cut[r];
    int to = p, side = p >= at ? 1 : 0;
    if (axis[r] >= 0) {
        const int left_before = scan[p] - scan[first];
        side = flags[p] ? 0 : 1;
        to = flags[p] ? first + left_before : at + (p - first - left_before);
    }
    order_out[to] = id;
    seg_of_out[to] = part_range[2 * (size_t)r + side];
}

// One wave per range of 2 to kDevSmall triangles: the whole subtree, triangles kept in LDS. Same rules (SahBuilder::split with
// stride 1); where no plane separates anything the range is halved after sorting it along its widest axis, which gives the halves
// nth_element gives the host.
__global__ __launch_bounds__(64) void sahd_small(int ranges, const int* __restrict__ small_first, const int* __restrict__ small_last,
                                                 const int* __restrict__ small_node, const int* __restrict__ small_depth, uint32_t* __restrict__ order,
                                                 const float* __restrict__ tri_box, int* __restrict__ child, int* __restrict__ range) {
    __shared__ uint32_t ids[kDevSmall];
    __shared__ float cen[kDevSmall][3], tb[kDevSmall][6];
    __shared__ int st_lo[kDevSmall], st_hi[kDevSmall], st_node[kDevSmall], st_depth[kDevSmall];
    __shared__ int count[kDevPairs];
    __shared__ float blo[kDevPairs][3], bhi[kDevPairs][3];
    const int q = blockIdx.x, t = threadIdx.x;
    if (q >= ranges) return;
    const int first = small_first[q], total = small_last[q] - first + 1;
    if (t < total) {
        const uint32_t id = order[first + t];
        ids[t] = id;
        const float* b = tri_box + 6 * (size_t)id;
        for (int k = 0; k < 6; ++k) tb[t][k] = b[k];
        for (int k = 0; k < 3; ++k) cen[t][k] = 0.5f * b[k] + 0.5f * b[3 + k];
    }
    if (t == 0) { st_lo[0] = 0; st_hi[0] = total - 1; st_node[0] = small_node[q]; st_depth[0] = small_depth[q]; }
    int sp = 1;
    __syncthreads();
    while (sp > 0) {
        --sp;
        const int lo = __builtin_amdgcn_readfirstlane(st_lo[sp]), hi = __builtin_amdgcn_readfirstlane(st_hi[sp]),
                  node = __builtin_amdgcn_readfirstlane(st_node[sp]), depth = __builtin_amdgcn_readfirstlane(st_depth[sp]);
        const int c = hi - lo + 1;
        const bool in = t >= lo && t <= hi;
        float clo[3], chi[3], scale[3];
        for (int k = 0; k < 3; ++k) {
            const bool number = in && cen[in ? t : 0][k] == cen[in ? t : 0][k];
            const float v = cen[in ? t : 0][k];
            clo[k] = ord2f(wave_min_u32(number ? f2ord(v) : f2ord(3.0e38f)));
            chi[k] = ord2f(wave_max_u32(number ? f2ord(v) : f2ord(-3.0e38f)));
            scale[k] = chi[k] > clo[k] ? (float)kDevBins / (chi[k] - clo[k]) : 0.f;
        }
        int best = -1;
        if (c > 2 && depth < SahBuilder::kMaxCostDepth) {
            if (t < kDevPairs) {
                const int k = t >> 4, j = t & 15;
                float l3[3] = {3.0e38f, 3.0e38f, 3.0e38f}, h3[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
                int n_in = 0;
                for (int i = lo; i <= hi; ++i)
                    if (dev_bin_of(cen[i][k], clo[k], scale[k]) == j) {
                        for (int a = 0; a < 3; ++a) { l3[a] = fminf(l3[a], tb[i][a]); h3[a] = fmaxf(h3[a], tb[i][3 + a]); }
                        ++n_in;
                    }
                count[t] = n_in;
                for (int a = 0; a < 3; ++a) { blo[t][a] = l3[a]; bhi[t][a] = h3[a]; }
            }
            __syncthreads();
            float cost = __builtin_inff();
            if (t < kDevPairs && scale[t >> 4] != 0.f && (t & 15) < kDevBins - 1) cost = sahd_plane_cost(count, blo, bhi, t >> 4, t & 15);
            best = sahd_cheapest(cost, t);
        }
        // where every triangle of the range goes
        int at = lo + c / 2, to = t;
        bool moved = false;
        if (best >= 0) {
            const int k = best >> 4, j = best & 15;
            const bool left = in && dev_bin_of(cen[in ? t : 0][k], clo[k], scale[k]) <= j;
            const unsigned long long lefts = __builtin_amdgcn_ballot_w64(left), all = __builtin_amdgcn_ballot_w64(in);
            const int n_left = __builtin_popcountll(lefts);
            if (n_left > 0 && n_left < c) {
                const unsigned long long below = (1ull << t) - 1ull;
                at = lo + n_left;
                to = left ? lo + __builtin_popcountll(lefts & below) : at + __builtin_popcountll(all & ~lefts & below);
                moved = true;
            }
        }
        if (!moved) {
            int k = 0;
            for (int a = 1; a < 3; ++a) if (chi[a] - clo[a] > chi[k] - clo[k]) k = a;
            if (chi[k] > clo[k]) {
                // rank by (centroid coordinate as ordered bits, triangle id): a total order whatever the values
                const unsigned long long mine = ((unsigned long long)f2ord(cen[in ? t : 0][k]) << 32) | ids[in ? t : 0];
                int rank = 0;
                for (int i = lo; i <= hi; ++i) rank += ((((unsigned long long)f2ord(cen[i][k]) << 32) | ids[i]) < mine) ? 1 : 0;
                to = lo + rank;
                moved = true;
            }
        }
        if (moved) {   // (uniform)
            uint32_t id = 0;
            float v[9];
            if (in) {
                id = ids[t];
                for (int a = 0; a < 3; ++a) v[a] = cen[t][a];
                for (int a = 0; a < 6; ++a) v[3 + a] = tb[t][a];
            }
            __syncthreads();
            if (in) {
                ids[to] = id;
                for (int a = 0; a < 3; ++a) cen[to][a] = v[a];
                for (int a = 0; a < 6; ++a) tb[to][a] = v[3 + a];
            }
        }
        const int n_left = at - lo, n_right = hi - at + 1;
        if (t == 0) {
            range[2 * (size_t)node] = first + lo;
            range[2 * (size_t)node + 1] = first + hi;
            child[2 * (size_t)node] = n_left == 1 ? ~(first + lo) : node + 1;
            child[2 * (size_t)node + 1] = n_right == 1 ? ~(first + at) : node + n_left;
            int top = sp;
            if (n_left > 1) { st_lo[top] = lo; st_hi[top] = at - 1; st_node[top] = node + 1; st_depth[top] = depth + 1; ++top; }
            if (n_right > 1) { st_lo[top] = at; st_hi[top] = hi; st_node[top] = node + n_left; st_depth[top] = depth + 1; }
        }
        sp += (n_left > 1 ? 1 : 0) + (n_right > 1 ? 1 : 0);
        __syncthreads();
    }
    if (t < total) order[first + t] = ids[t];
}

}  // namespace

hipError_t pt_bvh_build_topology_binned(const PtBvhBuild& b, hipStream_t stream, int32_t* out_passes) {
    const int n = b.num_triangles;
    if (out_passes) *out_passes = 0;
    if (n < 2) return hipSuccess;
    // Morton order to start from: the partitions are stable, so a range stays in that order and its halves (where no plane
    // separates anything) are halves in space as far as the curve allows
    hipError_t e = morton_sorted_order(b, stream);
    if (e != hipSuccess) return e;
    const bool report = getenv("PT_DEBUG_SCHED") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    DeviceScratch scratch;
    const size_t nn = (size_t)n, max_ranges = nn / (kDevSmall + 1) + 2, max_small = nn / 2 + 2;
    int *seg[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}, *seg_of[2] = {nullptr, nullptr}, *axis = nullptr, *bin = nullptr, *cut = nullptr,
        *part_range = nullptr, *flags = nullptr, *scan = nullptr, *small[4] = {nullptr, nullptr, nullptr, nullptr}, *counters = nullptr;
    float *plane_lo = nullptr, *plane_scale = nullptr;
    uint32_t *cb = nullptr, *cnt = nullptr, *box = nullptr;
    for (int a = 0; a < 2; ++a) {
        for (int k = 0; k < 3; ++k)
            if ((e = scratch.get(&seg[a][k], max_ranges)) != hipSuccess) return e;
        if ((e = scratch.get(&seg_of[a], nn)) != hipSuccess) return e;
    }
    for (int k = 0; k < 4; ++k)
        if ((e = scratch.get(&small[k], max_small)) != hipSuccess) return e;
    if ((e = scratch.get(&axis, max_ranges)) != hipSuccess || (e = scratch.get(&bin, max_ranges)) != hipSuccess || (e = scratch.get(&cut, max_ranges)) != hipSuccess ||
        (e = scratch.get(&part_range, 2 * max_ranges)) != hipSuccess || (e = scratch.get(&plane_lo, max_ranges)) != hipSuccess ||
        (e = scratch.get(&plane_scale, max_ranges)) != hipSuccess || (e = scratch.get(&flags, nn)) != hipSuccess || (e = scratch.get(&scan, nn)) != hipSuccess ||
        (e = scratch.get(&cb, 6 * max_ranges)) != hipSuccess || (e = scratch.get(&cnt, kDevPairs * max_ranges)) != hipSuccess ||
        (e = scratch.get(&box, 6 * kDevPairs * max_ranges)) != hipSuccess || (e = scratch.get(&counters, 2)) != hipSuccess)
        return e;
    uint32_t* order[2] = {b.order, b.order_tmp};
    int ranges = 0, smalls = 0, levels = 0, cur = 0;
    e = hipMemsetAsync(counters, 0, 2 * sizeof(int), stream);
    if (e != hipSuccess) return e;
    {
        const int root[4] = {0, n - 1, 0, 0};
        if (n > kDevSmall) {
            for (int k = 0; k < 3 && e == hipSuccess; ++k) e = hipMemcpyAsync(seg[0][k], &root[k], sizeof(int), hipMemcpyHostToDevice, stream);
            hipLaunchKernelGGL(sahd_iota_ranges, grid_for(n), dim3(256), 0, stream, n, seg_of[0], 0);
            ranges = 1;
        } else {
            for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipMemcpyAsync(small[k], &root[k], sizeof(int), hipMemcpyHostToDevice, stream);
            smalls = 1;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);   // (`root` leaves scope)
        if (e != hipSuccess) return e;
    }
    while (ranges > 0) {
        if (levels > 4096) return hipErrorUnknown;   // (a level halves its ranges at the latest from depth kMaxCostDepth on)
        const int cost_driven = levels < SahBuilder::kMaxCostDepth ? 1 : 0;
        int** s = seg[cur];
        if (cost_driven) {
            hipLaunchKernelGGL(sahd_reset, grid_for(ranges * kDevPairs * 6), dim3(256), 0, stream, ranges, cb, cnt, box);
            hipLaunchKernelGGL(sahd_bounds, grid_for(n), dim3(256), 0, stream, order[cur], b.tri_box, seg_of[cur], s[0], s[1], n, cb);
            hipLaunchKernelGGL(sahd_bins, grid_for(n), dim3(256), 0, stream, order[cur], b.tri_box, seg_of[cur], s[0], s[1], n, cb, cnt, box);
            hipLaunchKernelGGL(sahd_choose, dim3((unsigned)ranges), dim3(64), 0, stream, cb, cnt, box, axis, bin, plane_lo, plane_scale);
        }
        hipLaunchKernelGGL(sahd_flags, grid_for(n), dim3(256), 0, stream, order[cur], b.tri_box, seg_of[cur], n, axis, bin, plane_lo, plane_scale, cost_driven, flags);
        size_t bytes = b.sort_tmp_bytes;
        e = hipcub::DeviceScan::ExclusiveSum(b.sort_tmp, bytes, flags, scan, n, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(sahd_split, grid_for(ranges), dim3(256), 0, stream, ranges, s[0], s[1], s[2], flags, scan, cost_driven, levels, axis, cut, part_range,
                           b.child, b.range, seg[cur ^ 1][0], seg[cur ^ 1][1], seg[cur ^ 1][2], small[0], small[1], small[2], small[3], counters);
        hipLaunchKernelGGL(sahd_scatter, grid_for(n), dim3(256), 0, stream, order[cur], seg_of[cur], n, s[0], axis, cut, part_range, flags, scan, order[cur ^ 1],
                           seg_of[cur ^ 1]);
        int h[2] = {0, 0};
        e = hipMemcpyAsync(h, counters, sizeof h, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipMemsetAsync(counters, 0, sizeof(int), stream);   // (the count of small ranges runs on)
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        if ((size_t)h[0] > max_ranges || (size_t)h[1] > max_small) return hipErrorUnknown;
        ranges = h[0];
        smalls = h[1];
        cur ^= 1;
        ++levels;
    }
    if (smalls > 0)
        hipLaunchKernelGGL(sahd_small, dim3((unsigned)smalls), dim3(64), 0, stream, smalls, small[0], small[1], small[2], small[3], order[cur], b.tri_box, b.child,
                           b.range);
    if (cur != 0) {
        e = hipMemcpyAsync(b.order, b.order_tmp, nn * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
    }
    e = finish_topology(b, stream, out_passes);
    if (e != hipSuccess) return e;
    if (out_passes && *out_passes > PT_BVH_STACK - 16) {   // (cannot happen: cost-driven splits end at kMaxCostDepth; kept as a guard)
        e = morton_sorted_order(b, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(bvh_hierarchy, grid_for(n - 1), dim3(256), 0, stream, b.keys, n, b.child, b.range);
        e = finish_topology(b, stream, out_passes);
        if (e != hipSuccess) return e;
    }
    // wide nodes (scratch: four arrays of at least n - 1 ints that the build no longer needs)
    e = collapse_wide_on_device(b, seg_of[0], seg_of[1], flags, scan, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (report)
        fprintf(stderr, "[pt bvh] device surface-area builder, %d triangles: %d levels of large ranges, %d small ranges, %d sweeps, %.2f ms with the wide nodes\n", n,
                levels, smalls, out_passes ? *out_passes : 0, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    return e;
}

hipError_t pt_bvh_emit(const PtBvhBuild& b, const float ro[3], const float* tris, const float* tris_abs, hipStream_t stream) {
    const int n = b.num_triangles;
    if (n < 1) return hipSuccess;
    if (n >= 2) {
        hipLaunchKernelGGL(bvh_emit_nodes, grid_for(n - 1), dim3(256), 0, stream, b.order, b.child, b.range, b.tri_box, b.node_box, n, ro[0],
                           ro[1], ro[2], 0.0f, b.nodes);
        // Absolute boxes for rays that start anywhere in the scene: the slab arithmetic then errs relative to the
        // larger of box and origin coordinates, so the pad also carries a term of the scene's size.
        hipLaunchKernelGGL(bvh_emit_nodes, grid_for(n - 1), dim3(256), 0, stream, b.order, b.child, b.range, b.tri_box, b.node_box, n, 0.0f,
                           0.0f, 0.0f, 4.0e-6f * b.scene_scale, b.nodes_abs);
        hipLaunchKernelGGL(bvh_emit_qframe, dim3(1), dim3(64), 0, stream, b.node_box, b.scene_scale, b.qframe);
        hipLaunchKernelGGL(bvh_emit_wide, grid_for(n - 1), dim3(256), 0, stream, b.order, b.wide_child, b.range, b.tri_box, b.node_box, b.wide_index, n,
                           4.0e-6f * b.scene_scale, b.qframe, b.nodes_w);
    }
    hipLaunchKernelGGL(bvh_gather_qrecords, grid_for(n + PT_TRI_PAD), dim3(256), 0, stream, b.order, reinterpret_cast<const float4*>(tris_abs), n,
                       reinterpret_cast<float4*>(b.tris_q));
    hipLaunchKernelGGL(bvh_gather_records, grid_for(n + PT_TRI_PAD), dim3(256), 0, stream, b.order, reinterpret_cast<const float4*>(tris_abs), n,
                       reinterpret_cast<float4*>(b.sorted_tris_abs), b.sorted_ids, false);
    hipLaunchKernelGGL(bvh_gather_records, grid_for(n + PT_TRI_PAD), dim3(256), 0, stream, b.order, reinterpret_cast<const float4*>(tris), n,
                       reinterpret_cast<float4*>(b.sorted_tris), b.sorted_ids, true);
    return hipGetLastError();
}
