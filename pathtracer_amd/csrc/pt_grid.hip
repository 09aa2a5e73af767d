// Direction grid: an acceleration structure for rays that all start at one point.
//
// The reference shader never moves its ray origin (pathtracing.fs: every "ro = ..." in Shade is commented out), so every
// closest-hit query -- camera rays and all bounce rays alike -- is "what does the camera position see in direction d".
// That makes the search two-dimensional: the directions are cut into the cells of a cube map around the origin (6 faces
// x R x R cells), and each cell lists the triangles that a direction inside it could hit. A query looks up its cell and
// runs the exact triangle test on that list only: no hierarchy to walk, no chain of dependent fetches.
//
// Exactness: a cell's list must contain every triangle the exact fp32 test could accept for ANY direction of the cell.
// The lists are built from the same origin-relative records the test uses (A = -ap, B = A + e1, C = A + e2): the triangle
// is clipped to the face's pyramid (widened by a margin), projected, and entered into every cell its projected outline
// may touch, with margins far above the angular slop of the fp32 test (about 1e-7 rad; margins are 1e-5 of the face
// width plus a relative term). Triangles the origin sees from behind or edge-on (e2r <= 0) can never be hit and are
// left out. Triangles in a list are tested in any order with the explicit lowest-index rule for equal distances.
#include "pt_internal.h"

#include <hipcub/hipcub.hpp>

namespace {

struct P3 { float x, y, z; };

// face f = 2 * axis + (negative ? 1 : 0); local coordinates (u, v, w) with w > 0 inside the face's pyramid. (u, v) are the
// cube-map coordinates (sc, tc) the hardware's cube instructions give a direction of that face, which is what a query uses
// (cube_coords in pt_kernels.hip): +x: (-z, -y), -x: (z, -y), +y: (x, z), -y: (x, -z), +z: (x, -y), -z: (-x, -y).
__device__ __forceinline__ P3 to_face(P3 p, int f) {
    P3 q;
    switch (f) {
        case 0: q.x = -p.z; q.y = -p.y; q.z = p.x; break;
        case 1: q.x = p.z; q.y = -p.y; q.z = -p.x; break;
        case 2: q.x = p.x; q.y = p.z; q.z = p.y; break;
        case 3: q.x = p.x; q.y = -p.z; q.z = -p.y; break;
        case 4: q.x = p.x; q.y = -p.y; q.z = p.z; break;
        default: q.x = -p.x; q.y = -p.y; q.z = -p.z; break;
    }
    return q;
}

constexpr int kMaxPoly = 8;
constexpr float kPyramidSlack = 1.0e-4f;  // the pyramid is widened by this fraction before clipping
constexpr float kCellSlack = 2.0e-5f;     // in face coordinates (the face is 2 wide)

// Sutherland-Hodgman step: keeps the part of the polygon with nx*x + ny*y + nz*z >= 0
__device__ int clip_plane(const P3* in, int n, P3* out, float nx, float ny, float nz) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const P3 a = in[i], b = in[(i + 1) % n];
        const float da = nx * a.x + ny * a.y + nz * a.z, db = nx * b.x + ny * b.y + nz * b.z;
        if (da >= 0.f) out[m++] = a;
        if ((da >= 0.f) != (db >= 0.f)) {
            const float t = da / (da - db);
            P3 c;
            c.x = a.x + t * (b.x - a.x); c.y = a.y + t * (b.y - a.y); c.z = a.z + t * (b.z - a.z);
            if (m < kMaxPoly) out[m++] = c;
        }
        if (m >= kMaxPoly) break;
    }
    return m;
}

// One (triangle, face) job: the triangle clipped to the face's pyramid and projected; which cells its outline may touch.
struct FaceJob {
    float pu[kMaxPoly], pv[kMaxPoly];
    int m;                      // polygon vertices
    int iu0, iu1, iv0, iv1;     // cell range of the bounding box
    float slack, orient;
    bool use_edges;
};

// The triangle (A, A + e1, A + e2 relative to the ray origin) on face f.
__device__ bool setup_face(P3 A, P3 B, P3 C, P3 e1, P3 e2, int f, int res, FaceJob& job) {
    P3 pa[kMaxPoly], pb[kMaxPoly];
    pa[0] = to_face(A, f); pa[1] = to_face(B, f); pa[2] = to_face(C, f);
    // cheap reject: the whole triangle behind the face's apex plane
    if (pa[0].z <= 0.f && pa[1].z <= 0.f && pa[2].z <= 0.f) return false;
    const float k = 1.0f + kPyramidSlack;
    int m = clip_plane(pa, 3, pb, -1.f, 0.f, k);   // k*w - u >= 0
    m = clip_plane(pb, m, pa, 1.f, 0.f, k);        // k*w + u >= 0
    m = clip_plane(pa, m, pb, 0.f, -1.f, k);
    m = clip_plane(pb, m, pa, 0.f, 1.f, k);
    if (m < 1) return false;
    // project; bounding box in face coordinates
    float u0 = 4.f, u1 = -4.f, v0 = 4.f, v1 = -4.f;
    int valid = 0;
    for (int q = 0; q < m; ++q) {
        if (!(pa[q].z > 0.f)) continue;  // only the apex itself can have w = 0 after clipping
        const float iw = 1.0f / pa[q].z;
        job.pu[valid] = pa[q].x * iw; job.pv[valid] = pa[q].y * iw;
        u0 = fminf(u0, job.pu[valid]); u1 = fmaxf(u1, job.pu[valid]);
        v0 = fminf(v0, job.pv[valid]); v1 = fmaxf(v1, job.pv[valid]);
        ++valid;
    }
    if (valid == 0) return false;
    m = valid;
    // How far outside its true outline can the fp32 test still accept this triangle? Each of its three conditions is a
    // plane through the origin and one edge; the computed side of a direction is off by a few eps x S over the doubled
    // area of the triangle (origin, edge), S bounding the magnitudes that enter the sums. That is ~1e-7 rad for ordinary
    // views and grows only when an edge points at the origin. x4: radians -> face coordinates, both axes.
    const float lA = sqrtf(A.x * A.x + A.y * A.y + A.z * A.z), l1 = sqrtf(e1.x * e1.x + e1.y * e1.y + e1.z * e1.z),
                l2 = sqrtf(e2.x * e2.x + e2.y * e2.y + e2.z * e2.z);
    auto cross_len = [](P3 a, P3 b) {
        const float x = a.y * b.z - a.z * b.y, y = a.z * b.x - a.x * b.z, z = a.x * b.y - a.y * b.x;
        return sqrtf(x * x + y * y + z * z);
    };
    const float S = lA * (l1 + l2) + l1 * l2;
    const float weakest = fminf(cross_len(A, B), fminf(cross_len(A, C), cross_len(B, C)));
    const float fuzz = weakest > 0.f ? fminf(4.0f * 16.0f * 5.97e-8f * S / weakest, 4.0f) : 4.0f;
    const float slack = kCellSlack + fuzz + 1.0e-5f * fmaxf(fmaxf(fabsf(u0), fabsf(u1)), fmaxf(fabsf(v0), fabsf(v1)));
    const float half = 0.5f * (float)res;
    auto cell_of = [half, res](float c) { const int q = (int)floorf((c + 1.0f) * half); return q < 0 ? 0 : (q >= res ? res - 1 : q); };
    if (u1 + slack < -1.f || u0 - slack > 1.f || v1 + slack < -1.f || v0 - slack > 1.f) return false;
    job.iu0 = cell_of(u0 - slack); job.iu1 = cell_of(u1 + slack); job.iv0 = cell_of(v0 - slack); job.iv1 = cell_of(v1 + slack);
    // outline test: a cell wholly outside one edge of the (convex) projected polygon cannot be touched. Only when the
    // polygon has a safe area; a sliver falls back to its bounding box.
    float area2 = 0.f;
    for (int q = 0; q < m; ++q) { const int q2 = (q + 1) % m; area2 += job.pu[q] * job.pv[q2] - job.pu[q2] * job.pv[q]; }
    const float extent = fmaxf(u1 - u0, v1 - v0);
    job.use_edges = m >= 3 && fabsf(area2) > 1.0e-3f * extent * extent && extent > 0.f;
    job.orient = area2 > 0.f ? 1.f : -1.f;
    job.m = m;
    job.slack = slack;
    return true;
}

// record: e1.xyz e2.x | e2.yz ap.xy | ap.z r.xyz | e2r - - id
struct TriCorners { P3 A, B, C, e1, e2; };
__device__ __forceinline__ bool load_corners(const float4* __restrict__ tris, int i, TriCorners& t) {
    const float4 r0 = tris[4 * (size_t)i], r1 = tris[4 * (size_t)i + 1], r2 = tris[4 * (size_t)i + 2], r3 = tris[4 * (size_t)i + 3];
    if (!(r3.x > 0.0f)) return false;  // e2r <= 0 (or NaN): no ray from the origin can be accepted
    t.A.x = -r1.z; t.A.y = -r1.w; t.A.z = -r2.x;
    t.e1.x = r0.x; t.e1.y = r0.y; t.e1.z = r0.z;
    t.e2.x = r0.w; t.e2.y = r1.x; t.e2.z = r1.y;
    t.B.x = t.A.x + r0.x; t.B.y = t.A.y + r0.y; t.B.z = t.A.z + r0.z;
    t.C.x = t.A.x + r0.w; t.C.y = t.A.y + r1.x; t.C.z = t.A.z + r1.y;
    return true;
}

// Is the triangle wholly outside one side plane of face f's (widened) pyramid? The same sums as the first clipping step
// forms for each plane, on the three corners: if all three are negative for one plane nothing of the triangle is on this face.
__device__ __forceinline__ bool outside_pyramid(const TriCorners& t, int f) {
    const P3 a = to_face(t.A, f), b = to_face(t.B, f), c = to_face(t.C, f);
    if (a.z <= 0.f && b.z <= 0.f && c.z <= 0.f) return true;
    const float k = 1.0f + kPyramidSlack;
    auto side = [k](P3 p, float nx, float ny) { return nx * p.x + ny * p.y + k * p.z; };
    if (side(a, -1.f, 0.f) < 0.f && side(b, -1.f, 0.f) < 0.f && side(c, -1.f, 0.f) < 0.f) return true;
    if (side(a, 1.f, 0.f) < 0.f && side(b, 1.f, 0.f) < 0.f && side(c, 1.f, 0.f) < 0.f) return true;
    if (side(a, 0.f, -1.f) < 0.f && side(b, 0.f, -1.f) < 0.f && side(c, 0.f, -1.f) < 0.f) return true;
    if (side(a, 0.f, 1.f) < 0.f && side(b, 0.f, 1.f) < 0.f && side(c, 0.f, 1.f) < 0.f) return true;
    return false;
}

__device__ bool setup_job(const float4* __restrict__ tris, int i, int f, int res, FaceJob& job) {
    TriCorners t;
    if (!load_corners(tris, i, t)) return false;
    return setup_face(t.A, t.B, t.C, t.e1, t.e2, f, res, job);
}

__device__ __forceinline__ bool cell_touched(const FaceJob& job, int res, int iu, int iv) {
    if (!job.use_edges) return true;
    const float cw = 2.0f / (float)res;
    const float cu0 = -1.f + cw * (float)iu - job.slack, cu1 = -1.f + cw * (float)(iu + 1) + job.slack;
    const float cv0 = -1.f + cw * (float)iv - job.slack, cv1 = -1.f + cw * (float)(iv + 1) + job.slack;
    for (int q = 0; q < job.m; ++q) {
        const int q2 = (q + 1) % job.m;
        // inward normal of edge q -> q2 for a counter-clockwise polygon: (-dy, dx)
        const float nx = -(job.pv[q2] - job.pv[q]) * job.orient, ny = (job.pu[q2] - job.pu[q]) * job.orient;
        // the cell corner farthest along the inward normal
        const float cx = nx > 0.f ? cu1 : cu0, cy = ny > 0.f ? cv1 : cv0;
        const float d = nx * (cx - job.pu[q]) + ny * (cy - job.pv[q]);
        if (d < -1.0e-5f * (fabsf(nx) + fabsf(ny))) return false;  // even that corner is outside this edge
    }
    return true;
}

// What a query reads (see intersect_grid in pt_kernels.hip): one word per cell --
//   PT_GRID_EMPTY                 no triangle
//   id            (top bit clear) exactly one triangle: the cell IS the list, no second fetch
//   1<<31 | begin                 two or more: list[begin...], the last entry of a cell carries the top bit
// so only cells with two or more triangles take list space (cell_start = prefix sums of those counts).
template <bool COUNT>
__device__ __forceinline__ void enter(int res, int f, int iu, int iv, uint32_t id, uint32_t* __restrict__ cell_count,
                                      const uint32_t* __restrict__ cell_start, uint32_t* __restrict__ list, uint32_t* __restrict__ cells) {
    const size_t cell = ((size_t)f * res + iv) * res + iu;
    if (COUNT) {
        atomicAdd(&cell_count[cell], 1u);
        return;
    }
    const uint32_t begin = cell_start[cell], need = cell_start[cell + 1] - begin;
    if (need == 0u) {
        cells[cell] = id;  // the cell's only triangle
    } else {
        const uint32_t pos = atomicAdd(&cell_count[cell], 1u);
        list[begin + pos] = id | (pos + 1u == need ? 0x80000000u : 0u);
        if (pos == 0u) cells[cell] = 0x80000000u | begin;
    }
}

// list space of a cell: none for 0 or 1 triangles
struct ListNeed {
    __host__ __device__ __forceinline__ uint32_t operator()(const uint32_t& count) const { return count >= 2u ? count : 0u; }
};

// Jobs covering more cells than this go to the queue of big jobs. A small mesh queues every job: its few threads would
// otherwise walk their cells one after the other on a nearly empty device (3,936 triangles: 0.20 ms for the two passes).
constexpr int kInlineCells = 48;
constexpr int kSmallMesh = 16384;
__host__ __device__ inline int inline_cells(int n) { return n <= kSmallMesh ? 0 : kInlineCells; }

// The count pass refuses to walk once the bounding-box cells of all jobs together (an upper bound of the list length,
// summed in 64 bits) reach this: the resolution is too fine for this mesh, and walking would take as long as the lists are.
constexpr unsigned long long kMaxListBound = 1ull << 30;

// One thread per triangle. COUNT = true: cell_count[cell]++ and bound[] += the job's bounding-box cells; false: the
// triangle is entered into the cell (enter()). A job whose bounding box covers many cells is not walked here but queued
// for grid_bin_big (one workgroup per job).
template <bool COUNT>
__global__ __launch_bounds__(128) void grid_bin_triangles(const float4* __restrict__ tris, int n, int res, uint32_t* __restrict__ cell_count,
                                                          const uint32_t* __restrict__ cell_start, uint32_t* __restrict__ list,
                                                          uint32_t* __restrict__ cells_out, uint32_t* __restrict__ big_jobs, uint32_t* __restrict__ big_cells,
                                                          uint32_t* __restrict__ big_count, uint32_t big_capacity, unsigned long long* __restrict__ bound) {
    // One thread per triangle, its six faces in turn: neighbouring triangles mostly fall on the same face, so the clipping
    // (the expensive part) runs with most lanes of the wave instead of one lane in six.
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    TriCorners t;
    const bool live = i < n && load_corners(tris, i, t);
    unsigned long long my_cells = 0;
    unsigned my_entries = 0;
    for (int f = 0; f < 6; ++f) {
        if (!live || outside_pyramid(t, f)) continue;
        FaceJob job;
        if (!setup_face(t.A, t.B, t.C, t.e1, t.e2, f, res, job)) continue;
        const int cells = (job.iu1 - job.iu0 + 1) * (job.iv1 - job.iv0 + 1);
        my_cells += (unsigned long long)cells;
        if (cells > inline_cells(n)) {
            const uint32_t slot = atomicAdd(big_count, 1u);
            if (slot < big_capacity) {
                big_jobs[slot] = (uint32_t)(i * 6 + f);
                big_cells[slot] = (uint32_t)cells;
                continue;
            }
            if (COUNT) {
                atomicAdd(&bound[0], kMaxListBound);  // more big jobs than the queue holds: too fine a resolution as well
                continue;
            }
            // (fill pass: the count pass queued the same jobs, so the queue cannot be full here)
        }
        for (int iv = job.iv0; iv <= job.iv1; ++iv)
            for (int iu = job.iu0; iu <= job.iu1; ++iu)
                if (cell_touched(job, res, iu, iv)) {
                    enter<COUNT>(res, f, iu, iv, (uint32_t)i, cell_count, cell_start, list, cells_out);
                    ++my_entries;
                }
    }
    if (COUNT) {
        // (spread over PT_GRID_BOUND_SLOTS words: atomics on one address from every wave doubled the time of this kernel)
        unsigned long long entries = my_entries;
        for (int off = 32; off > 0; off >>= 1) {
            my_cells += __shfl_down(my_cells, off);
            entries += __shfl_down(entries, off);
        }
        const unsigned slot = (blockIdx.x * 2u + (threadIdx.x >> 6)) % PT_GRID_BOUND_SLOTS;
        if ((threadIdx.x & 63) == 0 && my_cells) atomicAdd(&bound[slot], my_cells);
        if ((threadIdx.x & 63) == 0 && entries) atomicAdd(&bound[PT_GRID_BOUND_SLOTS + slot], entries);
    }
}

template <bool COUNT>
__global__ __launch_bounds__(256) void grid_bin_big(const float4* __restrict__ tris, int res, uint32_t* __restrict__ cell_count,
                                                    const uint32_t* __restrict__ cell_start, uint32_t* __restrict__ list,
                                                    uint32_t* __restrict__ cells_out, const uint32_t* __restrict__ big_jobs, const uint32_t* __restrict__ big_cells,
                                                    const uint32_t* __restrict__ big_count, uint32_t big_capacity, unsigned long long* __restrict__ bound) {
    if (COUNT) {  // (the bound is complete: grid_bin_triangles ran before this kernel)
        static_assert(PT_GRID_BOUND_SLOTS == 256, "one slot per thread of this block");
        __shared__ unsigned long long partial[4];
        unsigned long long sum = bound[threadIdx.x];
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
        if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (partial[0] + partial[1] + partial[2] + partial[3] >= kMaxListBound) return;
    }
    const uint32_t queued = *big_count < big_capacity ? *big_count : big_capacity;
    if (queued == 0u) return;
    // Few jobs (a small mesh seen from close by: 36 triangles make about a hundred jobs of up to a whole face each): the
    // cells of a job are shared out over `parts` workgroups, so that the launch still fills the device.
    const uint32_t parts = queued < gridDim.x ? gridDim.x / queued : 1u;
    const uint32_t part = blockIdx.x / queued;
    if (parts > 1u && part >= parts) return;
    // (a part beyond the job's cells has nothing to do: leave before the clipping)
    if (parts > 1u && part * blockDim.x >= big_cells[blockIdx.x % queued]) return;
    unsigned long long my_entries = 0;
    for (uint32_t slot = parts > 1u ? blockIdx.x % queued : blockIdx.x; slot < queued; slot += gridDim.x) {
        const int j = (int)big_jobs[slot], i = j / 6, f = j % 6;
        FaceJob job;
        if (!setup_job(tris, i, f, res, job)) continue;  // (cannot fail: it succeeded when the job was queued)
        const int wu = job.iu1 - job.iu0 + 1, cells = wu * (job.iv1 - job.iv0 + 1);
        for (int c = (int)(part * blockDim.x + threadIdx.x); c < cells; c += (int)(parts * blockDim.x)) {
            const int iu = job.iu0 + c % wu, iv = job.iv0 + c / wu;
            if (cell_touched(job, res, iu, iv)) {
                enter<COUNT>(res, f, iu, iv, (uint32_t)i, cell_count, cell_start, list, cells_out);
                ++my_entries;
            }
        }
    }
    if (COUNT) {
        for (int off = 32; off > 0; off >>= 1) my_entries += __shfl_down(my_entries, off);
        if ((threadIdx.x & 63) == 0 && my_entries) atomicAdd(&bound[PT_GRID_BOUND_SLOTS + (blockIdx.x * 4u + (threadIdx.x >> 6)) % PT_GRID_BOUND_SLOTS], my_entries);
    }
}

// summary[0] = upper bound of the list length, [1] = (cell, triangle) entries, [2] = list words needed
__global__ __launch_bounds__(PT_GRID_BOUND_SLOTS) void grid_summary(const unsigned long long* __restrict__ slots, const uint32_t* __restrict__ list_need,
                                                                   unsigned long long* __restrict__ summary) {
    __shared__ unsigned long long partial[2][PT_GRID_BOUND_SLOTS / 64];
    unsigned long long b = slots[threadIdx.x], e = slots[PT_GRID_BOUND_SLOTS + threadIdx.x];
    for (int off = 32; off > 0; off >>= 1) {
        b += __shfl_down(b, off);
        e += __shfl_down(e, off);
    }
    if ((threadIdx.x & 63) == 0) { partial[0][threadIdx.x >> 6] = b; partial[1][threadIdx.x >> 6] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long bs = 0, es = 0;
        for (int w = 0; w < PT_GRID_BOUND_SLOTS / 64; ++w) { bs += partial[0][w]; es += partial[1][w]; }
        summary[0] = bs;
        summary[1] = es;
        summary[2] = *list_need;
    }
}

// After the count pass: one bit per 4x4 block of cells, set when any cell of the block was entered by a triangle; *empty += the
// blocks with none. A query looks at the bit first where much of the sphere of directions is empty (open scenes: most bounce rays
// leave): the bits of a whole face fit in the caches next to the compute units, a cell word of a fine grid costs a line from
// the fabric. One thread per block, a wave writes its 64 bits as two words.
__global__ __launch_bounds__(1024) void grid_occupancy(const uint32_t* __restrict__ cell_count, int res, uint32_t* __restrict__ occ,
                                                      unsigned long long* __restrict__ empty) {
    const int edge = PT_GRID_OCC_EDGE(res);
    const uint32_t blocks = 6u * (uint32_t)edge * (uint32_t)edge;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    bool any = false;
    if (b < blocks) {
        const int face = (int)(b / (uint32_t)(edge * edge)), rem = (int)(b % (uint32_t)(edge * edge));
        const int bv = rem / edge, bu = rem % edge;
        if ((res & 3) == 0) {
            // every block is whole and its four rows of four counts are 16-byte aligned: one load per row, and neighbouring
            // threads read neighbouring 16 bytes (the sixteen single loads of the general form took 0.30 ms of a camera move
            // on a grid of 2048 cells per face edge, more than either binning pass)
            const uint4* row = reinterpret_cast<const uint4*>(cell_count + ((size_t)face * res + 4 * (size_t)bv) * res) + bu;
            const size_t pitch = (size_t)res / 4;
#pragma unroll
            for (int dv = 0; dv < 4; ++dv) {
                const uint4 c = row[dv * pitch];
                any = any || (c.x | c.y | c.z | c.w) != 0u;
            }
        } else {
            for (int dv = 0; dv < 4; ++dv)
                for (int du = 0; du < 4; ++du) {
                    const int iu = 4 * bu + du, iv = 4 * bv + dv;
                    if (iu < res && iv < res) any = any || cell_count[((size_t)face * res + iv) * res + iu] != 0u;
                }
        }
    }
    // (the count of empty blocks goes to ONE word: summed over the workgroup's sixteen waves first -- an atomic per wave, 24,576
    // of them on one address for a grid of 2048 cells per face edge, took 0.30 ms of a camera move, more than either binning pass)
    __shared__ unsigned s_none[16];
    const unsigned long long bits = __ballot(any), valid = __ballot(b < blocks);
    if ((threadIdx.x & 63) == 0) {
        if (valid) {
            occ[(b >> 6) * 2] = (uint32_t)bits;
            occ[(b >> 6) * 2 + 1] = (uint32_t)(bits >> 32);
        }
        s_none[threadIdx.x >> 6] = (unsigned)__builtin_popcountll(valid & ~bits);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned none = 0;
        for (unsigned w = 0; w < blockDim.x / 64; ++w) none += s_none[w];
        if (none) atomicAdd(empty, (unsigned long long)none);
    }
}

// Start of a pass, in one launch instead of three fills: the cell counters (fill cursors) to zero, the queue length to
// zero; count pass: the bound slots to zero; fill pass: every cell word to PT_GRID_EMPTY. words = 6*res*res + 1.
template <bool COUNT>
__global__ __launch_bounds__(256) void grid_reset(uint32_t* __restrict__ cell_count, uint32_t* __restrict__ cells, size_t words,
                                                  uint32_t* __restrict__ big_count, unsigned long long* __restrict__ bound) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t quads = words / 4;
    if (i < quads) {
        reinterpret_cast<uint4*>(cell_count)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (!COUNT) reinterpret_cast<uint4*>(cells)[i] = make_uint4(PT_GRID_EMPTY, PT_GRID_EMPTY, PT_GRID_EMPTY, PT_GRID_EMPTY);
    }
    if (i < words - quads * 4) {
        cell_count[quads * 4 + i] = 0u;
        if (!COUNT) cells[quads * 4 + i] = PT_GRID_EMPTY;
    }
    if (i == 0) *big_count = 0u;
    if (COUNT && i < 2 * PT_GRID_BOUND_SLOTS) bound[i] = 0ull;
    if (COUNT && i == 0) bound[2 * PT_GRID_BOUND_SLOTS + 3] = 0ull;  // the summary's count of empty blocks (grid_occupancy adds to it)
}

template <bool COUNT>
hipError_t bin_all(const PtGridBuild& g, const float* tris, int32_t n, unsigned long long* bound, hipStream_t stream) {
    const size_t words = (size_t)6 * g.res * g.res + 1;
    const size_t threads = std::max<size_t>(words / 4, 2 * PT_GRID_BOUND_SLOTS);  // (6 * 32 * 32 / 4 = 1536 at the coarsest grid)
    hipLaunchKernelGGL(grid_reset<COUNT>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, g.cell_count, g.cells, words, g.big_count, bound);
    const float4* t4 = reinterpret_cast<const float4*>(tris);
    hipLaunchKernelGGL(grid_bin_triangles<COUNT>, dim3((n + 127) / 128), dim3(128), 0, stream, t4, n, g.res, g.cell_count, g.cell_start, g.list,
                       g.cells, g.big_jobs, g.big_cells, g.big_count, g.big_capacity, bound);
    hipLaunchKernelGGL(grid_bin_big<COUNT>, dim3(2048), dim3(256), 0, stream, t4, g.res, g.cell_count, g.cell_start, g.list, g.cells, g.big_jobs,
                       g.big_cells, g.big_count, g.big_capacity, bound);
    return hipGetLastError();
}

}  // namespace

size_t pt_grid_scan_temp_bytes(size_t cells) {
    size_t bytes = 0;
    hipcub::TransformInputIterator<uint32_t, ListNeed, const uint32_t*> need((const uint32_t*)nullptr, ListNeed());
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, need, (uint32_t*)nullptr, (int)(cells + 1), nullptr);
    return bytes;
}

hipError_t pt_grid_count(const PtGridBuild& g, const float* tris, int32_t n, unsigned long long* slots, unsigned long long* summary, hipStream_t stream) {
    const size_t cells = (size_t)6 * g.res * g.res;
    hipError_t e = bin_all<true>(g, tris, n, slots, stream);
    if (e != hipSuccess) return e;
    {
        const uint32_t edge = (uint32_t)PT_GRID_OCC_EDGE(g.res), blocks = 6u * edge * edge;
        hipLaunchKernelGGL(grid_occupancy, dim3((blocks + 1023) / 1024), dim3(1024), 0, stream, g.cell_count, g.res, g.occ, summary + 3);
    }
    // list space of every cell (none for cells with fewer than two triangles)
    size_t bytes = g.scan_tmp_bytes;
    hipcub::TransformInputIterator<uint32_t, ListNeed, const uint32_t*> need(g.cell_count, ListNeed());
    e = hipcub::DeviceScan::ExclusiveSum(g.scan_tmp, bytes, need, g.cell_start, (int)(cells + 1), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(grid_summary, dim3(1), dim3(PT_GRID_BOUND_SLOTS), 0, stream, slots, g.cell_start + cells, summary);
    return hipGetLastError();
}

hipError_t pt_grid_fill(const PtGridBuild& g, const float* tris, int32_t n, hipStream_t stream) {
    return bin_all<false>(g, tris, n, nullptr, stream);
}
