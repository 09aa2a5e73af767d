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

// face f = 2 * axis + (negative ? 1 : 0); local coordinates (u, v, w) with w > 0 inside the face's pyramid
__device__ __forceinline__ P3 to_face(P3 p, int f) {
    const int axis = f >> 1;
    const float s = (f & 1) ? -1.0f : 1.0f;
    P3 q;
    if (axis == 0) { q.x = p.y; q.y = p.z; q.z = s * p.x; }
    else if (axis == 1) { q.x = p.z; q.y = p.x; q.z = s * p.y; }
    else { q.x = p.x; q.y = p.y; q.z = s * p.z; }
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

__device__ bool setup_job(const float4* __restrict__ tris, int i, int f, int res, FaceJob& job) {
    const float4 r0 = tris[4 * (size_t)i], r1 = tris[4 * (size_t)i + 1], r2 = tris[4 * (size_t)i + 2], r3 = tris[4 * (size_t)i + 3];
    if (!(r3.x > 0.0f)) return false;  // e2r <= 0 (or NaN): no ray from the origin can be accepted
    // record: e1.xyz e2.x | e2.yz ap.xy | ap.z r.xyz | e2r - - id
    P3 A, B, C;
    A.x = -r1.z; A.y = -r1.w; A.z = -r2.x;
    B.x = A.x + r0.x; B.y = A.y + r0.y; B.z = A.z + r0.z;
    C.x = A.x + r0.w; C.y = A.y + r1.x; C.z = A.z + r1.y;
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
    const float lA = sqrtf(A.x * A.x + A.y * A.y + A.z * A.z), l1 = sqrtf(r0.x * r0.x + r0.y * r0.y + r0.z * r0.z),
                l2 = sqrtf(r0.w * r0.w + r1.x * r1.x + r1.y * r1.y);
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

constexpr int kInlineCells = 48;  // jobs covering more cells than this go to the queue of big jobs

// One thread per (triangle, face). COUNT = true: cell_count[cell]++; false: the triangle is entered into the cell (enter()).
// A job whose bounding box covers many cells is not walked here but queued for grid_bin_big (one workgroup per job).
template <bool COUNT>
__global__ __launch_bounds__(128) void grid_bin_triangles(const float4* __restrict__ tris, int n, int res, uint32_t* __restrict__ cell_count,
                                                          const uint32_t* __restrict__ cell_start, uint32_t* __restrict__ list,
                                                          uint32_t* __restrict__ cells_out, uint32_t* __restrict__ big_jobs, uint32_t* __restrict__ big_count, uint32_t big_capacity) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n * 6) return;
    const int i = j / 6, f = j % 6;
    FaceJob job;
    if (!setup_job(tris, i, f, res, job)) return;
    const int cells = (job.iu1 - job.iu0 + 1) * (job.iv1 - job.iv0 + 1);
    if (cells > kInlineCells) {
        const uint32_t slot = atomicAdd(big_count, 1u);
        if (slot < big_capacity) {
            big_jobs[slot] = (uint32_t)j;
            return;
        }
        // queue full: walk it here after all
    }
    for (int iv = job.iv0; iv <= job.iv1; ++iv)
        for (int iu = job.iu0; iu <= job.iu1; ++iu)
            if (cell_touched(job, res, iu, iv)) enter<COUNT>(res, f, iu, iv, (uint32_t)i, cell_count, cell_start, list, cells_out);
}

template <bool COUNT>
__global__ __launch_bounds__(256) void grid_bin_big(const float4* __restrict__ tris, int res, uint32_t* __restrict__ cell_count,
                                                    const uint32_t* __restrict__ cell_start, uint32_t* __restrict__ list,
                                                    uint32_t* __restrict__ cells_out, const uint32_t* __restrict__ big_jobs, const uint32_t* __restrict__ big_count, uint32_t big_capacity) {
    const uint32_t queued = *big_count < big_capacity ? *big_count : big_capacity;
    for (uint32_t slot = blockIdx.x; slot < queued; slot += gridDim.x) {
        const int j = (int)big_jobs[slot], i = j / 6, f = j % 6;
        FaceJob job;
        if (!setup_job(tris, i, f, res, job)) continue;  // (cannot fail: it succeeded when the job was queued)
        const int wu = job.iu1 - job.iu0 + 1, cells = wu * (job.iv1 - job.iv0 + 1);
        for (int c = threadIdx.x; c < cells; c += blockDim.x) {
            const int iu = job.iu0 + c % wu, iv = job.iv0 + c / wu;
            if (cell_touched(job, res, iu, iv)) enter<COUNT>(res, f, iu, iv, (uint32_t)i, cell_count, cell_start, list, cells_out);
        }
    }
}

// Upper bound of the list length at resolution `res`: the bounding-box cells of every job, summed in 64 bits.
__global__ __launch_bounds__(256) void grid_estimate(const float4* __restrict__ tris, int n, int res, unsigned long long* __restrict__ total) {
    __shared__ unsigned long long partial[4];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long cells = 0;
    if (j < n * 6) {
        FaceJob job;
        if (setup_job(tris, j / 6, j % 6, res, job)) cells = (unsigned long long)(job.iu1 - job.iu0 + 1) * (unsigned long long)(job.iv1 - job.iv0 + 1);
    }
    for (int off = 32; off > 0; off >>= 1) cells += __shfl_down(cells, off);
    if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = cells;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long sum = partial[0] + partial[1] + partial[2] + partial[3];
        if (sum) atomicAdd(total, sum);
    }
}

template <bool COUNT>
hipError_t bin_all(const PtGridBuild& g, const float* tris, int32_t n, hipStream_t stream) {
    const size_t cells = (size_t)6 * g.res * g.res;
    hipError_t e = hipMemsetAsync(g.cell_count, 0, (cells + 1) * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(g.big_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const int jobs = n * 6;
    const float4* t4 = reinterpret_cast<const float4*>(tris);
    hipLaunchKernelGGL(grid_bin_triangles<COUNT>, dim3((jobs + 127) / 128), dim3(128), 0, stream, t4, n, g.res, g.cell_count, g.cell_start, g.list,
                       g.cells, g.big_jobs, g.big_count, g.big_capacity);
    hipLaunchKernelGGL(grid_bin_big<COUNT>, dim3(2048), dim3(256), 0, stream, t4, g.res, g.cell_count, g.cell_start, g.list, g.cells, g.big_jobs,
                       g.big_count, g.big_capacity);
    return hipGetLastError();
}

}  // namespace

size_t pt_grid_scan_temp_bytes(size_t cells) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)(cells + 1), nullptr);
    size_t bytes2 = 0;
    hipcub::TransformInputIterator<uint32_t, ListNeed, const uint32_t*> need((const uint32_t*)nullptr, ListNeed());
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes2, need, (uint32_t*)nullptr, (int)(cells + 1), nullptr);
    return bytes > bytes2 ? bytes : bytes2;
}

hipError_t pt_grid_estimate(const float* tris, int32_t n, int res, unsigned long long* total, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(total, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const int jobs = n * 6;
    hipLaunchKernelGGL(grid_estimate, dim3((jobs + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const float4*>(tris), n, res, total);
    return hipGetLastError();
}

hipError_t pt_grid_count(const PtGridBuild& g, const float* tris, int32_t n, hipStream_t stream) {
    const size_t cells = (size_t)6 * g.res * g.res;
    hipError_t e = bin_all<true>(g, tris, n, stream);
    if (e != hipSuccess) return e;
    // all entries (cells[cells] <- their number; `cells` is rewritten by the fill pass), then the list space
    size_t bytes = g.scan_tmp_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(g.scan_tmp, bytes, g.cell_count, g.cells, (int)(cells + 1), stream);
    if (e != hipSuccess) return e;
    bytes = g.scan_tmp_bytes;
    hipcub::TransformInputIterator<uint32_t, ListNeed, const uint32_t*> need(g.cell_count, ListNeed());
    e = hipcub::DeviceScan::ExclusiveSum(g.scan_tmp, bytes, need, g.cell_start, (int)(cells + 1), stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

hipError_t pt_grid_fill(const PtGridBuild& g, const float* tris, int32_t n, hipStream_t stream) {
    const size_t cells = (size_t)6 * g.res * g.res;
    hipError_t e = hipMemsetAsync(g.cells, 0xff, (cells + 1) * sizeof(uint32_t), stream);  // PT_GRID_EMPTY
    if (e != hipSuccess) return e;
    return bin_all<false>(g, tris, n, stream);
}
