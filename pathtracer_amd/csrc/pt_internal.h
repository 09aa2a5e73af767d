// Internal declarations shared by the kernels (pt_kernels.hip) and the C ABI (pt_capi.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// One record per triangle, 16 floats = 64 bytes, 64-byte aligned so that a wave fetches it with a single
// scalar load (the record address is the same for every lane). Everything in it depends only on the triangle
// and on the ray origin, and the origin is the camera position for EVERY ray of the reference shader (its
// Shade() never moves `ro`, pathtracing.fs:224,262,274), so the per-ray work of pathtracing.fs:136-160 that
// does not involve the ray direction is done once per triangle when the mesh or the camera changes:
//   e1 = b - a, e2 = c - a                (pathtracing.fs:136-137)
//   ap = ro - a, r = cross(ap, e1)        (pathtracing.fs:144-145)
//   e2r = dot(e2, r)                      (numerator of pathtracing.fs:160)
// The values are produced by the same fp32 operations in the same order, so results are bit-identical.
#define PT_TRI_FLOATS 16
// All-zero records appended to the triangle buffer: the triangle loop runs in steps of 4 and requests records up to
// 5 past the last real one.
#define PT_TRI_PAD 8
// Cull record per triangle: unit normals of the three edge planes through the ray origin and of the front-facing
// side (12 floats, padded to 16 so a lane fetches it with three aligned 16-byte loads).
#define PT_CULL_FLOATS 16
struct PtTriRecordLayout {
    enum { E1 = 0, E2 = 3, AP = 6, R = 9, E2R = 12 };
};

// LBVH: traversal record of one internal node, 16 floats = 64 bytes (one scalar fetch):
//   [0..2] left child box centre, [3..5] its half extents, [6..8] right centre, [9..11] right half extents -- relative
//   to the ray origin (or absolute, for rays with their own origins), padded
//   [12] left ref, [13] right ref (uint bits): 0x80000000 | first << 3 | count-1 = leaf of `count` triangles starting
//   at sorted position `first`; otherwise the index of an internal node.
#define PT_BVH_NODE_FLOATS 16
// Wide, compact form of the same hierarchy for walks that run once per LANE (rays with their own origins). Such a walk is a chain
// of dependent fetches, so a node holds FOUR children -- every second level of the binary hierarchy is skipped: the children of
// a node's internal children take their place -- in 64 bytes: per child three words (lo | hi << 16 per axis, 16-bit steps of a
// frame around the whole hierarchy: base + q * step, rounded outwards by a step and more) and its reference (a leaf as in the
// binary form, the index of a wide node, or PT_BVH_NO_CHILD). A triangle record is 48 bytes: a.xyz e1.x | e1.yz e2.xy | e2.z id - -.
#define PT_BVH_WNODE_WORDS 16
#ifndef PT_BVH_WIDE_LEAF_MAX
#define PT_BVH_WIDE_LEAF_MAX 2  // triangles per leaf of the wide form (1M triangles: 11.1 ms at 2, 11.3 at 1, 11.8 at 4)
#endif
#define PT_BVH_NO_CHILD 0xffffffffu
#define PT_BVH_NO_WIDE_CHILD ((int)0x80000000)  // in PtBvhBuild::wide_child (no leaf position reaches ~INT_MIN)
#define PT_BVH_QTRI_FLOATS 12
#ifndef PT_BVH_LEAF_MAX
#define PT_BVH_LEAF_MAX 4
#endif
#define PT_BVH_STACK 96
#define PT_LANE_STACK 24  // per-lane walk: stack entries per lane (LDS); deeper paths fall back to the per-wave walk
// PT_ACCEL_AUTO walks the hierarchy above these triangle counts (measured: 36 triangles: loop = walk for camera-origin rays,
// loop 1.3x faster for rays with their own origins; 64 triangles: walk 2.8x faster; 968: 40x)
#define PT_BVH_AUTO_MIN_TRIANGLES 48
#define PT_BVH_AUTO_MIN_TRIANGLES_MOVING 96

// Statistics counters are spread over many cache lines: every wave adds to them once, and with one address those
// additions queue up behind each other in L2 (measured: 32,400 short waves took 0.83 ms for 0.12 ms of work).
#define PT_COUNTER_SLOTS 1024
#define PT_COUNTER_STRIDE 16  // 64-bit words per slot = one 128-byte line

struct PtBvhBuild {
    int32_t num_triangles;
    float scene_scale;            // largest absolute vertex coordinate (host-computed at upload)
    const float* vertices;        // raw upload (6 floats per vertex)
    const uint32_t* indices;      // raw upload
    float* tri_box;               // n x 6, absolute, original triangle order
    uint32_t* scene_box;          // 6 order-preserving uints: centroid bounds
    unsigned long long* keys_tmp; // n
    unsigned long long* keys;     // n, sorted 63-bit Morton codes of the centroids (duplicates allowed)
    uint32_t* order_tmp;          // n
    uint32_t* order;              // n: triangle id at each sorted position
    void* sort_tmp;
    size_t sort_tmp_bytes;
    int* child;                   // (n-1) x 2
    int* range;                   // (n-1) x 2: first,last sorted position
    float* node_box;              // (n-1) x 6, absolute
    int* done;                    // (n-1): refit sweep in which the node was finished
    int* remaining;               // 1
    float* nodes;                 // (n-1) x PT_BVH_NODE_FLOATS traversal records
    float* nodes_abs;             // (n-1) x PT_BVH_NODE_FLOATS, absolute boxes
    float* sorted_tris_abs;       // (n + PT_TRI_PAD) x PT_TRI_FLOATS: a, e1, e2 in leaf order
    int* wide_index;              // (n-1): index of the wide node made of this binary node, or -1 (absorbed by a wide node, or inside a leaf)
    int* wide_child;              // (n-1) x 4: binary refs of each wide node's children (PT_BVH_NO_WIDE_CHILD: slot unused)
    uint32_t* nodes_w;            // up to (n-1) x PT_BVH_WNODE_WORDS: wide nodes, absolute boxes in 16-bit steps of the frame
    float* tris_q;                // (n + PT_TRI_PAD) x PT_BVH_QTRI_FLOATS: a, e1, e2, id in leaf order
    float* qframe;                // 8: base xyz, step xyz of the frame, -, -
    float* sorted_tris;           // (n + PT_TRI_PAD) x PT_TRI_FLOATS
    int* sorted_ids;              // n + PT_TRI_PAD
};

// Direction grid (pt_grid.hip): for rays that all start at the camera, cube-map cells -> triangles a direction of the
// cell could hit.
#define PT_GRID_EMPTY 0xffffffffu
#define PT_PROBE_SAMPLES 8  // samples a long dispatch without tile costs renders first, to have costs for the rest
struct PtGridBuild {
    int32_t res;
    uint32_t* cell_count;   // 6*res*res + 1: counts during the build, fill cursors afterwards
    uint32_t* cell_start;   // 6*res*res + 1: prefix sums of the list space of the cells (build only)
    uint32_t* cells;        // 6*res*res + 1: what a query reads (PT_GRID_EMPTY, the one triangle, or 1<<31 | list position)
    uint32_t* list;         // triangle indices
    uint32_t* occ;          // one bit per 4x4 block of cells: some cell of the block holds a triangle (PT_GRID_OCC_WORDS(res) words)
    size_t list_capacity;   // entries
    void* scan_tmp;
    size_t scan_tmp_bytes;
    uint32_t* big_jobs;     // queue of (triangle, face) jobs that cover many cells: one workgroup each
    uint32_t* big_cells;    // ... and how many cells each covers (its bounding box)
    uint32_t* big_count;    // 1
    uint32_t big_capacity;
};
size_t pt_grid_scan_temp_bytes(size_t cells);
// Counts the entries of every cell; afterwards cell_start[6*res*res] = the list length needed and summary[0..2] = an upper
// bound of the list length (bounding-box cells of every triangle and face), the (cell, triangle) entries, the list words
// needed, summary[3] = the 4x4 blocks of cells that hold no triangle at all (g.occ is written too). A bound of PT_GRID_MAX_LIST_BOUND or more means the resolution is too fine for this mesh: the large jobs were
// then not walked and the counts are not usable. slots: 2 * PT_GRID_BOUND_SLOTS words of device scratch.
#define PT_GRID_OCC_EDGE(res) (((res) + 3) / 4)
#define PT_GRID_OCC_WORDS(res) (((size_t)6 * PT_GRID_OCC_EDGE(res) * PT_GRID_OCC_EDGE(res) + 63) / 64 * 2)
#define PT_GRID_MAX_LIST_BOUND (1ull << 30)
#define PT_GRID_BOUND_SLOTS 256
hipError_t pt_grid_count(const PtGridBuild& g, const float* tris, int32_t n, unsigned long long* slots, unsigned long long* summary,
                         hipStream_t stream);
// fills cells and list (capacity must cover cell_start[6*res*res])
hipError_t pt_grid_fill(const PtGridBuild& g, const float* tris, int32_t n, hipStream_t stream);

size_t pt_bvh_sort_temp_bytes(int32_t n);
// Morton codes, sort, hierarchy, boxes. Synchronises the stream (a handful of times) to learn when the refit is done.
hipError_t pt_bvh_build_topology(const PtBvhBuild& b, hipStream_t stream, int32_t* out_passes);       // Morton codes, Karras hierarchy
hipError_t pt_bvh_build_topology_ploc(const PtBvhBuild& b, hipStream_t stream, int32_t* out_passes);  // clusters merged along the Morton curve
hipError_t pt_bvh_build_topology_binned(const PtBvhBuild& b, hipStream_t stream, int32_t* out_passes);  // the host builder's rules on the device (default)
hipError_t pt_bvh_build_topology_sah(const PtBvhBuild& b, hipStream_t stream, int32_t* out_passes);   // host-side builder, same arrays
// Traversal records for ray origin `ro` + triangle records/ids in leaf order (from the per-triangle records `tris`).
hipError_t pt_bvh_emit(const PtBvhBuild& b, const float ro[3], const float* tris, const float* tris_abs, hipStream_t stream);

// Work unit of the render kernels: one workgroup renders a PT_TILE_DIM x PT_TILE_DIM pixel tile with PT_TILE_WAVES waves
// (each wave an 8x8 block of it). With PT_TILE_WAVES = 1 every wave is its own workgroup: waves share nothing, and a
// wave slot is refilled as soon as ONE wave ends instead of when four do (measured +6 % on the 36-triangle scene,
// +2 % elsewhere, against 4-wave 16x16 tiles).
#ifndef PT_TILE_WAVES
#define PT_TILE_WAVES 1
#endif
#if PT_TILE_WAVES == 4
#define PT_TILE_DIM 16
#elif PT_TILE_WAVES == 1
#define PT_TILE_DIM 8
#else
#error "PT_TILE_WAVES must be 1 or 4"
#endif
#define PT_TILE_THREADS (64 * PT_TILE_WAVES)
#define PT_TILES_ACROSS(n) (((n) + PT_TILE_DIM - 1) / PT_TILE_DIM)

// One block's work: `sample_count` consecutive samples (from the dispatch's sample_begin-th on) of one tile. A tile
// with scratch_slot < 0 has a single item that folds its samples into the image itself; the items of a split tile write
// their samples to the scratch slot and a second kernel folds them in sample order.
#define PT_NO_TILE 0xffffffffu  // PtWorkItem::tile of a list position that holds no work
struct PtWorkItem {
    uint32_t tile;
    uint32_t sample_begin;
    uint32_t sample_count;
    int32_t scratch_slot;
};

struct PtKernelParams {
    float ro[3], llc[3], hor[3], ver[3];
    int32_t width, height;              // full frame: textureSize(u_Texture, 0)
    int32_t win_x0, win_y0, win_w, win_h;
    int32_t row_block, row_index, row_count;  // row_count > 1: only rows whose block (y / row_block) is row_index modulo row_count are rendered
    int32_t num_triangles, depth;
    int32_t first_sample, num_samples, sample_stride;
    int32_t accum_mode;                 // PtAccumMode
    int32_t normal_mode;                // PtNormalMode
    const float* tris;                  // num_triangles x PT_TRI_FLOATS
    const float* vnormals;              // num_triangles x 9: the three vertex normals (na, nb, nc)
    const float* gnormals;              // num_triangles x 3: normalize(cross(e1, e2))
    const float* cull;                  // num_triangles x PT_CULL_FLOATS, or null = no wave-level culling
    const uint32_t* grid_cells;         // direction grid: one word per cell (see PtGridBuild::cells), or null
    const uint32_t* grid_list;          // triangle indices of the cells with two or more (last one of a cell: top bit set)
    const uint32_t* grid_occ;           // one bit per 4x4 block of cells (0 = no triangle in the block), or null = not consulted
    const uint32_t* grid_occ_flag;      // device word: nonzero = queries look at grid_occ first (pt_launch_grid_occ_flag keeps it)
    int32_t grid_carry;                 // nonzero: the render loop tests one triangle per trip and carries the rest of a cell's list into the next
                                        // (pt_render_kernel<7, false>; the host picks the form that was quicker on this view)
    const float* grid_tris;             // per triangle: e1, e2, ap, r, e2r, geometric normal (origin-relative)
    int32_t grid_res;
    const uint32_t* live_count;         // device scalar: records in `tris` when it is the compacted array; null = num_triangles
    const float* tris_abs;              // num_triangles x PT_TRI_FLOATS: a, e1, e2 (for rays with their own origin)
    const float* materials;             // num_materials x (r, g, b, e)
    const int32_t* material_ids;        // per triangle
    int32_t shading_mode;               // PtShadingMode
    int32_t primary_reuse;              // 1: evaluate a pixel's camera ray once per block instead of once per sample
    const float* bvh_nodes;             // LBVH traversal records, or null = test every triangle in index order
    const float* bvh_tris;              // triangle records in leaf order
    const int32_t* bvh_ids;             // triangle id of each leaf-order record
    const float* bvh_nodes_abs;         // LBVH records with absolute boxes (rays with their own origin)
    int32_t lane_walk;                  // one hierarchy walk per lane (incoherent rays) instead of one per wave: 1 = lanes decoupled
                                        // (pt_render_decoupled_kernel), 2 = all lanes of a wave in step (pt_render_kernel<5, true>)
    int32_t lane_stack_limit;           // entries of a lane's stack that may be used (<= PT_LANE_STACK)
    const float* bvh_tris_abs;          // a, e1, e2 records in leaf order
    const uint32_t* bvh_nodes_w;        // wide nodes, compact records and frame of the per-lane walks (PT_BVH_WNODE_WORDS, PT_BVH_QTRI_FLOATS)
    const float* bvh_tris_q;
    const float* bvh_qframe;
    const float* spheres;               // 10 x 8 floats: per-sphere records for the current camera position (pt_prepare_spheres_kernel)
    const float* src;                   // RGBA32F image read (previous result)
    float* dst;                         // RGBA32F image written
    unsigned long long* counters;       // PT_COUNTER_SLOTS x PT_COUNTER_STRIDE: per slot [0] rays, [1] lane slots spent (wave loop trips x 64),
                                        // [2] node records, [3] triangle records fetched by the hierarchy walk, [4] queries actually evaluated,
                                        // [5] per-lane triangle tests / walk steps, [6] hits shaded, [7] records visited and [8] tests past the
                                        // back-face skip by the all-triangles loop (both per wave)
    const PtWorkItem* items;            // work list (block -> item), or null = block b renders tile b with all samples
    const uint32_t* sched_counts;       // [0] number of items, [1] number of split tiles (device scalars written by the scheduler)
    const uint32_t* split_tiles;        // scratch slot -> tile
    float* sample_scratch;              // slots x num_samples x 64 lanes x (throughput, direction y | materials mode: r, g, b, -):
                                        // samples of split tiles, folded afterwards
    int32_t max_items, max_split_tiles; // launch sizes (upper bounds of the two counts)
    uint32_t* tile_cost;                // per tile: loop trips spent on it in this dispatch (scheduling hint), or null
    const uint32_t* pixel_list;         // sparse views: tile t, lane l render the pixel pixel_list[64 t + l] = y << 16 | x (PT_NO_PIXEL: none) instead of
                                        // the 8x8 block t of the window: the pixels whose camera ray hits the mesh, gathered in block order
    int32_t pixel_tiles;                // ... and how many such tiles there are
    float4* primary_cache;              // optional, one per pixel of the target: the camera ray's hit (normal.xyz, triangle index bits; < 0 = miss)
    int32_t primary_cache_valid;        // 1: read it; 0: evaluate the camera ray and write it
    int32_t* out_ids;                   // primary pass only
    float* out_t;                       // primary pass only
    const float* raw_spheres;           // PT_PROGRAM_CPU_TRACER: the 10 spheres as uploaded (cx, cy, cz, r)
    float cpu_bounds[6];                // PT_PROGRAM_CPU_TRACER: the mesh's bounding box, min then max (Mesh::bounds, PathTracer.cpp:1180-1189)
    float cpu_max_t;                    // PT_PROGRAM_CPU_TRACER: far - near, the camera ray's maxt (PathTracer.cpp:745)
    uint32_t cpu_seed;                  // PT_PROGRAM_CPU_TRACER: seed of the counter-based generator
    const float* verify_tris;           // self-check only: the records of ALL triangles in index order (never compacted)
    unsigned long long* verify_out;     // self-check only: [0] queries compared, [1] ids differing, [2] distances differing
    int32_t verify_bounces;             // self-check only: generations of bounce rays compared after the camera rays
};

// Launchers (defined in pt_kernels.hip). All are asynchronous on `stream`.
hipError_t pt_launch_prepare_triangles(const float* vertices, const uint32_t* indices, int32_t num_triangles,
                                       const float ro[3], float* tris, float* vnormals, float* gnormals, float* cull,
                                       float* tris_abs, float* tris_grid, int write_static, hipStream_t stream);
// recs (10 x 8 floats) <- the direction-independent terms of the sphere test for ray origin ro
hipError_t pt_launch_prepare_spheres(const float* spheres, const float ro[3], float* recs, hipStream_t stream);
// live <- the records of `tris` that any camera ray can hit at all (e2r > 0), order kept, zero padding after; *live_count <- how many.
// tmp: pt_compact_temp_words(num_triangles) words of device scratch.
size_t pt_compact_temp_words(int32_t num_triangles);
hipError_t pt_launch_compact_triangles(const float* tris, int32_t num_triangles, float* live, uint32_t* live_count, uint32_t* tmp, hipStream_t stream);
// bad[0..2] += inputs for which the short reciprocal / square root / log2 quotient differ from the IEEE results (see the kernel)
hipError_t pt_launch_arith_selftest(unsigned long long* bad, hipStream_t stream);
hipError_t pt_launch_fill_rgba(float* img, size_t num_pixels, float r, float g, float b, float a, hipStream_t stream);
hipError_t pt_launch_render(const PtKernelParams& p, int program_kind, hipStream_t stream);
hipError_t pt_launch_present(const float* img, size_t num_pixels, uint32_t* out_rgba8, hipStream_t stream);
// Work list for the next dispatch from the per-tile costs of the previous one: tiles whose cost exceeds
// total / max_split_tiles are split into several items (at most num_samples), items ordered most expensive first.
// Always: items <= num_tiles + max_split_tiles and split tiles <= max_split_tiles. tile_cost is only read.
// sort = 0 keeps the split but leaves the list in tile order. groups = 1: one list. groups = 8: the tiles are cut into eight
// runs of equal cost (bands of the image) and run g's items stand at the list positions g, g + 8, ... -- the blocks one XCD
// takes -- with PT_NO_TILE entries where a run is shorter than the longest; the list then has 8 x the longest run's items
// positions (at most 8 x (num_tiles + max_split_tiles)).
// sched_counts <- {list positions, split tiles}.
hipError_t pt_launch_schedule_items(const uint32_t* tile_cost, PtWorkItem* items, uint32_t* sched_counts, uint32_t* split_tiles, uint32_t* sched_tmp,
                                    int num_tiles, int num_samples, int max_split_tiles, int sort, int groups, hipStream_t stream);
#define PT_SCHED_TMP_WORDS (16 + 2 * 64 + 64 * 8 * 64)  // scratch of the scheduler's launches: totals, per-workgroup share costs and bucket counts
// *flag <- whether the direction grid's queries should consult the block bits, from the statistics counters (see the kernel)
hipError_t pt_launch_grid_occ_flag(const unsigned long long* counters, unsigned long long samples, uint32_t* flag, hipStream_t stream);
hipError_t pt_launch_primary(const PtKernelParams& p, int program_kind, hipStream_t stream);
// Sparse views (see PtKernelParams::pixel_list). `p` describes the view and its direction grid (pixel_list null).
// pt_launch_pixel_list: mask[b] <- which pixels of the window's 8x8 block b have a camera ray that hits the mesh (bit l = lane l),
// list <- those pixels block after block, padded with PT_NO_PIXEL to a multiple of 64 (`list` holds one word per window pixel,
// rounded up to 64), total[0] <- how many, total[1] <- how many blocks hold at least one. counts / offsets: one word per block each (scratch).
// pt_launch_sky_fold: what the render kernels do for the window's OTHER pixels (every sample is the sky colour of the camera
// ray): folds p.num_samples samples from p.src into p.dst, counts their rays, fills their primary-cache entries when asked.
#define PT_NO_PIXEL 0xffffffffu
hipError_t pt_launch_pixel_list(const PtKernelParams& p, unsigned long long* mask, uint32_t* counts, uint32_t* offsets, uint32_t* list, uint32_t* total,
                                hipStream_t stream);
hipError_t pt_launch_sky_fold(const PtKernelParams& p, const unsigned long long* mask, hipStream_t stream);
// every camera ray (and p.verify_bounces generations of bounce rays) through the structure `p` selects AND through the loop
// over all of p.verify_tris; counts into p.verify_out. hipErrorInvalidValue when `p` selects no structure to check.
hipError_t pt_launch_verify(const PtKernelParams& p, hipStream_t stream);
