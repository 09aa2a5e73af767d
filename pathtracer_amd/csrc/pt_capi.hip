// C ABI (include/pt_hip.h) over the HIP runtime. No exceptions leave this file; every failure becomes a
// PtStatus plus a thread-local message.
#include "../../include/pt_hip.h"
#include "pt_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

// ---- scratch of pt_dispatch_static (the dispatch that allocates nothing) -------------------------------------------------
// One array per GPU, part of the library's device image, held by every device the library runs on until the process ends.
// Layout (byte offsets): per-tile costs | work list | scratch slot -> tile | the scheduler's two counts | the scheduler's
// bucket counts | parked samples of split tiles. In host code the array's NAME is only a handle for hipGetSymbolAddress;
// the device address it has on each GPU is looked up once per device in pt_create and kept in g_scratch_ptr.
#define PT_STATIC_MAX_TILES 131072u  // 8x8-pixel tiles of the window: 3840x2160 has 129,600
#define PT_STATIC_MAX_SPLIT 8192u    // tiles whose samples several blocks share (= the wave slots of the device)
#define PT_STATIC_SCRATCH_BYTES ((size_t)64 << 20)
static __device__ __align__(256) unsigned char scratch[PT_STATIC_SCRATCH_BYTES];

namespace {

constexpr size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr size_t kStaticCostOff = 0;
constexpr size_t kStaticItemsOff = align256(kStaticCostOff + (size_t)PT_STATIC_MAX_TILES * sizeof(uint32_t));
constexpr size_t kStaticSplitOff = align256(kStaticItemsOff + ((size_t)PT_STATIC_MAX_TILES + PT_STATIC_MAX_SPLIT) * sizeof(PtWorkItem));
constexpr size_t kStaticCountsOff = align256(kStaticSplitOff + (size_t)PT_STATIC_MAX_SPLIT * sizeof(uint32_t));
constexpr size_t kStaticTmpOff = align256(kStaticCountsOff + 2 * sizeof(uint32_t));
constexpr size_t kStaticSamplesOff = align256(kStaticTmpOff + (size_t)PT_SCHED_TMP_WORDS * sizeof(uint32_t));
static_assert(kStaticSamplesOff < PT_STATIC_SCRATCH_BYTES / 8, "the parked samples get most of the array");
constexpr size_t kStaticSampleBytes = PT_STATIC_SCRATCH_BYTES - kStaticSamplesOff;

constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices] = {};  // device address of `scratch` on each GPU (each holds its own copy)
std::mutex g_scratch_mutex;
// Who used a GPU's scratch array last: the context and the event it recorded behind its launches. Another context's
// pt_dispatch_static on the same GPU is refused while that event has not completed (the array is shared, see pt_hip.h).
struct ScratchUse {
    PtContext* owner = nullptr;
    hipEvent_t done = nullptr;
    bool issuing = false;   // the owner is inside pt_dispatch_static right now (its event is not recorded yet)
};
ScratchUse g_scratch_use[kMaxDevices];

// the modes in which Shade() keeps the fetched material and moves the ray origin
inline bool is_materials(PtShadingMode m) { return m == PT_SHADE_MATERIALS || m == PT_SHADE_MATERIALS_UNIT; }

thread_local char g_error[512] = "";

PtStatus fail(PtStatus code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return code;
}

#define PT_HIP_TRY(expr)                                                                                     \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(PT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));        \
    } while (0)

#define PT_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) return fail(code, __VA_ARGS__); \
    } while (0)

struct Program {
    uint32_t handle;
    PtProgramKind kind;
};

}  // namespace

struct PtContext {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    std::vector<Program> programs;
    uint32_t next_program = 1;
    uint32_t current_program = 0;

    // scene
    float* d_spheres = nullptr;
    float* d_sphere_recs = nullptr;   // 10 x 8: direction-independent terms of the sphere test for sphere_recs_ro
    bool sphere_recs_valid = false;
    float sphere_recs_ro[3] = {0, 0, 0};
    float* d_materials = nullptr;
    int32_t num_materials = 0;
    bool have_spheres = false;
    float* d_vertices = nullptr;    // raw upload, reference layout (6 floats per vertex)
    uint32_t* d_indices = nullptr;  // raw upload
    int32_t num_vertices = 0, num_triangles = 0;
    float* d_tris = nullptr;        // SoA records, rebuilt when the mesh or the camera position changes
    float* d_tris_live = nullptr;   // the records a camera ray can hit at all, compacted (all-triangles loop)
    uint32_t* d_live_count = nullptr;
    float* d_vnormals = nullptr;
    float* d_gnormals = nullptr;
    float* d_cull = nullptr;
    float* d_tris_abs = nullptr;    // a, e1, e2 records (origin-independent)
    float* d_tris_grid = nullptr;   // records of the direction-grid query (geometric normal in place of the index)
    int32_t* d_material_ids = nullptr;
    int32_t num_material_ids = 0;
    int32_t max_material_id = -1;
    float mesh_extent = 0.f;        // largest absolute vertex coordinate
    float mesh_bounds[6] = {0, 0, 0, 0, 0, 0};  // min, max over the vertex positions (Mesh::bounds of the reference's CPU tracer)
    uint32_t cpu_seed = 1;          // PT_PROGRAM_CPU_TRACER
    float cpu_max_t = 100.f - 0.1f;
    PtShadingMode shading_mode = PT_SHADE_REFERENCE;
    bool records_valid = false;
    bool static_records_valid = false;  // normals and origin-independent records: written after an upload, kept over camera moves
    bool cull_valid = false;            // the cull records match records_ro
    bool live_valid = false, bvh_emit_valid = false, grid_valid = false;  // derived from the records, refreshed on use
    float records_ro[3] = {0, 0, 0};
    // LBVH (optional)
    PtBvhBuild bvh{};
    bool bvh_built = false;
    bool bvh_trivial = false;        // pt_build_accel ran on a mesh small enough to be one leaf (no hierarchy kept)
    int32_t bvh_refit_passes = 0;
    PtAccelMode accel_mode = PT_ACCEL_AUTO;
    PtBvhBuilder bvh_builder = PT_BVH_BUILDER_BINNED;  // PT_BVH_BUILDER=binned / ploc / morton / sah sets a context's initial choice
    // direction grid (optional, rebuilt with the records when the camera position changes)
    PtGridBuild grid{};
    PtGridBuild grid_spare{};   // the arrays of the other resolution (coarse while the fine grid is in use and the other way round),
                                // kept so that going back and forth between the two allocates nothing
    bool grid_want_fine = false;   // this camera position has gathered enough samples for the fine grid to pay (pt_dispatch decides)
    bool grid_fine_over_move = false;  // set by pt_dispatch around its own update of the records (see ensure_records)
    double grid_fine_after = 2.0e9;    // pixel-samples at one camera position from which the fine grid is used (PT_GRID_FINE_AFTER=x
                                       // sets it, for tests and measurements; negative: never)
    double px_samples_here = 0.0;  // pixel-samples dispatched since the mesh or the camera position last changed
    bool grid_wanted = false;   // pt_build_accel was called for this mesh
    bool grid_built = false;
    int32_t grid_res_override = 0;
    int32_t grid_queue_override = 0;  // PT_GRID_QUEUE=n (testing): capacity of the grid build's queue of large jobs
    bool grid_res_reduced = false;  // the preferred resolution gave lists too long for this mesh
    uint64_t grid_entries = 0;
    bool grid_use_occ = false;       // the block bits are worth a look: a quarter of the blocks or more are empty (build_grid)
    uint32_t* d_grid_occ_flag = nullptr;  // ... and the view's queries mostly miss (device word, kept up to date with the work list)
    int grid_occ_override = -1;      // PT_GRID_OCC=0/1 (experiments)
    // The direction-grid kernel has two forms of its loop (pt_render_kernel<4 / 7, false>: the wave runs its list loop as often as its
    // longest list / a lane tests one triangle per trip and carries the rest of its list into the next). Same image; which is
    // quicker depends on the view (Cornell and the 1M triangles: the second by 6-11 %; the two heads: the first by 4-9 %) in a way
    // no statistic of the queries predicted, so a view's first long dispatches with a work list time one form each (kernel time per
    // sample, two events around the launch) and the quicker one is kept until the grid is rebuilt.
    int grid_carry_override = -1;    // PT_GRID_CARRY=0/1 (experiments, tests): never / always the second form
    bool grid_carry = false;         // the form in use
    bool last_grid_carry = false;    // ... by the last dispatch that went through the grid (PtStats::grid_lists_carried)
    int carry_trial = 0;             // 0 none yet, 1 first form being timed, 2 second form run once untimed (a kernel's first launch pays for
                                     // its loading), 3 second form being timed, 4 decided
    double carry_trial_ms[2] = {0.0, 0.0};   // kernel ms per sample of the two forms
    int carry_trial_samples = 0;     // samples of the dispatch being timed
    hipEvent_t ev_trial[2] = {nullptr, nullptr};
    unsigned long long* d_grid_count_scratch = nullptr;   // scratch of the grid's count pass and its three-word summary
    unsigned long long* h_grid_summary = nullptr;    // pinned host copy of the summary

    // target
    float* d_image[2] = {nullptr, nullptr};
    bool image_owned = false;
    int32_t width = 0, height = 0;
    int current = 0;  // image the NEXT pass writes is image[current]; the last written is image[current ^ 1]
    int32_t win[4] = {0, 0, 0, 0};
    PtAccumMode accum_mode = PT_ACCUM_MEAN;
    PtNormalMode normal_mode = PT_NORMALS_VERTEX;
    int32_t* d_ids = nullptr;
    float* d_t = nullptr;
    // tile scheduling: per-tile cost of the last dispatch and the order derived from it
    uint32_t* d_tile_cost = nullptr;
    PtWorkItem* d_items = nullptr;        // tiles + max_split entries
    uint32_t* d_sched_counts = nullptr;   // 2
    uint32_t* d_sched_tmp = nullptr;      // PT_SCHED_TMP_WORDS
    uint32_t* d_split_tiles = nullptr;    // max_split entries
    float* d_sample_scratch = nullptr;    // grows with the largest num_samples dispatched
    size_t sample_scratch_bytes = 0;
    int32_t wave_slots = 0;               // waves the device holds at once (CUs x 4 SIMDs x 8)
    float split_factor = -1.f;            // < 0: chosen per kernel family, see pt_dispatch
    bool primary_reuse = true;
    int32_t lane_stack_limit = PT_LANE_STACK;  // PT_LANE_STACK_LIMIT=n (testing): a lane hands over to the per-wave walk earlier
    int lane_walk = 1;                    // materials mode: 1 = one hierarchy walk per lane, lanes decoupled; PT_LANE_WALK=2: per lane, all
                                          // lanes of a wave in step; PT_LANE_WALK=0: one walk per wave
    bool auto_grid = true;                // PT_AUTO_GRID=0: PT_ACCEL_AUTO never picks the direction grid (for comparisons)
    bool camera_ray_grid = true;          // PT_CAMERA_RAY_GRID=0: the materials mode walks the hierarchy for its camera rays too
    uint32_t dispatches_here = 0;         // dispatches finished since the mesh or the camera position last changed
    bool keep_unreachable = false;        // PT_KEEP_UNREACHABLE=1: the loop walks all records (to measure the compaction)
    int32_t sched_win[4] = {-1, -1, -1, -1};  // window the costs were measured for
    bool sched_have_cost = false;
    float list_base_factor = -1.f;        // the list was built for this split factor (before the automatic step to the eager one)
    int list_entry_bytes = 0;             // bytes per parked sample the scratch was sized for
    int list_items = 0, list_split = 0;   // what the scheduler produced: items and split tiles (launch sizes)
    bool concentrated = false;            // the work of this view sits in few tiles: lists are built with the eager split factor
    int32_t row_block = 0, row_index = 0, row_count = 1;  // pt_set_row_interleave
    bool costs_stale = false;             // the tile costs were gathered for another mesh or camera position
    // the work list built last and what it was built for; reused while nothing it depends on changes
    bool list_valid = false;
    int32_t list_samples = 0, list_age = 0;
    int64_t cost_samples = 0;   // samples per pixel behind the costs gathered since the list was built
    int64_t list_basis = 0;     // samples per pixel behind the costs the list was built from
    // camera-ray hit per pixel (normal, triangle), kept from one dispatch to the next of the same view: the shader has no
    // pixel jitter, so a frame loop of one sample per dispatch would evaluate the same query every frame
    float4* d_primary_cache = nullptr;
    bool primary_cache_enabled = true;   // PT_PRIMARY_CACHE=0 turns it off (for comparisons)
    bool primary_cache_valid = false;
    struct PrimaryCacheKey {
        float camera[12];
        int32_t num_triangles, win[4], rows[3], normal_mode, accel_mode;
        uint32_t program;
    } primary_cache_key{};
    // Sparse views: the pixels whose camera ray hits the mesh, gathered block after block into tiles of 64 (PtKernelParams::pixel_list),
    // so that the sky pixels of blocks on a silhouette do not occupy lanes; the other pixels are folded by pt_sky_fold_kernel.
    // Built per view (camera, window, row partition, mesh) from one query of the direction grid per pixel.
    uint32_t* d_pixel_list = nullptr;              // one word per target pixel, rounded up to 64
    unsigned long long* d_pixel_mask = nullptr;    // per 8x8 block of the window: which of its pixels are in the list
    uint32_t* d_pixel_counts = nullptr;            // per block (scratch of the build), then d_pixel_offsets; one word for the total
    uint32_t* d_pixel_offsets = nullptr;
    uint32_t* d_pixel_total = nullptr;
    int pixel_list_mode = -1;       // PT_PIXEL_LIST=0: never; =1: whenever a view has a direction grid; default: from a view's second
                                    // dispatch on, for dispatches of kPixelListMinSamples samples or more, when 1/32 or more of the lanes of the
                                    // blocks that hold its hit pixels would belong to sky pixels
    bool pixel_list_valid = false;
    int32_t pixel_tiles = 0;
    uint32_t pixel_hits = 0, pixel_blocks = 0;   // pixels that see the mesh, and the window's 8x8 blocks that hold them
    uint32_t pixel_list_id = 0;     // counts the builds (never 0 once built): what the tile costs and the work list refer to
    struct PixelListKey {
        float camera[12];
        int32_t win[4], rows[3], size[2], num_triangles;
        uint32_t grid_id;
    } pixel_list_key{}, last_view_key{};   // (last_view_key: what the previous pt_dispatch rendered, grid_id left 0)
    bool last_view_valid = false;
    uint32_t grid_id = 0;           // counts the builds of the direction grid
    bool use_pixel_list = false;    // what pt_dispatch decided for its launches
    bool last_gathered = false;     // ... and what the last dispatch did (PtStats::gathered_pixels)
    uint32_t sched_numbering = 0;   // 0: the tile costs refer to the window's 8x8 blocks; otherwise to the tiles of that pixel list
    int xcd_groups = 0;             // PT_XCD_GROUPS=1: the work list of multi-sample dispatches is dealt to the XCDs in image bands of
                                    // equal cost (an experiment that is kept: see kXcdGroups)
    int list_groups = 1;            // what the list in use was built with
    int sched_sort = 1;             // PT_SCHED_SORT=0: the work list keeps its split but is not ordered by cost
    int sched_mode = 2;             // 0 raster order, 1 cost-ordered tiles, 2 cost-ordered items with heavy tiles split by samples
    int32_t fail_grid_alloc = 0, grid_allocs = 0;  // PT_FAIL_GRID_ALLOC=k (testing): the k-th allocation of the grid build fails
    bool no_alloc = false;          // inside pt_dispatch_static: whatever would allocate device or host memory fails instead
    hipEvent_t ev_static = nullptr; // recorded behind the launches of every pt_dispatch_static of this context
    int fault_inject = 0;           // PT_FAULT_INJECT=grid|bvh (testing pt_verify_accel): the structure is damaged on purpose after it is built
    unsigned long long* d_verify_out = nullptr;   // three counters of pt_verify_accel
    uint32_t* d_compact_tmp = nullptr;            // per-workgroup counts of the record compaction

    // uniforms
    PtUniforms uniforms{};
    bool have_uniforms = false;

    // stats
    unsigned long long* d_counters = nullptr;  // [0] rays
    uint64_t samples = 0;
    uint32_t dispatches = 0;
    bool timing = false;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool ev_pending = false;
    float last_ms = 0.f;
    double total_ms = 0.0;
};

namespace {

constexpr size_t kCounterBytes = (size_t)PT_COUNTER_SLOTS * PT_COUNTER_STRIDE * sizeof(unsigned long long);
constexpr int PT_MAX_SPLIT_FACTOR = 32;  // arrays of the scheduler are sized for this many split tiles per wave slot
constexpr float kEagerSplitFactor = 8.0f;     // views whose work sits in few tiles (loop, direction grid)
constexpr float kWaveWalkSplitFactor = 32.0f; // the per-wave hierarchy walk of the reference shading
constexpr size_t kXcdGroups = 8;        // XCDs of the device: blocks go to them round-robin by block index
// PT_XCD_GROUPS=1 (off by default) deals the work list to the XCDs in image bands of equal cost, so that every XCD keeps reading
// the records of one part of the scene. Measured on the 1M-triangle scene, materials mode, 16 samples at 4K: L2 hit rate 0.57 ->
// 0.75 (129 M -> 74 M misses per dispatch) and the same 8.40 ms -- the per-lane walk is not bound by where its records come
// from -- while scenes that live in every L2 anyway lose 3-9 % to the coarser balance (36 triangles: 7.06 -> 7.70 ms).
// Split factor of throughput-bound kernels (loop, direction grid, per-lane walk) on a view whose work is spread out: a tile is
// split when its cost exceeds total / (factor x wave slots). Slightly below 1 (the kernels hold seven waves per SIMD where the
// slots count eight, and the list starts the expensive tiles first, so a tile a little above a slot's share still ends
// within the dispatch): on the 1080p Cornell box factor 1 split 686 tiles that sit right at the threshold and parked 170-230 MB of
// samples per dispatch for nothing -- same time at 1, 0.85 and 0.7 on every workload (tools/ab_bench.py, variants split1 and split07),
// no split tile and no parked sample at 0.85. PT_V_SPREAD_SPLIT sets it in experiment builds.
#ifdef PT_V_SPREAD_SPLIT
constexpr float kSpreadSplitFactor = PT_V_SPREAD_SPLIT;
#else
constexpr float kSpreadSplitFactor = 0.85f;
#endif

PtStatus bind(PtContext* ctx) {
    PT_REQUIRE(ctx != nullptr, PT_ERR_INVALID_ARGUMENT, "null context");
    PT_HIP_TRY(hipSetDevice(ctx->device));
    return PT_OK;
}

const Program* find_program(const PtContext* ctx, uint32_t handle) {
    for (const Program& p : ctx->programs)
        if (p.handle == handle) return &p;
    return nullptr;
}

PtStatus collect_timing(PtContext* ctx) {
    if (ctx->ev_pending) {
        PT_HIP_TRY(hipEventSynchronize(ctx->ev_stop));
        float ms = 0.f;
        PT_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
        ctx->last_ms = ms;
        ctx->total_ms += ms;
        ctx->ev_pending = false;
    }
    return PT_OK;
}

void free_target(PtContext* ctx) {
    if (ctx->image_owned) {
        if (ctx->d_image[0]) (void)hipFree(ctx->d_image[0]);
        if (ctx->d_image[1] && ctx->d_image[1] != ctx->d_image[0]) (void)hipFree(ctx->d_image[1]);
    }
    ctx->d_image[0] = ctx->d_image[1] = nullptr;
    ctx->image_owned = false;
    if (ctx->d_ids) (void)hipFree(ctx->d_ids);
    if (ctx->d_t) (void)hipFree(ctx->d_t);
    if (ctx->d_tile_cost) (void)hipFree(ctx->d_tile_cost);
    if (ctx->d_items) (void)hipFree(ctx->d_items);
    if (ctx->d_sched_counts) (void)hipFree(ctx->d_sched_counts);
    if (ctx->d_sched_tmp) (void)hipFree(ctx->d_sched_tmp);
    ctx->d_sched_tmp = nullptr;
    if (ctx->d_split_tiles) (void)hipFree(ctx->d_split_tiles);
    if (ctx->d_primary_cache) (void)hipFree(ctx->d_primary_cache);
    ctx->d_primary_cache = nullptr;
    ctx->primary_cache_valid = false;
    void* pixel_arrays[] = {ctx->d_pixel_list, ctx->d_pixel_mask, ctx->d_pixel_counts, ctx->d_pixel_offsets, ctx->d_pixel_total};
    for (void* q : pixel_arrays)
        if (q) (void)hipFree(q);
    ctx->d_pixel_list = ctx->d_pixel_counts = ctx->d_pixel_offsets = ctx->d_pixel_total = nullptr;
    ctx->d_pixel_mask = nullptr;
    ctx->pixel_list_valid = false;
    ctx->last_view_valid = false;
    ctx->use_pixel_list = false;
    ctx->sched_numbering = 0;
    if (ctx->d_sample_scratch) (void)hipFree(ctx->d_sample_scratch);
    ctx->d_items = nullptr;
    ctx->d_sched_counts = ctx->d_split_tiles = nullptr;
    ctx->d_sample_scratch = nullptr;
    ctx->sample_scratch_bytes = 0;
    ctx->d_ids = nullptr;
    ctx->d_t = nullptr;
    ctx->d_tile_cost = nullptr;
    ctx->sched_have_cost = false;
    ctx->width = ctx->height = 0;
}

void free_bvh(PtContext* ctx) {
    PtBvhBuild& b = ctx->bvh;
    void* ptrs[] = {b.tri_box, b.scene_box, b.keys_tmp, b.keys, b.order_tmp, b.order, b.sort_tmp, b.child, b.range, b.node_box, b.done, b.remaining,
                    b.nodes, b.sorted_tris, b.sorted_ids, b.nodes_abs, b.sorted_tris_abs, b.nodes_w, b.tris_q, b.qframe, b.wide_index, b.wide_child};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    b = PtBvhBuild{};
    ctx->bvh_built = false;
    ctx->bvh_trivial = false;
}

void free_grid_arrays(PtGridBuild& g) {
    void* ptrs[] = {g.cell_count, g.cell_start, g.cells, g.list, g.occ, g.scan_tmp, g.big_jobs, g.big_cells, g.big_count};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    g = PtGridBuild{};
}

// the grid in use (the spare set of arrays stays)
void free_grid(PtContext* ctx) {
    free_grid_arrays(ctx->grid);
    ctx->grid_built = false;
}

// hipMalloc for the grid's arrays. PT_FAIL_GRID_ALLOC=k (testing) makes the k-th of these calls in the context's life
// report out-of-memory, to exercise the clean-up paths.
template <typename T>
hipError_t grid_alloc(PtContext* ctx, T** ptr, size_t bytes) {
    if (ctx->fail_grid_alloc > 0 && ++ctx->grid_allocs == ctx->fail_grid_alloc) return hipErrorOutOfMemory;
    return hipMalloc(reinterpret_cast<void**>(ptr), bytes);
}

// Cube-map resolution per face, within [768, 2048]
int grid_resolution(const PtContext* ctx) {
    if (ctx->grid_res_override > 0) return ctx->grid_res_override;
    // Measured (ms per 256-spp dispatch at 512 / 768 / 1024 / 1536 cells per face edge, static camera): finer cells keep paying
    // (fewer triangle tests per query) until the cell table's misses cost more -- 36 triangles in a closed box, where every query
    // fetches a cell word: 7.33 / 7.23 / 7.36 / 7.36; 3,936 triangles in the open: 1.45 / 1.37 / 1.32 / 1.28; 968 and 64 triangles:
    // no difference; 98,400 triangles (25 copies of the 3,936) at 1024 / 1536 / 2048: 0.56 / 0.51 / 0.50, 251,904: 0.83 / 0.73 /
    // 0.67, 31,488: no difference; a million: 2048, the largest offered. A camera move rebuilds the grid, and that grows with the
    // cell count (0.25 / 0.29 / 0.36 ms for the small meshes), so the small meshes stop at 768 and 1024; 2048 from about 65,000
    // triangles on (96 triangles' worth of cells over the sphere).
    int res = ctx->num_triangles >= 2000 ? 1024 : 768;
    while (res < 2048 && (double)6 * res * res < 96.0 * (double)ctx->num_triangles) res *= 2;
    return res;
}

// Resolution for a view that stays: a finer grid keeps paying on a large mesh (1M triangles at 4K, ms per 64-spp dispatch at
// 2048 / 3072 / 4096 / 8192: 1.61 / 1.36 / 1.28 / 1.19), but its build grows with the cell count (first frame after a camera
// move: 0.97 / 1.44 / 2.10 ms at 2048 / 3072 / 4096), and on meshes of 100-500 thousand triangles the gain per sample is a
// quarter of that. So 4096 serves large meshes only, and only once a camera position has gathered the pixel-samples that repay
// the longer build (pt_dispatch decides, PtContext::grid_fine_after); a camera that keeps moving stays with 2048.
constexpr int32_t kFineGridMinTriangles = 750000;   // (PtContext::grid_fine_after: 2e9 pixel-samples, a 4K frame's 240 samples per pixel)
int grid_fine_resolution(const PtContext* ctx) {
    const int coarse = grid_resolution(ctx);
    if (ctx->grid_res_override > 0 || coarse < 2048 || ctx->num_triangles < kFineGridMinTriangles) return coarse;
    return 4096;
}

PtStatus build_grid(PtContext* ctx) {
    ctx->grid_built = false;
    // Resolution: the preferred one, halved while the lists could exceed 2^30 entries (4 GiB) -- many triangles that each
    // cover a large part of a fine grid. Below 32 cells per face edge the grid is given up; the other search modes remain.
    if (!ctx->d_grid_count_scratch || !ctx->h_grid_summary) {
        PT_REQUIRE(!ctx->no_alloc, PT_ERR_STATE, "the direction grid has never been built for this context: render one frame with pt_dispatch first");
        if (!ctx->d_grid_count_scratch) PT_HIP_TRY(hipMalloc(&ctx->d_grid_count_scratch, (2 * PT_GRID_BOUND_SLOTS + 4) * sizeof(unsigned long long)));
        if (!ctx->h_grid_summary) PT_HIP_TRY(hipHostMalloc(&ctx->h_grid_summary, 4 * sizeof(unsigned long long)));  // pinned: one short copy per build
    }
    PtGridBuild& g = ctx->grid;
    uint32_t total = 0, entries = 0;  // list words needed (cells with two or more triangles); all (cell, triangle) entries
    size_t cells = 0;
    // (a mesh that needed a coarser grid once keeps it until the next upload, so camera moves do not reallocate)
    const int preferred = ctx->grid_want_fine ? grid_fine_resolution(ctx) : grid_resolution(ctx);
    for (int res = (ctx->grid_res_reduced && g.res > 0) ? g.res : preferred;; res /= 2, ctx->grid_res_reduced = true) {
        if (res < 32) return PT_OK;
        cells = (size_t)6 * res * res;
        if (g.res != res && ctx->grid_spare.res == res) std::swap(g, ctx->grid_spare);  // the other resolution's arrays are at hand
        if (g.res != res) {
            PT_REQUIRE(!ctx->no_alloc, PT_ERR_STATE, "the direction grid must be rebuilt at %d cells per face edge (it has %d), which allocates: "
                       "render one frame of this view with pt_dispatch first", res, g.res);
            // (coarse and fine of one mesh: the set not in use is kept instead of freed; a mesh whose lists forced a coarser
            // grid than preferred has one resolution only)
            if (g.res != 0 && ctx->grid_spare.res == 0 && !ctx->grid_res_reduced && grid_fine_resolution(ctx) != grid_resolution(ctx))
                std::swap(g, ctx->grid_spare);
            free_grid(ctx);
            g.res = res;
            g.scan_tmp_bytes = pt_grid_scan_temp_bytes(cells);
            g.big_capacity = (uint32_t)std::min<size_t>((size_t)ctx->num_triangles * 6, (size_t)4 << 20);
            if (ctx->grid_queue_override > 0) g.big_capacity = (uint32_t)ctx->grid_queue_override;
            hipError_t e = grid_alloc(ctx, &g.cell_count, (cells + 1) * sizeof(uint32_t));
            if (e == hipSuccess) e = grid_alloc(ctx, &g.cell_start, (cells + 1) * sizeof(uint32_t));
            if (e == hipSuccess) e = grid_alloc(ctx, &g.cells, (cells + 1) * sizeof(uint32_t));
            if (e == hipSuccess) e = grid_alloc(ctx, &g.occ, PT_GRID_OCC_WORDS(res) * sizeof(uint32_t));
            if (e == hipSuccess) e = grid_alloc(ctx, &g.scan_tmp, g.scan_tmp_bytes ? g.scan_tmp_bytes : 16);
            if (e == hipSuccess) e = grid_alloc(ctx, &g.big_jobs, (size_t)g.big_capacity * sizeof(uint32_t));
            if (e == hipSuccess) e = grid_alloc(ctx, &g.big_cells, (size_t)g.big_capacity * sizeof(uint32_t));
            if (e == hipSuccess) e = grid_alloc(ctx, &g.big_count, sizeof(uint32_t));
            if (e != hipSuccess) {
                // no half-made grid: res goes back to 0, so the next build starts with the allocations again
                free_grid(ctx);
                return fail(PT_ERR_HIP, "direction grid: allocation failed at %d cells per face edge: %s", res, hipGetErrorString(e));
            }
        }
        // the count pass also sums the bound and stops short of the large jobs when it is exceeded
        unsigned long long* summary = ctx->d_grid_count_scratch + 2 * PT_GRID_BOUND_SLOTS;
        PT_HIP_TRY(pt_grid_count(g, ctx->d_tris, ctx->num_triangles, ctx->d_grid_count_scratch, summary, ctx->stream));
        PT_HIP_TRY(hipMemcpyAsync(ctx->h_grid_summary, summary, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
        const unsigned long long bound = ctx->h_grid_summary[0];
        entries = (uint32_t)std::min<unsigned long long>(ctx->h_grid_summary[1], 0xffffffffull);
        total = (uint32_t)ctx->h_grid_summary[2];
        if (bound < PT_GRID_MAX_LIST_BOUND) break;
    }
    if ((size_t)total > g.list_capacity) {
        PT_REQUIRE(!ctx->no_alloc, PT_ERR_STATE, "the direction grid of this camera position needs %u list entries and %zu are allocated: "
                   "render one frame of this view with pt_dispatch first", total, g.list_capacity);
        if (g.list) (void)hipFree(g.list);
        g.list = nullptr;
        g.list_capacity = 0;
        const size_t cap = (size_t)total + (size_t)total / 8 + 64;
        hipError_t e = grid_alloc(ctx, &g.list, cap * sizeof(uint32_t));
        if (e != hipSuccess) return fail(PT_ERR_HIP, "direction grid: allocating %zu list entries failed: %s", cap, hipGetErrorString(e));
        g.list_capacity = cap;
    }
    PT_HIP_TRY(pt_grid_fill(g, ctx->d_tris, ctx->num_triangles, ctx->stream));
    if (ctx->fault_inject == 1) {
        // testing pt_verify_accel: a sixteenth of the rows of cells, through the middle of the +z face, forget their triangles
        const size_t rows = (size_t)std::max(2, g.res / 16);
        const size_t first = ((size_t)4 * g.res + (size_t)g.res / 2 - rows / 2) * g.res;
        PT_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(g.cells + first), (int)PT_GRID_EMPTY, rows * g.res, ctx->stream));
    }
    {
        // Queries consult the block bits where they can end many of them early: a quarter of the blocks empty, or more.
        // (Cornell, a closed box: none, and the look-up would only lengthen every query.)
        const unsigned long long edge = PT_GRID_OCC_EDGE(g.res), blocks = 6ull * edge * edge;
        ctx->grid_use_occ = ctx->grid_occ_override >= 0 ? ctx->grid_occ_override != 0 : 4ull * ctx->h_grid_summary[3] >= blocks;
        // until the view's own statistics say otherwise (PT_GRID_OCC=1 pins it)
        PT_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->d_grid_occ_flag), 1u, 1, ctx->stream));
        // (a new grid: which form of the render loop suits the view is found out again)
        ctx->grid_carry = ctx->grid_carry_override > 0;
        ctx->carry_trial = ctx->grid_carry_override >= 0 ? 4 : 0;
        ++ctx->grid_id;   // (and which pixels see the mesh)
    }
    ctx->grid_entries = entries;
    ctx->grid_built = true;
    return PT_OK;
}

void free_mesh(PtContext* ctx) {
    free_bvh(ctx);
    free_grid(ctx);
    free_grid_arrays(ctx->grid_spare);
    ctx->grid_want_fine = false;
    ctx->grid_wanted = false;
    ctx->grid_res_reduced = false;
    float** f[] = {&ctx->d_vertices, &ctx->d_tris, &ctx->d_vnormals, &ctx->d_gnormals, &ctx->d_cull, &ctx->d_tris_abs, &ctx->d_tris_live, &ctx->d_tris_grid};
    if (ctx->d_live_count) (void)hipFree(ctx->d_live_count);
    ctx->d_live_count = nullptr;
    if (ctx->d_compact_tmp) (void)hipFree(ctx->d_compact_tmp);
    ctx->d_compact_tmp = nullptr;
    for (float** q : f) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    if (ctx->d_indices) (void)hipFree(ctx->d_indices);
    ctx->d_indices = nullptr;
    if (ctx->d_material_ids) (void)hipFree(ctx->d_material_ids);
    ctx->d_material_ids = nullptr;
    ctx->num_material_ids = 0;
    ctx->max_material_id = -1;
    ctx->num_vertices = ctx->num_triangles = 0;
    ctx->records_valid = false;
    ctx->static_records_valid = false;
    ctx->cull_valid = false;
}

// Makes sure the per-triangle records match the current mesh and camera position.
PtStatus ensure_records(PtContext* ctx) {
    const float* ro = ctx->uniforms.camera_position;
    // the cull records (a mode no default path reads) are only written while that mode is selected
    const bool want_cull = ctx->accel_mode == PT_ACCEL_CULL;
    if (ctx->records_valid && memcmp(ro, ctx->records_ro, sizeof ctx->records_ro) == 0 && (!want_cull || ctx->cull_valid)) return PT_OK;
    // a camera move rewrites what depends on the ray origin; normals and origin-independent records only after an upload
    PT_HIP_TRY(pt_launch_prepare_triangles(ctx->d_vertices, ctx->d_indices, ctx->num_triangles, ro, ctx->d_tris,
                                           ctx->d_vnormals, ctx->d_gnormals, want_cull ? ctx->d_cull : nullptr, ctx->d_tris_abs, ctx->d_tris_grid,
                                           ctx->static_records_valid ? 0 : 1, ctx->stream));
    ctx->static_records_valid = true;
    ctx->cull_valid = want_cull;
    // what is derived from the records is brought up to date when a dispatch actually uses it (ensure_* below)
    ctx->live_valid = false;
    ctx->bvh_emit_valid = false;
    ctx->grid_valid = false;
    memcpy(ctx->records_ro, ro, sizeof ctx->records_ro);
    ctx->records_valid = true;
    ctx->primary_cache_valid = false;
    ctx->list_valid = false;  // new mesh or camera position: the tile costs will change
    ctx->costs_stale = true;
    ctx->dispatches_here = 0;
    ctx->px_samples_here = 0.0;
    // a new view starts with the grid that is quick to build -- unless the dispatch that brought the move is itself long enough
    // to repay the fine one (pt_dispatch has looked: grid_fine_over_move)
    ctx->grid_want_fine = ctx->grid_fine_over_move;
    return PT_OK;
}

// compacted record array of the all-triangles loop
PtStatus ensure_live(PtContext* ctx) {
    if (ctx->live_valid) return PT_OK;
    PT_HIP_TRY(pt_launch_compact_triangles(ctx->d_tris, ctx->num_triangles, ctx->d_tris_live, ctx->d_live_count, ctx->d_compact_tmp, ctx->stream));
    ctx->live_valid = true;
    return PT_OK;
}

// the BVH's boxes and leaf-order records are expressed relative to the ray origin too
PtStatus ensure_bvh_emit(PtContext* ctx) {
    if (ctx->bvh_emit_valid || !ctx->bvh_built) return PT_OK;
    const float* ro = ctx->records_ro;
    // absolute-box variant: its padding scales with the largest coordinate a ray origin or a box can have
    ctx->bvh.scene_scale = std::max(ctx->mesh_extent, std::max(std::fabs(ro[0]), std::max(std::fabs(ro[1]), std::fabs(ro[2]))));
    PT_HIP_TRY(pt_bvh_emit(ctx->bvh, ro, ctx->d_tris, ctx->d_tris_abs, ctx->stream));
    if (ctx->fault_inject == 2 && ctx->num_triangles >= 2) {
        // testing pt_verify_accel: the root forgets its first child -- the binary forms take the second child twice, the wide
        // form gets an unused slot
        PT_HIP_TRY(hipMemcpyAsync(ctx->bvh.nodes + 12, ctx->bvh.nodes + 13, sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        PT_HIP_TRY(hipMemcpyAsync(ctx->bvh.nodes_abs + 12, ctx->bvh.nodes_abs + 13, sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        PT_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->bvh.nodes_w + 3), (int)PT_BVH_NO_CHILD, 1, ctx->stream));
    }
    ctx->bvh_emit_valid = true;
    return PT_OK;
}

PtStatus ensure_grid(PtContext* ctx) {
    if (ctx->grid_valid || !ctx->grid_wanted) return PT_OK;
    PtStatus s = build_grid(ctx);
    if (s != PT_OK) return s;
    ctx->grid_valid = true;  // (grid_built may still be false: lists too long for this scene)
    return PT_OK;
}

PtStatus fill_params(PtContext* ctx, PtKernelParams* p, const Program** prog_out) {
    PT_REQUIRE(ctx->d_image[0] != nullptr, PT_ERR_STATE, "no accumulation target: call pt_target_create first");
    PT_REQUIRE(ctx->have_uniforms, PT_ERR_STATE, "uniforms not set: call pt_set_uniforms first");
    const Program* prog = find_program(ctx, ctx->current_program);
    PT_REQUIRE(prog != nullptr, PT_ERR_STATE, "no program in use: call pt_use_program first");
    const PtUniforms& u = ctx->uniforms;
    memset(p, 0, sizeof *p);
    memcpy(p->ro, u.camera_position, 12);
    memcpy(p->llc, u.camera_lower_left_corner, 12);
    memcpy(p->hor, u.camera_horizontal, 12);
    memcpy(p->ver, u.camera_vertical, 12);
    p->width = ctx->width;
    p->height = ctx->height;
    p->win_x0 = ctx->win[0]; p->win_y0 = ctx->win[1]; p->win_w = ctx->win[2]; p->win_h = ctx->win[3];
    p->row_block = ctx->row_block > 0 ? ctx->row_block : 1; p->row_index = ctx->row_index; p->row_count = ctx->row_count;
    p->depth = u.depth;
    p->accum_mode = (int32_t)ctx->accum_mode;
    p->normal_mode = (int32_t)ctx->normal_mode;
    p->counters = ctx->d_counters;
    p->primary_reuse = ctx->primary_reuse ? 1 : 0;
    if (prog->kind == PT_PROGRAM_CPU_TRACER) {
        PT_REQUIRE(u.num_triangles >= 0 && u.num_triangles <= ctx->num_triangles, PT_ERR_INVALID_ARGUMENT,
                   "u_NumTriangles = %d but %d triangles are uploaded", u.num_triangles, ctx->num_triangles);
        PT_REQUIRE(ctx->have_spheres, PT_ERR_STATE, "no spheres uploaded: call pt_upload_spheres first");
        PT_REQUIRE(ctx->d_materials != nullptr && ctx->num_materials >= 10, PT_ERR_STATE,
                   "the CPU tracer's scene has one material per sphere: upload 10 materials (r, g, b, type) first");
        p->num_triangles = u.num_triangles;
        if (u.num_triangles > 0) {
            PtStatus s = ensure_records(ctx);
            if (s != PT_OK) return s;
        }
        p->tris_abs = ctx->d_tris_abs;
        p->vnormals = ctx->d_vnormals;
        p->materials = ctx->d_materials;
        p->raw_spheres = ctx->d_spheres;
        memcpy(p->cpu_bounds, ctx->mesh_bounds, sizeof p->cpu_bounds);
        p->cpu_max_t = ctx->cpu_max_t;
        p->cpu_seed = ctx->cpu_seed;
        *prog_out = prog;
        return PT_OK;
    }
    if (prog->kind == PT_PROGRAM_MESH) {
        PT_REQUIRE(u.num_triangles >= 0 && u.num_triangles <= ctx->num_triangles, PT_ERR_INVALID_ARGUMENT,
                   "u_NumTriangles = %d but %d triangles are uploaded", u.num_triangles, ctx->num_triangles);
        p->num_triangles = u.num_triangles;
        if (u.num_triangles > 0) {
            PtStatus s = ensure_records(ctx);
            if (s != PT_OK) return s;
        }
        p->tris = ctx->d_tris;
        // the compacted array covers the whole mesh: usable when every uploaded triangle is traced
        if (u.num_triangles == ctx->num_triangles && u.num_triangles > 0 && ctx->accel_mode != PT_ACCEL_CULL && !ctx->keep_unreachable) {
            p->tris = ctx->d_tris_live;
            p->live_count = ctx->d_live_count;
        }
        p->vnormals = ctx->d_vnormals;
        p->gnormals = ctx->d_gnormals;
        // (a mesh of up to PT_BVH_LEAF_MAX triangles is a single leaf: pt_build_accel keeps no hierarchy and the loop IS its walk)
        PT_REQUIRE(ctx->accel_mode != PT_ACCEL_BVH || ctx->bvh_built || ctx->bvh_trivial, PT_ERR_STATE,
                   "PT_ACCEL_BVH but no BVH: call pt_build_accel first");
        // the hierarchy covers the whole mesh, so it serves only when every uploaded triangle is traced
        // and in AUTO mode only when the mesh is big enough for the walk to beat the plain loop
        const bool use_bvh = ctx->bvh_built && ctx->accel_mode != PT_ACCEL_BRUTE && ctx->accel_mode != PT_ACCEL_CULL &&
                             u.num_triangles == ctx->num_triangles &&
                             (ctx->accel_mode == PT_ACCEL_BVH ||
                              u.num_triangles > (is_materials(ctx->shading_mode) ? PT_BVH_AUTO_MIN_TRIANGLES_MOVING : PT_BVH_AUTO_MIN_TRIANGLES));
        PT_REQUIRE(use_bvh || ctx->accel_mode != PT_ACCEL_BVH || (ctx->bvh_trivial && u.num_triangles == ctx->num_triangles), PT_ERR_STATE,
                   "the BVH covers all %d triangles but u_NumTriangles = %d",
                   ctx->num_triangles, u.num_triangles);
        p->shading_mode = (int32_t)ctx->shading_mode;
        // (the grid's query forms the byte offset of a 64-byte triangle record in 32 bits: meshes of at most 2^26 triangles)
        PT_REQUIRE(ctx->accel_mode != PT_ACCEL_GRID || u.num_triangles <= (1 << 26), PT_ERR_INVALID_ARGUMENT,
                   "PT_ACCEL_GRID serves meshes of at most %d triangles (this one: %d)", 1 << 26, u.num_triangles);
        const bool grid_possible = ctx->grid_wanted && ctx->shading_mode == PT_SHADE_REFERENCE && u.num_triangles == ctx->num_triangles &&
                                   u.num_triangles > 0 && u.num_triangles <= (1 << 26) &&
                                   (ctx->accel_mode == PT_ACCEL_GRID || (ctx->accel_mode == PT_ACCEL_AUTO && ctx->auto_grid));
        if (grid_possible) {
            PtStatus gs = ensure_grid(ctx);
            if (gs != PT_OK) return gs;
        }
        PT_REQUIRE(ctx->accel_mode != PT_ACCEL_GRID || (ctx->grid_built && ctx->shading_mode == PT_SHADE_REFERENCE && u.num_triangles == ctx->num_triangles),
                   PT_ERR_STATE, "PT_ACCEL_GRID needs pt_build_accel, PT_SHADE_REFERENCE and u_NumTriangles = all %d triangles", ctx->num_triangles);
        const bool use_grid = grid_possible && ctx->grid_built;  // measured faster than loop and hierarchy at every size
        if (use_bvh && !use_grid) {
            PtStatus bs = ensure_bvh_emit(ctx);
            if (bs != PT_OK) return bs;
        }
        // The materials mode walks the hierarchy (its bounce rays start anywhere), but the camera rays of a view all start at the
        // camera: with the search left to the library their hits come from the direction grid. Measured per dispatch of the
        // 4K frame of 1M triangles: 0.37 ms less (1 sample: 1.74 -> 1.37 ms; 16: 10.4 -> 10.2), 968 triangles at 1080p,
        // 16 samples: 1.44 -> 1.27 ms. Building the grid of the million triangles takes 0.9 ms per camera position, so it
        // starts with the second dispatch from a position: a camera that moves every frame never pays for it. A call that
        // may not allocate uses the grid only if an earlier dispatch of this camera position built it.
        bool grid_for_camera_rays = false;
        if (is_materials(ctx->shading_mode) && use_bvh && ctx->lane_walk == 1 && ctx->primary_reuse && ctx->grid_wanted && ctx->auto_grid &&
            ctx->accel_mode == PT_ACCEL_AUTO && ctx->camera_ray_grid && u.depth > 0 && u.num_triangles <= (1 << 26) &&
            (ctx->grid_valid || (ctx->dispatches_here >= 1 && !ctx->no_alloc))) {
            PtStatus gs = ensure_grid(ctx);
            if (gs != PT_OK) return gs;
            grid_for_camera_rays = ctx->grid_built;
        }
        if (use_grid || grid_for_camera_rays) {
            p->grid_cells = ctx->grid.cells;
            p->grid_list = ctx->grid.list;
            p->grid_occ = ctx->grid_use_occ ? ctx->grid.occ : nullptr;
            p->grid_occ_flag = ctx->d_grid_occ_flag;
            p->grid_tris = ctx->d_tris_grid;
            p->grid_res = ctx->grid.res;
        }
        if (use_grid) {
            // (the grid answers every query of the dispatch: nothing else to attach)
        } else if (is_materials(ctx->shading_mode)) {
            PT_REQUIRE(ctx->accel_mode != PT_ACCEL_CULL, PT_ERR_STATE, "PT_ACCEL_CULL serves rays from the camera only; PT_SHADE_MATERIALS moves the origin");
            PT_REQUIRE(ctx->d_materials != nullptr, PT_ERR_STATE, "PT_SHADE_MATERIALS but no material table: call pt_upload_materials first");
            PT_REQUIRE(ctx->num_material_ids == ctx->num_triangles, PT_ERR_STATE,
                       "PT_SHADE_MATERIALS needs one material id per triangle: %d ids for %d triangles (call pt_upload_material_ids after pt_upload_mesh)",
                       ctx->num_material_ids, ctx->num_triangles);
            PT_REQUIRE(ctx->max_material_id < ctx->num_materials, PT_ERR_STATE, "material id %d but the table has %d rows",
                       ctx->max_material_id, ctx->num_materials);
            p->tris_abs = ctx->d_tris_abs;
            p->materials = ctx->d_materials;
            p->material_ids = ctx->d_material_ids;
            if (use_bvh) {
                p->lane_walk = ctx->lane_walk;
                p->lane_stack_limit = ctx->lane_stack_limit;
                p->bvh_nodes_abs = ctx->bvh.nodes_abs;
                p->bvh_tris_abs = ctx->bvh.sorted_tris_abs;
                p->bvh_nodes_w = ctx->bvh.nodes_w;
                p->bvh_tris_q = ctx->bvh.tris_q;
                p->bvh_qframe = ctx->bvh.qframe;
                p->bvh_ids = ctx->bvh.sorted_ids;
            }
        } else if (use_bvh) {
            p->bvh_nodes = ctx->bvh.nodes;
            p->bvh_tris = ctx->bvh.sorted_tris;
            p->bvh_ids = ctx->bvh.sorted_ids;
        } else if (ctx->accel_mode == PT_ACCEL_CULL) {
            p->cull = ctx->d_cull;
            p->tris = ctx->d_tris;
            p->live_count = nullptr;
        }
        // Whatever kernel ends up walking the compacted array (the loop of the render or of the primary-hit kernel). The materials
        // mode's render kernels never do (they walk the absolute records), so a camera move there does not pay for the
        // compaction (5.5 ms for a million triangles); pt_readback_primary asks for it itself.
        if (!is_materials(ctx->shading_mode) && p->live_count && !p->grid_cells && !p->bvh_nodes) {
            PtStatus ls = ensure_live(ctx);
            if (ls != PT_OK) return ls;
        }
    } else {
        PT_REQUIRE(ctx->have_spheres, PT_ERR_STATE, "no spheres uploaded: call pt_upload_spheres first");
        if (!ctx->sphere_recs_valid || memcmp(u.camera_position, ctx->sphere_recs_ro, sizeof ctx->sphere_recs_ro) != 0) {
            PT_HIP_TRY(pt_launch_prepare_spheres(ctx->d_spheres, u.camera_position, ctx->d_sphere_recs, ctx->stream));
            memcpy(ctx->sphere_recs_ro, u.camera_position, sizeof ctx->sphere_recs_ro);
            ctx->sphere_recs_valid = true;
            ctx->primary_cache_valid = false;
            ctx->list_valid = false;
            ctx->costs_stale = true;
        }
        p->spheres = ctx->d_sphere_recs;
    }
    *prog_out = prog;
    return PT_OK;
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int32_t pt_abi_version(void) { return PT_ABI_VERSION; }

const char* pt_last_error(void) { return g_error; }

PtStatus pt_device_count(int32_t* count) {
    PT_REQUIRE(count != nullptr, PT_ERR_INVALID_ARGUMENT, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(PT_ERR_NO_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *count = n;
    return PT_OK;
}

PtStatus pt_create(int32_t device, PtContext** out_ctx) {
    PT_REQUIRE(out_ctx != nullptr, PT_ERR_INVALID_ARGUMENT, "out_ctx is null");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(PT_ERR_NO_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    PT_REQUIRE(device >= 0 && device < n, PT_ERR_NO_DEVICE, "device %d out of range [0,%d)", device, n);
    PT_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    PT_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PT_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    if (device < kMaxDevices) {
        // Device address of this GPU's copy of the scratch array of pt_dispatch_static: looked up once per device, with
        // that device current. From here on host code uses this pointer, never the array's name.
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device]) {
            void* ptr = nullptr;
            hipError_t se = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
            if (se != hipSuccess || ptr == nullptr)
                return fail(PT_ERR_HIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(se));
            g_scratch_ptr[device] = ptr;
        }
    }
    PtContext* ctx = new (std::nothrow) PtContext();
    PT_REQUIRE(ctx != nullptr, PT_ERR_HIP, "out of host memory");
    ctx->device = device;
    hipError_t err = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipMalloc(&ctx->d_counters, kCounterBytes);
    if (err == hipSuccess) err = hipMemsetAsync(ctx->d_counters, 0, kCounterBytes, ctx->own_stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->own_stream);
    if (err == hipSuccess) err = hipMalloc(&ctx->d_grid_occ_flag, 256);
    if (err == hipSuccess) err = hipMalloc(&ctx->d_spheres, 40 * sizeof(float));
    if (err == hipSuccess) err = hipMalloc(&ctx->d_sphere_recs, 80 * sizeof(float));
    if (err == hipSuccess) err = hipEventCreate(&ctx->ev_start);
    if (err == hipSuccess) err = hipEventCreate(&ctx->ev_stop);
    if (err == hipSuccess) err = hipEventCreate(&ctx->ev_trial[0]);
    if (err == hipSuccess) err = hipEventCreate(&ctx->ev_trial[1]);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&ctx->ev_static, hipEventDisableTiming);
    if (err != hipSuccess) {
        pt_destroy(ctx);
        return fail(PT_ERR_HIP, "context set-up failed: %s", hipGetErrorString(err));
    }
    ctx->stream = ctx->own_stream;
    // PT_TILE_SCHEDULE=0 renders tiles in raster order, =tile orders whole tiles by cost, default: cost-ordered work
    // list in which expensive tiles are split by samples (for measuring what the schedules are worth)
    const char* sched = getenv("PT_TILE_SCHEDULE");
    if (sched && sched[0] == '0') ctx->sched_mode = 0;
    else if (sched && sched[0] == 't') ctx->sched_mode = 1;
    ctx->wave_slots = prop.multiProcessorCount * 4 * 8;

    if (const char* g = getenv("PT_SCHED_SORT")) ctx->sched_sort = atoi(g);
    if (const char* g = getenv("PT_XCD_GROUPS")) ctx->xcd_groups = atoi(g) != 0;
    if (const char* g = getenv("PT_AUTO_GRID")) ctx->auto_grid = g[0] != '0';
    if (const char* g = getenv("PT_BVH_BUILDER")) ctx->bvh_builder = g[0] == 'm' ? PT_BVH_BUILDER_MORTON : (g[0] == 's' ? PT_BVH_BUILDER_SAH : (g[0] == 'p' ? PT_BVH_BUILDER_PLOC : PT_BVH_BUILDER_BINNED));
    if (const char* g = getenv("PT_CAMERA_RAY_GRID")) ctx->camera_ray_grid = g[0] != '0';
    if (const char* g = getenv("PT_PRIMARY_CACHE")) ctx->primary_cache_enabled = g[0] != '0';
    if (const char* g = getenv("PT_GRID_OCC")) ctx->grid_occ_override = atoi(g) != 0;
    if (const char* g = getenv("PT_GRID_CARRY")) ctx->grid_carry_override = atoi(g) != 0;
    if (const char* g = getenv("PT_PIXEL_LIST")) ctx->pixel_list_mode = atoi(g) != 0;
    if (const char* g = getenv("PT_GRID_QUEUE")) ctx->grid_queue_override = atoi(g);
    if (const char* g = getenv("PT_GRID_FINE_AFTER")) ctx->grid_fine_after = atof(g);
    if (const char* g = getenv("PT_FAIL_GRID_ALLOC")) ctx->fail_grid_alloc = atoi(g);
    if (const char* g = getenv("PT_FAULT_INJECT")) ctx->fault_inject = g[0] == 'g' ? 1 : (g[0] == 'b' ? 2 : 0);
    if (const char* g = getenv("PT_GRID_RES")) ctx->grid_res_override = std::min(atoi(g), 8192);  // cell indices stay far below 2^32
    if (const char* l = getenv("PT_LANE_WALK")) ctx->lane_walk = l[0] == '0' ? 0 : (l[0] == '2' ? 2 : 1);
    if (const char* l = getenv("PT_LANE_STACK_LIMIT")) ctx->lane_stack_limit = std::max(0, std::min(PT_LANE_STACK, atoi(l)));
    if (const char* k = getenv("PT_KEEP_UNREACHABLE")) ctx->keep_unreachable = k[0] == '1';
    if (const char* f = getenv("PT_SPLIT_FACTOR")) ctx->split_factor = (float)atof(f);  // for experiments
    *out_ctx = ctx;
    return PT_OK;
}

PtStatus pt_destroy(PtContext* ctx) {
    if (!ctx) return PT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->device < kMaxDevices) {  // (the stream is idle: whatever this context left in the GPU's scratch array is no longer in use)
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (g_scratch_use[ctx->device].owner == ctx) g_scratch_use[ctx->device] = ScratchUse{};
    }
    free_target(ctx);
    free_mesh(ctx);
    if (ctx->ev_static) (void)hipEventDestroy(ctx->ev_static);
    if (ctx->d_verify_out) (void)hipFree(ctx->d_verify_out);
    if (ctx->d_spheres) (void)hipFree(ctx->d_spheres);
    if (ctx->d_sphere_recs) (void)hipFree(ctx->d_sphere_recs);
    if (ctx->d_materials) (void)hipFree(ctx->d_materials);
    if (ctx->d_grid_count_scratch) (void)hipFree(ctx->d_grid_count_scratch);
    if (ctx->h_grid_summary) (void)hipHostFree(ctx->h_grid_summary);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_grid_occ_flag) (void)hipFree(ctx->d_grid_occ_flag);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    for (hipEvent_t e : ctx->ev_trial)
        if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return PT_OK;
}

PtStatus pt_set_stream(PtContext* ctx, void* hip_stream) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    s = collect_timing(ctx);
    if (s != PT_OK) return s;
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return PT_OK;
}

PtStatus pt_program_create(PtContext* ctx, PtProgramKind kind, uint32_t* out_program) {
    PT_REQUIRE(ctx != nullptr && out_program != nullptr, PT_ERR_INVALID_ARGUMENT, "null argument");
    *out_program = 0;
    PT_REQUIRE(kind == PT_PROGRAM_WORLD || kind == PT_PROGRAM_MESH || kind == PT_PROGRAM_CPU_TRACER, PT_ERR_INVALID_ARGUMENT, "unknown program kind %d", (int)kind);
    Program p;
    p.handle = ctx->next_program;
    p.kind = kind;
    try {
        ctx->programs.push_back(p);
    } catch (...) {
        return fail(PT_ERR_HIP, "out of host memory");
    }
    ++ctx->next_program;
    *out_program = p.handle;
    return PT_OK;
}

PtStatus pt_program_destroy(PtContext* ctx, uint32_t program) {
    PT_REQUIRE(ctx != nullptr, PT_ERR_INVALID_ARGUMENT, "null context");
    for (size_t i = 0; i < ctx->programs.size(); ++i) {
        if (ctx->programs[i].handle == program) {
            ctx->programs.erase(ctx->programs.begin() + (long)i);
            if (ctx->current_program == program) ctx->current_program = 0;
            return PT_OK;
        }
    }
    return fail(PT_ERR_INVALID_ARGUMENT, "unknown program handle %u", program);
}

PtStatus pt_use_program(PtContext* ctx, uint32_t program) {
    PT_REQUIRE(ctx != nullptr, PT_ERR_INVALID_ARGUMENT, "null context");
    PT_REQUIRE(find_program(ctx, program) != nullptr, PT_ERR_INVALID_ARGUMENT, "unknown program handle %u", program);
    ctx->current_program = program;
    return PT_OK;
}

PtStatus pt_upload_spheres(PtContext* ctx, const float* xyzr, int32_t count) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(xyzr != nullptr, PT_ERR_INVALID_ARGUMENT, "xyzr is null");
    PT_REQUIRE(count == 10, PT_ERR_INVALID_ARGUMENT, "the program loops over exactly 10 spheres; got %d", count);
    PT_HIP_TRY(hipMemcpyAsync(ctx->d_spheres, xyzr, 40 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->have_spheres = true;
    ctx->sphere_recs_valid = false;
    return PT_OK;
}

PtStatus pt_upload_materials(PtContext* ctx, const float* rgbt, int32_t count) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(rgbt != nullptr && count > 0, PT_ERR_INVALID_ARGUMENT, "rgbt is null or count <= 0");
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->d_materials) (void)hipFree(ctx->d_materials);
    ctx->d_materials = nullptr;
    ctx->num_materials = 0;
    PT_HIP_TRY(hipMalloc(&ctx->d_materials, (size_t)count * 4 * sizeof(float)));
    PT_HIP_TRY(hipMemcpyAsync(ctx->d_materials, rgbt, (size_t)count * 4 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->num_materials = count;
    return PT_OK;
}

PtStatus pt_upload_mesh(PtContext* ctx, const float* vertices, int32_t num_vertices, const uint32_t* indices,
                        int32_t num_triangles) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(vertices != nullptr && indices != nullptr, PT_ERR_INVALID_ARGUMENT, "null mesh arrays");
    PT_REQUIRE(num_vertices > 0 && num_triangles > 0, PT_ERR_INVALID_ARGUMENT, "empty mesh (%d vertices, %d triangles)",
               num_vertices, num_triangles);
    // A stray index would make the gather kernel read outside the vertex buffer: check on the host first.
    for (size_t i = 0; i < (size_t)num_triangles * 3; ++i)
        PT_REQUIRE(indices[i] < (uint32_t)num_vertices, PT_ERR_INVALID_ARGUMENT,
                   "index %u (corner %zu) out of range for %d vertices", indices[i], i, num_vertices);
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_mesh(ctx);
    const size_t nt = (size_t)num_triangles;
    // (one error variable and one clean-up: a failure half-way leaves no mesh, not a half-allocated one)
    hipError_t e = hipSuccess;
    auto alloc = [&e](auto** ptr, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(ptr), bytes);
    };
    alloc(&ctx->d_vertices, (size_t)num_vertices * 6 * sizeof(float));
    alloc(&ctx->d_indices, nt * 3 * sizeof(uint32_t));
    alloc(&ctx->d_tris, (nt + PT_TRI_PAD) * PT_TRI_FLOATS * sizeof(float));
    alloc(&ctx->d_tris_live, (nt + PT_TRI_PAD) * PT_TRI_FLOATS * sizeof(float));
    alloc(&ctx->d_live_count, sizeof(uint32_t));
    alloc(&ctx->d_compact_tmp, pt_compact_temp_words(num_triangles) * sizeof(uint32_t));
    alloc(&ctx->d_vnormals, nt * 9 * sizeof(float));
    alloc(&ctx->d_gnormals, nt * 3 * sizeof(float));
    alloc(&ctx->d_cull, nt * PT_CULL_FLOATS * sizeof(float));
    alloc(&ctx->d_tris_abs, (nt + PT_TRI_PAD) * PT_TRI_FLOATS * sizeof(float));
    alloc(&ctx->d_tris_grid, nt * PT_TRI_FLOATS * sizeof(float));
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_tris + nt * PT_TRI_FLOATS, 0, PT_TRI_PAD * PT_TRI_FLOATS * sizeof(float), ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_tris_abs + nt * PT_TRI_FLOATS, 0, PT_TRI_PAD * PT_TRI_FLOATS * sizeof(float), ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_vertices, vertices, (size_t)num_vertices * 6 * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_indices, indices, nt * 3 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ctx->stream);
        free_mesh(ctx);
        return fail(PT_ERR_HIP, "mesh upload failed (%d vertices, %d triangles): %s", num_vertices, num_triangles, hipGetErrorString(e));
    }
    ctx->num_vertices = num_vertices;
    ctx->num_triangles = num_triangles;
    ctx->records_valid = false;
    float extent = 0.f;
    for (size_t v = 0; v < (size_t)num_vertices; ++v)
        for (int k = 0; k < 3; ++k) {
            const float c = std::fabs(vertices[6 * v + k]);
            if (c > extent && c < 3.0e38f) extent = c;
        }
    ctx->mesh_extent = extent;
    for (int k = 0; k < 3; ++k) { ctx->mesh_bounds[k] = 3.402823466e+38f; ctx->mesh_bounds[3 + k] = -3.402823466e+38f; }
    for (size_t v = 0; v < (size_t)num_vertices; ++v)
        for (int k = 0; k < 3; ++k) {
            const float c = vertices[6 * v + k];
            if (c < ctx->mesh_bounds[k]) ctx->mesh_bounds[k] = c;
            if (c > ctx->mesh_bounds[3 + k]) ctx->mesh_bounds[3 + k] = c;
        }
    return PT_OK;
}

PtStatus pt_upload_material_ids(PtContext* ctx, const int32_t* ids, int32_t count) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(ids != nullptr, PT_ERR_INVALID_ARGUMENT, "ids is null");
    PT_REQUIRE(ctx->num_triangles > 0, PT_ERR_STATE, "no mesh uploaded: material ids belong to the triangles of pt_upload_mesh");
    PT_REQUIRE(count == ctx->num_triangles, PT_ERR_INVALID_ARGUMENT, "%d material ids for %d triangles", count, ctx->num_triangles);
    int32_t top = -1;
    for (int32_t i = 0; i < count; ++i) {
        PT_REQUIRE(ids[i] >= 0, PT_ERR_INVALID_ARGUMENT, "material id %d of triangle %d is negative (map 'no material' to a table row)", ids[i], i);
        if (ids[i] > top) top = ids[i];
    }
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->d_material_ids) (void)hipFree(ctx->d_material_ids);
    ctx->d_material_ids = nullptr;
    ctx->num_material_ids = 0;
    PT_HIP_TRY(hipMalloc(&ctx->d_material_ids, (size_t)count * sizeof(int32_t)));
    PT_HIP_TRY(hipMemcpyAsync(ctx->d_material_ids, ids, (size_t)count * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->num_material_ids = count;
    ctx->max_material_id = top;
    return PT_OK;
}

PtStatus pt_set_primary_reuse(PtContext* ctx, int32_t enable) {
    PT_REQUIRE(ctx != nullptr, PT_ERR_INVALID_ARGUMENT, "null context");
    ctx->primary_reuse = enable != 0;
    return PT_OK;
}

PtStatus pt_set_shading_mode(PtContext* ctx, PtShadingMode mode) {
    PT_REQUIRE(ctx != nullptr, PT_ERR_INVALID_ARGUMENT, "null context");
    PT_REQUIRE(mode == PT_SHADE_REFERENCE || is_materials(mode), PT_ERR_INVALID_ARGUMENT, "unknown shading mode %d", (int)mode);
    ctx->shading_mode = mode;
    return PT_OK;
}

PtStatus pt_build_accel(PtContext* ctx) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(ctx->num_triangles > 0, PT_ERR_STATE, "no mesh uploaded");
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_bvh(ctx);
    ctx->grid_wanted = true;
    ctx->records_valid = false;
    const int32_t n = ctx->num_triangles;
    if (n <= PT_BVH_LEAF_MAX) {  // a single leaf: the plain loop is the traversal
        ctx->bvh_trivial = true;
        return PT_OK;
    }
    PtBvhBuild& b = ctx->bvh;
    b.num_triangles = n;
    b.vertices = ctx->d_vertices;
    b.indices = ctx->d_indices;
    const size_t nn = (size_t)n, ni = (size_t)(n - 1);
    b.sort_tmp_bytes = pt_bvh_sort_temp_bytes(n);
    hipError_t e = hipSuccess;
    auto alloc = [&e](auto** ptr, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(ptr), bytes ? bytes : 16);
    };
    alloc(&b.tri_box, nn * 6 * sizeof(float));
    alloc(&b.scene_box, 6 * sizeof(uint32_t));
    alloc(&b.keys_tmp, nn * sizeof(unsigned long long));
    alloc(&b.keys, nn * sizeof(unsigned long long));
    alloc(&b.order_tmp, nn * sizeof(uint32_t));
    alloc(&b.order, nn * sizeof(uint32_t));
    alloc(&b.sort_tmp, b.sort_tmp_bytes);
    alloc(&b.child, ni * 2 * sizeof(int));
    alloc(&b.range, ni * 2 * sizeof(int));
    alloc(&b.node_box, ni * 6 * sizeof(float));
    alloc(&b.done, ni * sizeof(int));
    alloc(&b.remaining, sizeof(int));
    alloc(&b.nodes, ni * PT_BVH_NODE_FLOATS * sizeof(float));
    alloc(&b.sorted_tris, (nn + PT_TRI_PAD) * PT_TRI_FLOATS * sizeof(float));
    alloc(&b.sorted_ids, (nn + PT_TRI_PAD) * sizeof(int));
    alloc(&b.nodes_abs, ni * PT_BVH_NODE_FLOATS * sizeof(float));
    alloc(&b.sorted_tris_abs, (nn + PT_TRI_PAD) * PT_TRI_FLOATS * sizeof(float));
    alloc(&b.nodes_w, ni * PT_BVH_WNODE_WORDS * sizeof(uint32_t));
    alloc(&b.wide_index, ni * sizeof(int));
    alloc(&b.wide_child, ni * 4 * sizeof(int));
    alloc(&b.tris_q, (nn + PT_TRI_PAD) * PT_BVH_QTRI_FLOATS * sizeof(float));
    alloc(&b.qframe, 8 * sizeof(float));
    if (e == hipSuccess)
        e = ctx->bvh_builder == PT_BVH_BUILDER_SAH      ? pt_bvh_build_topology_sah(b, ctx->stream, &ctx->bvh_refit_passes)
            : ctx->bvh_builder == PT_BVH_BUILDER_MORTON ? pt_bvh_build_topology(b, ctx->stream, &ctx->bvh_refit_passes)
            : ctx->bvh_builder == PT_BVH_BUILDER_BINNED ? pt_bvh_build_topology_binned(b, ctx->stream, &ctx->bvh_refit_passes)
                                                        : pt_bvh_build_topology_ploc(b, ctx->stream, &ctx->bvh_refit_passes);
    if (e != hipSuccess) {
        free_bvh(ctx);
        return fail(PT_ERR_HIP, "BVH build failed: %s", hipGetErrorString(e));
    }
    if (getenv("PT_DEBUG_SCHED")) fprintf(stderr, "[pt bvh] %d triangles, hierarchy finished after %d bottom-up sweeps\n", n, ctx->bvh_refit_passes);
    ctx->bvh_built = true;
    ctx->records_valid = false;  // leaf-order records and origin-relative boxes are emitted with the next records
    return PT_OK;
}

PtStatus pt_set_bvh_builder(PtContext* ctx, PtBvhBuilder builder) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(builder == PT_BVH_BUILDER_SAH || builder == PT_BVH_BUILDER_MORTON || builder == PT_BVH_BUILDER_PLOC || builder == PT_BVH_BUILDER_BINNED,
               PT_ERR_INVALID_ARGUMENT,
               "unknown BVH builder %d", (int)builder);
    ctx->bvh_builder = builder;
    return PT_OK;
}

PtStatus pt_set_accel_mode(PtContext* ctx, PtAccelMode mode) {
    PT_REQUIRE(ctx != nullptr, PT_ERR_INVALID_ARGUMENT, "null context");
    PT_REQUIRE(mode == PT_ACCEL_AUTO || mode == PT_ACCEL_BRUTE || mode == PT_ACCEL_BVH || mode == PT_ACCEL_CULL || mode == PT_ACCEL_GRID, PT_ERR_INVALID_ARGUMENT, "unknown accel mode %d", (int)mode);
    ctx->accel_mode = mode;
    return PT_OK;
}

PtStatus pt_set_normal_mode(PtContext* ctx, PtNormalMode mode) {
    PT_REQUIRE(ctx != nullptr, PT_ERR_INVALID_ARGUMENT, "null context");
    PT_REQUIRE(mode == PT_NORMALS_VERTEX || mode == PT_NORMALS_GEOMETRIC, PT_ERR_INVALID_ARGUMENT, "unknown normal mode %d", (int)mode);
    ctx->normal_mode = mode;
    return PT_OK;
}

static PtStatus target_common(PtContext* ctx, int32_t width, int32_t height) {
    ctx->width = width;
    ctx->height = height;
    ctx->current = 0;
    ctx->win[0] = 0; ctx->win[1] = 0; ctx->win[2] = width; ctx->win[3] = height;
    const size_t n = (size_t)width * height;
    PT_HIP_TRY(hipMalloc(&ctx->d_ids, n * sizeof(int32_t)));
    PT_HIP_TRY(hipMalloc(&ctx->d_t, n * sizeof(float)));
    const size_t tiles = (size_t)PT_TILES_ACROSS(width) * PT_TILES_ACROSS(height);
    PT_HIP_TRY(hipMalloc(&ctx->d_tile_cost, tiles * sizeof(uint32_t)));
    // (a list dealt to the XCDs leaves gaps where a band has fewer items than the longest one: eight times the items then)
    PT_HIP_TRY(hipMalloc(&ctx->d_items, (ctx->xcd_groups != 0 ? kXcdGroups : 1) * (tiles + (size_t)PT_MAX_SPLIT_FACTOR * ctx->wave_slots) * sizeof(PtWorkItem)));
    PT_HIP_TRY(hipMalloc(&ctx->d_sched_counts, 2 * sizeof(uint32_t)));
    PT_HIP_TRY(hipMalloc(&ctx->d_sched_tmp, PT_SCHED_TMP_WORDS * sizeof(uint32_t)));
    PT_HIP_TRY(hipMalloc(&ctx->d_split_tiles, (size_t)PT_MAX_SPLIT_FACTOR * ctx->wave_slots * sizeof(uint32_t)));
    PT_HIP_TRY(hipMalloc(&ctx->d_primary_cache, n * sizeof(float4)));
    ctx->primary_cache_valid = false;
    if (PT_TILE_WAVES == 1 && width <= 65535 && height <= 65535) {   // (a list entry is y << 16 | x)
        PT_HIP_TRY(hipMalloc(&ctx->d_pixel_list, (n + 63) / 64 * 64 * sizeof(uint32_t)));
        PT_HIP_TRY(hipMalloc(&ctx->d_pixel_mask, tiles * sizeof(unsigned long long)));
        PT_HIP_TRY(hipMalloc(&ctx->d_pixel_counts, tiles * sizeof(uint32_t)));
        PT_HIP_TRY(hipMalloc(&ctx->d_pixel_offsets, tiles * sizeof(uint32_t)));
        PT_HIP_TRY(hipMalloc(&ctx->d_pixel_total, 2 * sizeof(uint32_t)));
    }
    ctx->pixel_list_valid = false;
    ctx->sched_have_cost = false;
    return PT_OK;
}

PtStatus pt_target_create(PtContext* ctx, int32_t width, int32_t height) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(width > 0 && height > 0 && width <= 65536 && height <= 65536, PT_ERR_INVALID_ARGUMENT, "bad target size %dx%d", width, height);
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_target(ctx);
    const size_t bytes = (size_t)width * height * 4 * sizeof(float);
    ctx->image_owned = true;
    hipError_t e = hipMalloc(&ctx->d_image[0], bytes);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_image[1], bytes);
    if (e != hipSuccess) {
        free_target(ctx);  // no half-made target: later calls report "no accumulation target"
        return fail(PT_ERR_HIP, "allocating two %dx%d RGBA32F images failed: %s", width, height, hipGetErrorString(e));
    }
    s = target_common(ctx, width, height);
    if (s != PT_OK) {
        free_target(ctx);  // no half-made target: later calls report "no accumulation target"
        return s;
    }
    return pt_target_clear(ctx);
}

PtStatus pt_target_bind_external(PtContext* ctx, void* rgba_a, void* rgba_b, int32_t width, int32_t height) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(rgba_a != nullptr && rgba_b != nullptr, PT_ERR_INVALID_ARGUMENT, "null image pointer");
    PT_REQUIRE(((uintptr_t)rgba_a % 16) == 0 && ((uintptr_t)rgba_b % 16) == 0, PT_ERR_INVALID_ARGUMENT, "images must be 16-byte aligned");
    PT_REQUIRE(width > 0 && height > 0 && width <= 65536 && height <= 65536, PT_ERR_INVALID_ARGUMENT, "bad target size %dx%d", width, height);
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_target(ctx);
    ctx->d_image[0] = (float*)rgba_a;
    ctx->d_image[1] = (float*)rgba_b;
    ctx->image_owned = false;
    s = target_common(ctx, width, height);
    if (s != PT_OK) free_target(ctx);  // (the caller's images are not freed: image_owned is false)
    return s;
}

PtStatus pt_target_clear(PtContext* ctx) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(ctx->d_image[0] != nullptr, PT_ERR_STATE, "no accumulation target");
    const size_t n = (size_t)ctx->width * ctx->height;
    const float a = ctx->accum_mode == PT_ACCUM_MEAN ? 1.0f : 0.0f;
    PT_HIP_TRY(pt_launch_fill_rgba(ctx->d_image[0], n, 0.f, 0.f, 0.f, a, ctx->stream));
    if (ctx->d_image[1] != ctx->d_image[0]) PT_HIP_TRY(pt_launch_fill_rgba(ctx->d_image[1], n, 0.f, 0.f, 0.f, a, ctx->stream));
    ctx->current = 0;
    return PT_OK;
}

PtStatus pt_set_window(PtContext* ctx, int32_t x0, int32_t y0, int32_t w, int32_t h) {
    PT_REQUIRE(ctx != nullptr, PT_ERR_INVALID_ARGUMENT, "null context");
    PT_REQUIRE(ctx->d_image[0] != nullptr, PT_ERR_STATE, "no accumulation target");
    if (w == 0 && h == 0) {
        ctx->win[0] = 0; ctx->win[1] = 0; ctx->win[2] = ctx->width; ctx->win[3] = ctx->height;
        return PT_OK;
    }
    PT_REQUIRE(x0 >= 0 && y0 >= 0 && w > 0 && h > 0 && x0 + w <= ctx->width && y0 + h <= ctx->height, PT_ERR_INVALID_ARGUMENT,
               "window (%d,%d,%d,%d) outside the %dx%d target", x0, y0, w, h, ctx->width, ctx->height);
    ctx->win[0] = x0; ctx->win[1] = y0; ctx->win[2] = w; ctx->win[3] = h;
    return PT_OK;
}

PtStatus pt_set_row_interleave(PtContext* ctx, int32_t block_rows, int32_t index, int32_t count) {
    PT_REQUIRE(ctx != nullptr, PT_ERR_INVALID_ARGUMENT, "null context");
    if (count <= 1) {
        ctx->row_block = 0; ctx->row_index = 0; ctx->row_count = 1;
    } else {
        PT_REQUIRE(block_rows >= 1 && index >= 0 && index < count, PT_ERR_INVALID_ARGUMENT,
                   "row interleave: block_rows %d must be >= 1 and index %d in [0, %d)", block_rows, index, count);
        ctx->row_block = block_rows; ctx->row_index = index; ctx->row_count = count;
    }
    ctx->sched_have_cost = false;  // another set of pixels: the tile costs no longer apply
    return PT_OK;
}

PtStatus pt_set_accum_mode(PtContext* ctx, PtAccumMode mode) {
    PT_REQUIRE(ctx != nullptr, PT_ERR_INVALID_ARGUMENT, "null context");
    PT_REQUIRE(mode == PT_ACCUM_MEAN || mode == PT_ACCUM_SUM, PT_ERR_INVALID_ARGUMENT, "unknown accumulation mode %d", (int)mode);
    ctx->accum_mode = mode;
    return PT_OK;
}

PtStatus pt_set_cpu_tracer(PtContext* ctx, uint32_t seed, float ray_max_t) {
    PT_REQUIRE(ctx != nullptr, PT_ERR_INVALID_ARGUMENT, "null context");
    PT_REQUIRE(ray_max_t > 0.f, PT_ERR_INVALID_ARGUMENT, "ray_max_t must be positive; got %g", (double)ray_max_t);
    ctx->cpu_seed = seed;
    ctx->cpu_max_t = ray_max_t;
    return PT_OK;
}

PtStatus pt_set_uniforms(PtContext* ctx, const PtUniforms* uniforms) {
    PT_REQUIRE(ctx != nullptr && uniforms != nullptr, PT_ERR_INVALID_ARGUMENT, "null argument");
    PT_REQUIRE(uniforms->num_triangles >= 0, PT_ERR_INVALID_ARGUMENT, "negative u_NumTriangles");
    ctx->uniforms = *uniforms;
    ctx->have_uniforms = true;
    return PT_OK;
}

}  // extern "C"
namespace {
// Whether the launches of this pt_dispatch render the gathered hit pixels of a sparse view (ctx->use_pixel_list), building the
// list when the view has none yet. `p` as fill_params left it.
constexpr int kPixelListMinSamples = 8;   // shorter dispatches (frame loops) keep the window's blocks: the sky pass would cost them more than the idle lanes
PtStatus choose_pixel_list(PtContext* ctx, const PtKernelParams& p, const Program* prog, int32_t num_samples) {
    ctx->use_pixel_list = false;
    PtContext::PixelListKey key;
    memset(&key, 0, sizeof key);
    memcpy(key.camera, p.ro, 12); memcpy(key.camera + 3, p.llc, 12); memcpy(key.camera + 6, p.hor, 12); memcpy(key.camera + 9, p.ver, 12);
    memcpy(key.win, ctx->win, sizeof key.win);
    key.rows[0] = p.row_block; key.rows[1] = p.row_index; key.rows[2] = p.row_count;
    key.size[0] = ctx->width; key.size[1] = ctx->height;
    key.num_triangles = p.num_triangles;
    // A view that stays: the previous pt_dispatch rendered the same camera (position AND orientation), window and partition. A
    // camera that turns from frame to frame never gets a list: every new list renumbers the tiles, which drops their costs.
    const bool view_stays = ctx->last_view_valid && memcmp(&key, &ctx->last_view_key, sizeof key) == 0;
    ctx->last_view_key = key;
    ctx->last_view_valid = true;
    if (PT_TILE_WAVES != 1 || ctx->pixel_list_mode == 0 || prog->kind != PT_PROGRAM_MESH || p.grid_cells == nullptr || !ctx->primary_reuse || p.depth <= 0 ||
        ctx->sched_mode == 0 || ctx->d_pixel_list == nullptr || ctx->no_alloc)
        return PT_OK;
    if (ctx->pixel_list_mode != 1 && num_samples < kPixelListMinSamples) return PT_OK;
    key.grid_id = ctx->grid_id;
    if (!ctx->pixel_list_valid || memcmp(&key, &ctx->pixel_list_key, sizeof key) != 0) {
        if (ctx->pixel_list_mode != 1 && !view_stays) return PT_OK;
        ctx->pixel_list_valid = false;
        PT_HIP_TRY(pt_launch_pixel_list(p, ctx->d_pixel_mask, ctx->d_pixel_counts, ctx->d_pixel_offsets, ctx->d_pixel_list, ctx->d_pixel_total, ctx->stream));
        uint32_t counted[2] = {0, 0};
        PT_HIP_TRY(hipMemcpyAsync(counted, ctx->d_pixel_total, sizeof counted, hipMemcpyDeviceToHost, ctx->stream));
        PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
        const uint32_t total = counted[0];
        ctx->pixel_hits = total;
        ctx->pixel_blocks = counted[1];
        ctx->pixel_tiles = (int32_t)((total + 63u) / 64u);
        ctx->pixel_list_key = key;
        ctx->pixel_list_valid = true;
        if (++ctx->pixel_list_id == 0) ctx->pixel_list_id = 1;
        if (getenv("PT_DEBUG_SCHED"))
            fprintf(stderr, "[pt pixels] %u of the window's %lld pixels see the mesh, in %u blocks of 8x8: %d tiles of 64\n", total,
                    (long long)ctx->win[2] * ctx->win[3], ctx->pixel_blocks, ctx->pixel_tiles);
    }
    // worth it when the blocks that hold the hit pixels would leave 1/32 or more of their lanes to sky pixels: the 1M triangles at
    // 4K leave 22 % (-11 % time), 968 triangles at 1080p 3.9 % (-4 %), 3,936 triangles 3.8 % (same time); Cornell, whose silhouette
    // is four straight edges, 1.0 % and the torus 1.8 % (no gain measured: they keep the blocks)
    ctx->use_pixel_list = ctx->pixel_tiles > 0 && (ctx->pixel_list_mode == 1 || 32ull * ctx->pixel_hits <= 31ull * 64ull * ctx->pixel_blocks);
    return PT_OK;
}

// First half of a render launch over a range of samples: kernel parameters, source and destination image.
PtStatus begin_pass(PtContext* ctx, PtKernelParams* pp, const Program** prog_out, int32_t first_sample, int32_t num_samples, int32_t stride) {
    PtKernelParams& p = *pp;
    PtStatus s = fill_params(ctx, &p, prog_out);
    if (s != PT_OK) return s;
    p.first_sample = first_sample;
    p.num_samples = num_samples;
    p.sample_stride = stride;
    // PathTracer.cpp:1384,1398: write fbo[current], read tex[current ^ 1]
    p.dst = ctx->d_image[ctx->current];
    p.src = ctx->d_image[ctx->current ^ 1];
    // A pass only rewrites the window; with two distinct images the pixels outside it must follow the swap.
    const bool partial = ctx->win[2] != ctx->width || ctx->win[3] != ctx->height || ctx->row_count > 1;
    if (partial && p.dst != p.src) {
        PT_HIP_TRY(hipMemcpyAsync(p.dst, p.src, (size_t)ctx->width * ctx->height * 16, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return PT_OK;
}

PtStatus end_pass(PtContext* ctx, PtKernelParams& p, const Program* prog, int32_t num_samples);

// One render launch (plus the fold of split tiles) over a range of samples.
PtStatus dispatch_range(PtContext* ctx, int32_t first_sample, int32_t num_samples, int32_t stride) {
    PtKernelParams p;
    const Program* prog = nullptr;
    PtStatus s = begin_pass(ctx, &p, &prog, first_sample, num_samples, stride);
    if (s != PT_OK) return s;
    // Tile order: the costs measured by the previous dispatch of this same window put the expensive tiles first.
    // It is only a scheduling hint (any order renders the same image), so stale costs after a camera or scene change
    // are harmless.
    if (ctx->use_pixel_list) {   // (decided by pt_dispatch for all its launches)
        p.pixel_list = ctx->d_pixel_list;
        p.pixel_tiles = ctx->pixel_tiles;
    }
    if (ctx->sched_mode != 0 && prog->kind != PT_PROGRAM_CPU_TRACER) {
        const int tiles = p.pixel_list ? p.pixel_tiles : PT_TILES_ACROSS(ctx->win[2]) * PT_TILES_ACROSS(ctx->win[3]);
        const uint32_t numbering = p.pixel_list ? ctx->pixel_list_id : 0u;
        const bool same_window = memcmp(ctx->sched_win, ctx->win, sizeof ctx->win) == 0 && ctx->sched_numbering == numbering;
        if (ctx->sched_have_cost && same_window) {
            // The list stays valid for any later dispatch of the same window and sample count (it only has to cover
            // every tile and sample once), so a frame loop does not pay for the scheduler on every frame. It is rebuilt
            // after scene or camera changes, every 16th dispatch, and as soon as the costs gathered since rest on twice as
            // many samples as the ones it was built from: one sample per pixel is a noisy estimate of a tile's cost, and
            // a list built from it left the truly long tiles unsplit (1M triangles, 16 spp: 11.0 ms against 7.6 ms with
            // costs from 16 samples).
            // How eagerly tiles are split (a tile is split when its cost exceeds total / (factor x wave slots)): the
            // hierarchy walk is latency-bound and gains from every additional wave (968 triangles: 8.7 ms unsplit, 3.1 at
            // factor 1, 2.6 at factor 4). The loop and the direction grid are throughput-bound, every extra block adds a
            // drain phase and scratch traffic, so a view whose work is spread over more tiles than the device holds waves
            // is best left alone (1080p Cornell: 8.2 ms at factor 1, 8.7 at 2, 9.5 at 4; 1M triangles at 4K: 2.0 / 2.1 /
            // 2.6). A view whose work sits in few tiles is different: when the factor-1 list already splits a fifth of
            // the wave slots' worth of tiles, the largest factor is used from then on (968 triangles: 1.19 -> 0.96 ms,
            // 3,936: 2.22 -> 1.94, the sphere scene 0.71 -> 0.65), until the costs are dropped (new view).
            // The per-lane walk of the materials mode sits between the two (1M triangles at 4K, 16 samples: 12.7 ms at factor 1, 14.0
            // at 4; 36 triangles through the hierarchy: 7.3 / 8.1; but 3,936 triangles at 1080p: 6.0 / 5.7, 968: 4.0 / 3.5, and
            // 6.1 / 3.9 at factor 16): factor 1, and 4 for a view found to be concentrated.
            // The per-wave walk of the reference shading is one serial chain of dependent fetches per wave, and what bounds a
            // dispatch is its longest chains: 1080p, 256 samples at factor 4 / 8 / 16 / 32 / 64: 3,936 triangles 21.6 / 17.4 / 12.5 /
            // 10.8 / 9.9 ms, 968: 2.7 / 2.1 / 1.8 / 1.65 / 1.6, 1M triangles at 4K, 64 samples: 15.0 / 14.8 / 10.2 / 7.8 / 7.0, 36
            // triangles: 23.3 at 32, 24.2 at 64. Factor 32 (parked samples: up to 2 GiB at these sizes).
            const float base_factor = ctx->split_factor >= 0.f ? std::min(ctx->split_factor, (float)PT_MAX_SPLIT_FACTOR)
                                                               : (p.bvh_nodes ? kWaveWalkSplitFactor : kSpreadSplitFactor);
            const float eager_factor = p.bvh_nodes_abs ? 4.0f : kEagerSplitFactor;
            const bool splitting = ctx->sched_mode == 2 && num_samples > 1;
            const int entry_bytes = is_materials(ctx->shading_mode) && prog->kind != PT_PROGRAM_WORLD ? 16 : 8;  // what a parked sample takes
            const size_t slot_bytes = (size_t)num_samples * PT_TILE_THREADS * (size_t)entry_bytes;
            const int groups = splitting && prog->kind == PT_PROGRAM_MESH && ctx->xcd_groups != 0 ? (int)kXcdGroups : 1;
            const bool reuse = ctx->list_valid && ctx->list_samples == num_samples && ctx->list_base_factor == base_factor && ctx->list_groups == groups &&
                               ctx->list_entry_bytes == entry_bytes && ctx->list_age < 16 && ctx->cost_samples < 2 * ctx->list_basis;
            if (!reuse) {
                // (direction grid: whether its queries look at the block bits first follows the view's statistics so far)
                if (p.grid_occ && ctx->grid_occ_override < 0)
                    PT_HIP_TRY(pt_launch_grid_occ_flag(ctx->d_counters, (unsigned long long)ctx->samples, ctx->d_grid_occ_flag, ctx->stream));
                // The scheduler's counts are read back (a short stall, at rebuilds only): they size the launches and the
                // scratch exactly, and they tell whether this view wants the eager split.
                float factor = (ctx->concentrated && ctx->split_factor < 0.f) ? eager_factor : base_factor;
                uint32_t c[2] = {0, 0};
                for (;;) {
                    const int max_split = splitting ? (int)(factor * (float)ctx->wave_slots) : 0;
                    PT_HIP_TRY(pt_launch_schedule_items(ctx->d_tile_cost, ctx->d_items, ctx->d_sched_counts, ctx->d_split_tiles, ctx->d_sched_tmp, tiles,
                                                        num_samples, max_split, ctx->sched_sort, groups, ctx->stream));
                    if (!splitting) {  // single-sample dispatches (frame loops): whole tiles in cost order, nothing to read back
                        c[0] = (uint32_t)tiles; c[1] = 0;
                        break;
                    }
                    PT_HIP_TRY(hipMemcpyAsync(c, ctx->d_sched_counts, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
                    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
                    if (getenv("PT_DEBUG_SCHED"))
                        fprintf(stderr, "[pt sched] %d tiles, %d samples, costs from %lld samples, factor %g, %d group(s): %u list positions, %u split tiles\n", tiles,
                                num_samples, (long long)ctx->cost_samples, (double)factor, groups, c[0], c[1]);
                    if (ctx->split_factor < 0.f && factor == kSpreadSplitFactor && c[1] >= (uint32_t)(ctx->wave_slots / 5)) {
                        factor = eager_factor;  // work concentrated in few tiles
                        ctx->concentrated = true;
                        continue;
                    }
                    if ((size_t)c[1] * slot_bytes > ((size_t)8 << 30) && factor > 0.f) {  // scratch capped at 8 GiB: split less
                        factor = factor > 0.26f ? factor * 0.5f : 0.f;
                        continue;
                    }
                    break;
                }
                PT_HIP_TRY(hipMemsetAsync(ctx->d_tile_cost, 0, (size_t)tiles * sizeof(uint32_t), ctx->stream));
                ctx->costs_stale = false;  // from here on the costs are those of the current mesh and camera position
                // split tiles park their samples in scratch memory: 8 (materials mode 16) bytes per pixel and sample
                const size_t need = (size_t)c[1] * slot_bytes;
                if (need > ctx->sample_scratch_bytes) {
                    ctx->list_valid = false;  // (stays so if the allocation fails: no list may outlive its scratch)
                    if (ctx->d_sample_scratch) (void)hipFree(ctx->d_sample_scratch);
                    ctx->d_sample_scratch = nullptr;
                    ctx->sample_scratch_bytes = 0;
                    PT_HIP_TRY(hipMalloc(&ctx->d_sample_scratch, need + need / 8));
                    ctx->sample_scratch_bytes = need + need / 8;
                }
                ctx->list_valid = true;
                ctx->list_samples = num_samples;
                ctx->list_base_factor = base_factor;
                ctx->list_entry_bytes = entry_bytes;
                ctx->list_groups = groups;
                ctx->list_items = (int)c[0];
                ctx->list_split = (int)c[1];
                ctx->list_age = 0;
                ctx->list_basis = ctx->cost_samples > 0 ? ctx->cost_samples : 1;
                ctx->cost_samples = 0;
            }
            ++ctx->list_age;
            p.items = ctx->d_items;
            p.sched_counts = ctx->d_sched_counts;
            p.split_tiles = ctx->d_split_tiles;
            p.sample_scratch = ctx->d_sample_scratch;
            p.max_items = ctx->list_items;
            p.max_split_tiles = ctx->list_split;
        } else {
            PT_HIP_TRY(hipMemsetAsync(ctx->d_tile_cost, 0, (size_t)tiles * sizeof(uint32_t), ctx->stream));
            p.items = nullptr;
            ctx->list_valid = false;
            ctx->cost_samples = 0;
            ctx->costs_stale = false;
            ctx->concentrated = false;
        }
        p.tile_cost = ctx->d_tile_cost;
        ctx->cost_samples += num_samples;
        memcpy(ctx->sched_win, ctx->win, sizeof ctx->win);
        ctx->sched_numbering = numbering;
        ctx->sched_have_cost = true;
    }
    return end_pass(ctx, p, prog, num_samples);
}

// Second half: the camera-ray hits kept between dispatches, the launch, the image swap, the sample count.
PtStatus end_pass(PtContext* ctx, PtKernelParams& p, const Program* prog, int32_t num_samples) {
    // Camera-ray hits of this view: read if an earlier dispatch left them, written otherwise (reference shading only; the
    // materials mode also needs the distance).
    const bool cache_usable = ctx->primary_cache_enabled && ctx->primary_reuse && p.depth > 0 && prog->kind != PT_PROGRAM_CPU_TRACER &&
                              (prog->kind == PT_PROGRAM_WORLD || ctx->shading_mode == PT_SHADE_REFERENCE);
    if (cache_usable) {
        PtContext::PrimaryCacheKey key;
        memset(&key, 0, sizeof key);
        memcpy(key.camera, p.ro, 12); memcpy(key.camera + 3, p.llc, 12); memcpy(key.camera + 6, p.hor, 12); memcpy(key.camera + 9, p.ver, 12);
        key.num_triangles = p.num_triangles;
        memcpy(key.win, ctx->win, sizeof key.win);
        key.rows[0] = p.row_block; key.rows[1] = p.row_index; key.rows[2] = p.row_count;
        key.normal_mode = p.normal_mode;
        key.accel_mode = (int32_t)ctx->accel_mode;
        key.program = prog->handle;
        p.primary_cache = ctx->d_primary_cache;
        p.primary_cache_valid = ctx->primary_cache_valid && memcmp(&key, &ctx->primary_cache_key, sizeof key) == 0;
        ctx->primary_cache_key = key;
        ctx->primary_cache_valid = true;  // (this dispatch writes every pixel of the window it does not read)
    }
    // Which form of the direction-grid kernel's loop: a view's first long dispatches that run from a work list time one form each.
    bool timed = false;
    if (p.grid_cells && p.shading_mode == 0 && prog->kind == PT_PROGRAM_MESH) {
        // (a list built from a probe's few samples balances worse than the later ones: such a dispatch is not compared with those)
        if (ctx->carry_trial < 4 && p.items != nullptr && num_samples >= 16 && ctx->list_basis >= 16 && !ctx->no_alloc) {
            if (ctx->carry_trial == 1 || ctx->carry_trial == 3) {   // the form timed by the previous such dispatch
                float ms = 0.f;
                PT_HIP_TRY(hipEventSynchronize(ctx->ev_trial[1]));
                PT_HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_trial[0], ctx->ev_trial[1]));
                ctx->carry_trial_ms[ctx->carry_trial == 1 ? 0 : 1] = (double)ms / (double)ctx->carry_trial_samples;
            }
            if (ctx->carry_trial == 3) {
                ctx->grid_carry = ctx->carry_trial_ms[1] < 0.98 * ctx->carry_trial_ms[0];
                ctx->carry_trial = 4;
                if (getenv("PT_DEBUG_SCHED"))
                    fprintf(stderr, "[pt grid] kernel ms per sample: list loop per wave %.5f, list remainders carried %.5f -> %s\n", ctx->carry_trial_ms[0],
                            ctx->carry_trial_ms[1], ctx->grid_carry ? "carried" : "loop");
            } else {
                ctx->grid_carry = ctx->carry_trial >= 1;   // 0: the first form now; 1, 2: the second
                timed = ctx->carry_trial != 1;             // (its first run is not timed)
                ctx->carry_trial += 1;
                ctx->carry_trial_samples = num_samples;
            }
        }
        p.grid_carry = ctx->grid_carry ? 1 : 0;
        ctx->last_grid_carry = ctx->grid_carry;
    }
    if (timed) PT_HIP_TRY(hipEventRecord(ctx->ev_trial[0], ctx->stream));
    PT_HIP_TRY(pt_launch_render(p, (int)prog->kind, ctx->stream));
    if (timed) PT_HIP_TRY(hipEventRecord(ctx->ev_trial[1], ctx->stream));
    ctx->last_gathered = p.pixel_list != nullptr;
    if (p.pixel_list) {   // the window's other pixels: every sample is the sky colour of their camera ray
        PtKernelParams q = p;
        q.pixel_list = nullptr;
        q.pixel_tiles = 0;
        PT_HIP_TRY(pt_launch_sky_fold(q, ctx->d_pixel_mask, ctx->stream));
    }
    ctx->current ^= 1;  // PathTracer.cpp:1493
    uint64_t rows = (uint64_t)ctx->win[3];
    if (ctx->row_count > 1) {  // rows of the window that belong to this renderer
        rows = 0;
        for (int y = ctx->win[1]; y < ctx->win[1] + ctx->win[3]; ++y) rows += (y / ctx->row_block) % ctx->row_count == ctx->row_index;
    }
    ctx->samples += (uint64_t)ctx->win[2] * rows * (uint64_t)num_samples;
    return PT_OK;
}
}  // namespace
extern "C" {

PtStatus pt_dispatch(PtContext* ctx, int32_t first_sample, int32_t num_samples, int32_t stride) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(first_sample >= 1, PT_ERR_INVALID_ARGUMENT, "first_sample is 1-based; got %d", first_sample);
    PT_REQUIRE(num_samples >= 1 && stride >= 1, PT_ERR_INVALID_ARGUMENT, "num_samples and stride must be >= 1");
    // u_Samples is a float: beyond 2^24 consecutive indices stop being representable.
    PT_REQUIRE((int64_t)first_sample + (int64_t)(num_samples - 1) * stride <= (1 << 24), PT_ERR_INVALID_ARGUMENT,
               "sample index exceeds 2^24 (u_Samples is a float)");
    // Large mesh, direction grid: once this camera position has gathered enough pixel-samples (this dispatch included) the grid
    // is rebuilt at its fine resolution (grid_fine_resolution); both sets of arrays are kept, so only the first switch allocates.
    const double my_px_samples = (double)ctx->win[2] * (double)ctx->win[3] / (double)(ctx->row_count > 1 ? ctx->row_count : 1) * (double)num_samples;
    {
        const bool moved = !ctx->records_valid || memcmp(ctx->uniforms.camera_position, ctx->records_ro, sizeof ctx->records_ro) != 0;
        const bool eligible = ctx->grid_wanted && ctx->have_uniforms && ctx->shading_mode == PT_SHADE_REFERENCE &&
                              grid_fine_resolution(ctx) != grid_resolution(ctx);
        const bool enough = eligible && ctx->grid_fine_after >= 0.0 && (moved ? 0.0 : ctx->px_samples_here) + my_px_samples >= ctx->grid_fine_after;
        if (moved) {
            ctx->grid_fine_over_move = enough;   // (ensure_records resets the view's state and takes this over)
        } else if (enough && !ctx->grid_want_fine) {
            ctx->grid_want_fine = true;
            ctx->grid_valid = false;             // rebuilt at the fine resolution by the update below
        }
    }
    {   // brings everything derived from mesh and camera up to date, which also tells whether the tile costs still apply
        PtKernelParams p;
        const Program* prog = nullptr;
        s = fill_params(ctx, &p, &prog);
        ctx->grid_fine_over_move = false;
        if (s != PT_OK) return s;
        s = choose_pixel_list(ctx, p, prog, num_samples);
        if (s != PT_OK) return s;
    }
    if (ctx->timing) {
        s = collect_timing(ctx);
        if (s != PT_OK) return s;
        PT_HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
    }
    // A long dispatch with no usable tile costs (first one of a scene, window or camera position) would run all its samples
    // in raster order with whole tiles as the unit: measured 12.1 instead of 8.0 ms for 256 samples of the 1080p Cornell
    // box, and far worse where few tiles carry the work. Its first samples are therefore rendered on their own; the costs
    // they record schedule the rest. Samples are folded in the same order either way, so the image does not change.
    const bool same_window = memcmp(ctx->sched_win, ctx->win, sizeof ctx->win) == 0 &&
                             ctx->sched_numbering == (ctx->use_pixel_list ? ctx->pixel_list_id : 0u);
    const Program* current = find_program(ctx, ctx->current_program);
    const bool blind = ctx->sched_mode == 2 && (!ctx->sched_have_cost || !same_window || ctx->costs_stale) &&
                       !(current && current->kind == PT_PROGRAM_CPU_TRACER);
    if (blind && num_samples >= 4 * PT_PROBE_SAMPLES) {
        ctx->sched_have_cost = false;  // stale costs are dropped: the probe starts them afresh
        s = dispatch_range(ctx, first_sample, PT_PROBE_SAMPLES, stride);
        if (s == PT_OK) s = dispatch_range(ctx, first_sample + PT_PROBE_SAMPLES * stride, num_samples - PT_PROBE_SAMPLES, stride);
    } else {
        s = dispatch_range(ctx, first_sample, num_samples, stride);
    }
    if (s != PT_OK) return s;
    if (ctx->timing) {
        PT_HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
        ctx->ev_pending = true;
    }
    ctx->dispatches += 1;
    ctx->dispatches_here += 1;
    ctx->px_samples_here += my_px_samples;
    return PT_OK;
}

PtStatus pt_dispatch_static(PtContext* ctx, int32_t first_sample, int32_t num_samples, int32_t stride) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(first_sample >= 1, PT_ERR_INVALID_ARGUMENT, "first_sample is 1-based; got %d", first_sample);
    PT_REQUIRE(num_samples >= 1 && stride >= 1, PT_ERR_INVALID_ARGUMENT, "num_samples and stride must be >= 1");
    PT_REQUIRE((int64_t)first_sample + (int64_t)(num_samples - 1) * stride <= (1 << 24), PT_ERR_INVALID_ARGUMENT,
               "sample index exceeds 2^24 (u_Samples is a float)");
    PT_REQUIRE(ctx->d_image[0] != nullptr, PT_ERR_STATE, "no accumulation target: call pt_target_create first");
    unsigned char* base = ctx->device < kMaxDevices ? static_cast<unsigned char*>(g_scratch_ptr[ctx->device]) : nullptr;
    PT_REQUIRE(base != nullptr, PT_ERR_STATE, "no scratch array on device %d", ctx->device);
    const size_t tiles = (size_t)PT_TILES_ACROSS(ctx->win[2]) * PT_TILES_ACROSS(ctx->win[3]);
    PT_REQUIRE(tiles <= PT_STATIC_MAX_TILES, PT_ERR_UNSUPPORTED,
               "pt_dispatch_static: the %dx%d window has %zu tiles of %dx%d pixels; its fixed scratch array serves at most %u "
               "(a 3840x2160 window). Use pt_dispatch for larger windows", ctx->win[2], ctx->win[3], tiles, PT_TILE_DIM, PT_TILE_DIM,
               PT_STATIC_MAX_TILES);
    uint32_t* cost = reinterpret_cast<uint32_t*>(base + kStaticCostOff);
    PtWorkItem* items = reinterpret_cast<PtWorkItem*>(base + kStaticItemsOff);
    uint32_t* split_tiles = reinterpret_cast<uint32_t*>(base + kStaticSplitOff);
    uint32_t* counts = reinterpret_cast<uint32_t*>(base + kStaticCountsOff);
    uint32_t* tmp = reinterpret_cast<uint32_t*>(base + kStaticTmpOff);
    float* parked = reinterpret_cast<float*>(base + kStaticSamplesOff);

    // Whatever a dispatch brings up to date for the mesh and the camera position (records, hierarchy boxes, direction
    // grid) is launched here too, but nothing may allocate: a grid that needs new arrays is an error of this call.
    struct NoAlloc {
        PtContext* c;
        explicit NoAlloc(PtContext* ctx) : c(ctx) { c->no_alloc = true; }
        ~NoAlloc() { c->no_alloc = false; }
    } guard(ctx);

    // The array belongs to the GPU, not to the context: a call of ANOTHER context whose launches may still be running would
    // have its work list and parked samples overwritten. Such a call is refused (the same context's calls follow each other
    // on its stream).
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        ScratchUse& use = g_scratch_use[ctx->device];
        if (use.owner != nullptr && use.owner != ctx) {
            const hipError_t q = use.issuing ? hipErrorNotReady : hipEventQuery(use.done);
            if (q == hipErrorNotReady)
                return fail(PT_ERR_STATE, "pt_dispatch_static: another context's pt_dispatch_static is still running on device %d, and all "
                            "contexts of a GPU share one scratch array: wait for it (pt_finish) before this call", ctx->device);
            if (q != hipSuccess) return fail(PT_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q));
        }
        use.owner = ctx;
        use.done = ctx->ev_static;
        use.issuing = true;
    }
    struct Issued {   // whatever was launched is followed by the event the next caller on this GPU looks at
        PtContext* c;
        ~Issued() {
            (void)hipEventRecord(c->ev_static, c->stream);
            std::lock_guard<std::mutex> lock(g_scratch_mutex);
            if (g_scratch_use[c->device].owner == c) g_scratch_use[c->device].issuing = false;
        }
    } issued{ctx};

    PtKernelParams p;
    const Program* prog = nullptr;
    // Everything that can refuse the call for its state (no program, no uniforms, a direction grid that would need new arrays)
    // is found out before anything is launched.
    s = fill_params(ctx, &p, &prog);
    if (s != PT_OK) return s;
    if (ctx->timing) {
        s = collect_timing(ctx);
        if (s != PT_OK) return s;
        PT_HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));   // (before the whole-image copy of a partial window, as in pt_dispatch)
    }
    // Like a pt_dispatch that has no tile costs to go by, a long range renders its first samples on their own and the
    // costs they record schedule the rest -- with the launches sized for the largest list the array can hold instead of
    // the scheduler's counts, which are not read back. Any list that covers every (tile, sample) once renders the image
    // pt_dispatch renders. Should a HIP call fail after the first launch, the image holds the first PT_PROBE_SAMPLES samples
    // of the range and PtStats::samples counts them; PtStats::dispatches is only incremented by a call that succeeded.
    const bool plain = prog->kind == PT_PROGRAM_CPU_TRACER;   // (no work list: one launch)
    const int32_t probe = (ctx->sched_mode == 2 && num_samples >= 4 * PT_PROBE_SAMPLES && !plain) ? PT_PROBE_SAMPLES : 0;
    auto launches = [&]() -> PtStatus {
        PtStatus st = begin_pass(ctx, &p, &prog, first_sample, probe ? probe : num_samples, stride);
        if (st != PT_OK) return st;
        if (probe) {
            // the array keeps what the last call left in it: the costs start at zero
            PT_HIP_TRY(hipMemsetAsync(cost, 0, tiles * sizeof(uint32_t), ctx->stream));
            p.tile_cost = cost;
        }
        st = end_pass(ctx, p, prog, probe ? probe : num_samples);
        if (st != PT_OK || !probe) return st;
        const int32_t rest = num_samples - probe;
        const int entry_bytes = is_materials(ctx->shading_mode) && prog->kind != PT_PROGRAM_WORLD ? 16 : 8;
        const size_t slot_bytes = (size_t)rest * PT_TILE_THREADS * (size_t)entry_bytes;
        st = begin_pass(ctx, &p, &prog, first_sample + probe * stride, rest, stride);
        if (st != PT_OK) return st;
        const float factor = ctx->split_factor >= 0.f ? std::min(ctx->split_factor, (float)PT_MAX_SPLIT_FACTOR)
                                                      : ((p.bvh_nodes || p.bvh_nodes_abs) ? 4.0f : kSpreadSplitFactor);
        size_t max_split = std::min<size_t>((size_t)(factor * (float)ctx->wave_slots), PT_STATIC_MAX_SPLIT);
        max_split = std::min(max_split, kStaticSampleBytes / slot_bytes);  // as many split tiles as the array can park
        if (p.grid_occ && ctx->grid_occ_override < 0)
            PT_HIP_TRY(pt_launch_grid_occ_flag(ctx->d_counters, (unsigned long long)ctx->samples, ctx->d_grid_occ_flag, ctx->stream));
        PT_HIP_TRY(pt_launch_schedule_items(cost, items, counts, split_tiles, tmp, (int)tiles, rest, (int)max_split, ctx->sched_sort, 1, ctx->stream));
        p.items = items;
        p.sched_counts = counts;
        p.split_tiles = split_tiles;
        p.sample_scratch = parked;
        p.max_items = (int32_t)(tiles + max_split);   // upper bounds: blocks beyond the scheduler's counts leave at once
        p.max_split_tiles = (int32_t)max_split;
        return end_pass(ctx, p, prog, rest);
    };
    s = launches();
    if (s != PT_OK) return s;
    if (ctx->timing) {
        PT_HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
        ctx->ev_pending = true;
    }
    ctx->dispatches += 1;
    ctx->dispatches_here += 1;
    return PT_OK;
}

PtStatus pt_finish(PtContext* ctx) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    PT_HIP_TRY(hipGetLastError());
    return collect_timing(ctx);
}

PtStatus pt_readback_rgba32f(PtContext* ctx, float* dst) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(dst != nullptr, PT_ERR_INVALID_ARGUMENT, "dst is null");
    PT_REQUIRE(ctx->d_image[0] != nullptr, PT_ERR_STATE, "no accumulation target");
    const float* img = ctx->d_image[ctx->current ^ 1];
    PT_HIP_TRY(hipMemcpyAsync(dst, img, (size_t)ctx->width * ctx->height * 16, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

PtStatus pt_upload_rgba32f(PtContext* ctx, const float* src) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(src != nullptr, PT_ERR_INVALID_ARGUMENT, "src is null");
    PT_REQUIRE(ctx->d_image[0] != nullptr, PT_ERR_STATE, "no accumulation target");
    // the image the last pass wrote is the one the next pass reads (PathTracer.cpp:1384,1398)
    float* img = ctx->d_image[ctx->current ^ 1];
    PT_HIP_TRY(hipMemcpyAsync(img, src, (size_t)ctx->width * ctx->height * 16, hipMemcpyHostToDevice, ctx->stream));
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

PtStatus pt_present_rgba8(PtContext* ctx, uint8_t* dst) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(dst != nullptr, PT_ERR_INVALID_ARGUMENT, "dst is null");
    PT_REQUIRE(ctx->d_image[0] != nullptr, PT_ERR_STATE, "no accumulation target");
    const size_t n = (size_t)ctx->width * ctx->height;
    // the primary-id buffer (4 bytes per pixel) doubles as the staging area of the 8-bit image
    PT_HIP_TRY(pt_launch_present(ctx->d_image[ctx->current ^ 1], n, reinterpret_cast<uint32_t*>(ctx->d_ids), ctx->stream));
    PT_HIP_TRY(hipMemcpyAsync(dst, ctx->d_ids, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

PtStatus pt_current_image(PtContext* ctx, void** out_device_ptr) {
    PT_REQUIRE(ctx != nullptr && out_device_ptr != nullptr, PT_ERR_INVALID_ARGUMENT, "null argument");
    PT_REQUIRE(ctx->d_image[0] != nullptr, PT_ERR_STATE, "no accumulation target");
    *out_device_ptr = ctx->d_image[ctx->current ^ 1];
    return PT_OK;
}

PtStatus pt_readback_primary(PtContext* ctx, int32_t* ids, float* t) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PtKernelParams p;
    const Program* prog = nullptr;
    s = fill_params(ctx, &p, &prog);
    if (s != PT_OK) return s;
    PT_REQUIRE(prog->kind != PT_PROGRAM_CPU_TRACER, PT_ERR_UNSUPPORTED, "pt_readback_primary serves the shader programs, not PT_PROGRAM_CPU_TRACER");
    if (p.live_count && !p.grid_cells && !p.bvh_nodes) {  // the primary-hit kernel's loop walks the compacted array in either shading mode
        s = ensure_live(ctx);
        if (s != PT_OK) return s;
    }
    p.out_ids = ctx->d_ids;
    p.out_t = ctx->d_t;
    const size_t n = (size_t)ctx->width * ctx->height;
    // pixels outside the window report a miss
    PT_HIP_TRY(hipMemsetAsync(ctx->d_ids, 0xff, n * sizeof(int32_t), ctx->stream));
    PT_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->d_t), 0x7e967699 /* 1e38f */, n, ctx->stream));
    PT_HIP_TRY(pt_launch_primary(p, (int)prog->kind, ctx->stream));
    if (ids) PT_HIP_TRY(hipMemcpyAsync(ids, ctx->d_ids, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (t) PT_HIP_TRY(hipMemcpyAsync(t, ctx->d_t, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

PtStatus pt_enable_timing(PtContext* ctx, int32_t enabled) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    s = collect_timing(ctx);
    if (s != PT_OK) return s;
    ctx->timing = enabled != 0;
    return PT_OK;
}

PtStatus pt_get_stats(PtContext* ctx, PtStats* out) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(out != nullptr, PT_ERR_INVALID_ARGUMENT, "out is null");
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    s = collect_timing(ctx);
    if (s != PT_OK) return s;
    unsigned long long* slots = new (std::nothrow) unsigned long long[(size_t)PT_COUNTER_SLOTS * PT_COUNTER_STRIDE];
    PT_REQUIRE(slots != nullptr, PT_ERR_HIP, "out of host memory");
    hipError_t ce = hipMemcpyAsync(slots, ctx->d_counters, kCounterBytes, hipMemcpyDeviceToHost, ctx->stream);
    if (ce == hipSuccess) ce = hipStreamSynchronize(ctx->stream);
    unsigned long long c[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (ce == hipSuccess)
        for (size_t k = 0; k < PT_COUNTER_SLOTS; ++k)
            for (int j = 0; j < 13; ++j) c[j] += slots[k * PT_COUNTER_STRIDE + j];
    delete[] slots;
    PT_HIP_TRY(ce);
    out->rays = c[0];
    out->rays_traced = c[4];
    out->samples = ctx->samples;
    out->lane_slots = c[1];
    out->bvh_node_fetches = c[2];
    out->bvh_triangle_fetches = c[3];
    out->grid_triangle_tests = c[5];
    out->grid_entries = ctx->grid_built ? ctx->grid_entries : 0;
    out->shaded_hits = c[6];
    out->loop_triangle_visits = c[7];
    out->loop_full_tests = c[8];
    out->lane_walk_records = c[9];
    if (getenv("PT_PRINT_PHASES")) {  // measurement builds (-DPT_V_STATE_COUNT): clock ticks of the wave loop's phases, summed over waves
        fprintf(stderr, "[pt phases] hand-out %llu shade %llu walk %llu ticks\n", c[10], c[11], c[12]);
    }
    out->dispatches = ctx->dispatches;
    out->bvh_sweeps = ctx->bvh_built ? (uint32_t)ctx->bvh_refit_passes : 0u;
    out->grid_lists_carried = ctx->last_grid_carry ? 1u : 0u;
    out->gathered_pixels = ctx->last_gathered ? ctx->pixel_hits : 0u;
    out->gathered_tiles = ctx->last_gathered ? (uint32_t)ctx->pixel_tiles : 0u;
    out->last_dispatch_ms = ctx->last_ms;
    out->total_dispatch_ms = ctx->total_ms;
    return PT_OK;
}

PtStatus pt_reset_stats(PtContext* ctx) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    s = collect_timing(ctx);
    if (s != PT_OK) return s;
    PT_HIP_TRY(hipMemsetAsync(ctx->d_counters, 0, kCounterBytes, ctx->stream));
    PT_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->samples = 0;
    ctx->dispatches = 0;
    ctx->last_ms = 0.f;
    ctx->total_ms = 0.0;
    return PT_OK;
}

PtStatus pt_verify_accel(PtContext* ctx, int32_t bounces, uint64_t result[3]) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(result != nullptr, PT_ERR_INVALID_ARGUMENT, "result is null");
    PT_REQUIRE(bounces >= 0 && bounces <= 64, PT_ERR_INVALID_ARGUMENT, "bounces must be in [0, 64]; got %d", bounces);
    result[0] = result[1] = result[2] = 0;
    PtKernelParams p;
    const Program* prog = nullptr;
    s = fill_params(ctx, &p, &prog);
    if (s != PT_OK) return s;
    PT_REQUIRE(prog->kind == PT_PROGRAM_MESH, PT_ERR_STATE, "pt_verify_accel checks the mesh search structures; the program in use traces spheres");
    PT_REQUIRE(p.num_triangles > 0, PT_ERR_STATE, "u_NumTriangles is 0: nothing to check");
    PT_REQUIRE(!p.cull, PT_ERR_UNSUPPORTED, "PT_ACCEL_CULL has no self-check");
    const bool structure = p.shading_mode != (int32_t)PT_SHADE_REFERENCE ? p.bvh_nodes_abs != nullptr : (p.grid_cells || p.bvh_nodes || p.live_count);
    PT_REQUIRE(structure, PT_ERR_STATE, "the current modes select the plain loop over all triangles: there is no search structure to check "
               "(call pt_build_accel, or trace all uploaded triangles)");
    p.first_sample = 1;
    p.num_samples = 1;
    p.sample_stride = 1;
    p.verify_tris = ctx->d_tris;
    p.verify_bounces = bounces;
    if (!ctx->d_verify_out) PT_HIP_TRY(hipMalloc(&ctx->d_verify_out, 3 * sizeof(unsigned long long)));
    unsigned long long* d = ctx->d_verify_out;
    p.verify_out = d;
    hipError_t e = hipMemsetAsync(d, 0, 3 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) e = pt_launch_verify(p, ctx->stream);
    unsigned long long h[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    PT_HIP_TRY(e);
    for (int k = 0; k < 3; ++k) result[k] = h[k];
    return PT_OK;
}

PtStatus pt_selftest_arithmetic(PtContext* ctx, uint64_t mismatches[3]) {
    PtStatus s = bind(ctx);
    if (s != PT_OK) return s;
    PT_REQUIRE(mismatches != nullptr, PT_ERR_INVALID_ARGUMENT, "mismatches is null");
    unsigned long long* d = nullptr;
    PT_HIP_TRY(hipMalloc(&d, 3 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 3 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) e = pt_launch_arith_selftest(d, ctx->stream);
    unsigned long long h[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    PT_HIP_TRY(e);
    for (int k = 0; k < 3; ++k) mismatches[k] = h[k];
    return PT_OK;
}

}  // extern "C"
#pragma GCC visibility pop
