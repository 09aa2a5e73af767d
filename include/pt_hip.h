/*
 * pt_hip.h -- C ABI of the MI355X (gfx950) path-tracing core.
 *
 * This library stands where the reference host dispatches its GLSL program: everything the fragment shader
 * data/shaders/pathtracing.fs computed per pixel (camera ray, closest hit over 10 spheres or the triangle mesh,
 * modified-Phong bounce with the shader's interleaved-gradient-noise RNG, sky, running-mean accumulation) is
 * done by hand-written HIP kernels behind these entry points. Plain C types only; no exceptions, no aborts:
 * every call returns a PtStatus and pt_last_error() holds the message of the calling thread's last failure.
 *
 * Ownership: the library owns all device memory it allocates. Host pointers passed in are read (or written)
 * during the call only -- uploads copy synchronously, like glTexImage2D, so the caller may free its arrays
 * right after the call returns. A context is bound to one HIP device and one stream and must be driven by one
 * host thread at a time. Multi-GPU: one context per device.
 *
 * Numerics contract: fp32, no fused multiply-add except where the reference's software GL evaluates one, IEEE
 * division and square root, denormals flushed to zero, and the GLSL built-ins sin/cos/exp2/pow evaluated with
 * the same polynomial forms as Mesa llvmpipe. With it, images are bit-identical to the reference shader run
 * on llvmpipe.
 *
 * Each entry point cites the reference interface (file:line under the reference checkout) that it replaces.
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 7

typedef enum PtStatus {
    PT_OK = 0,
    PT_ERR_INVALID_ARGUMENT = 1, /* null pointer, negative size, index out of range, ... */
    PT_ERR_HIP = 2,              /* a HIP runtime call failed; pt_last_error() carries hipGetErrorString */
    PT_ERR_STATE = 3,            /* call sequence error: no target, no program in use, no mesh uploaded, ... */
    PT_ERR_NO_DEVICE = 4,        /* no usable gfx950 device / device index out of range */
    PT_ERR_UNSUPPORTED = 5
} PtStatus;

/* Which closest-hit call the program makes. The reference keeps both calls in its shader and comments one out
 * (pathtracing.fs:395-396): as committed it traces the 10 spheres ("world"); "mesh" is the other line. */
typedef enum PtProgramKind {
    PT_PROGRAM_WORLD = 0, /* IntersectWorld, pathtracing.fs:83-121, call at :395 */
    PT_PROGRAM_MESH = 1,  /* IntersectMesh,  pathtracing.fs:125-174, call at :396 */
    PT_PROGRAM_CPU_TRACER = 2 /* not a shader: the reference's CPU tracer, Scene::color (PathTracer.cpp:932-1052) with scatter
                             (:830-852), compiled out of the reference's own build by #define USE_GLSL (:1246). Spheres AND mesh in
                             one scene (the mesh takes material 0), camera rays through pixel corners (:742-745), Lambert bounce
                             N + randomUnitVector() from the hit point, other material types pass the ray on, sky
                             (1-t) + (1, 0.7, 0.5) t, running SUM per pixel whatever pt_set_accum_mode says. Its random numbers
                             come from a counter-based generator keyed by (seed, pixel, sample index) instead of the reference's
                             serial rand() stream, so its parity with the reference is statistical (per-pixel mean and variance);
                             see pt_set_cpu_tracer. Vertex normals as uploaded; no search structure (the reference's own
                             bounding-box test, then every triangle). */
} PtProgramKind;

/* Shading normal of a mesh hit. */
typedef enum PtNormalMode {
    PT_NORMALS_VERTEX = 0,   /* barycentric blend of the three vertex normals, pathtracing.fs:164-165 */
    PT_NORMALS_GEOMETRIC = 1 /* normalize(cross(e1,e2)), the alternative the shader carries at pathtracing.fs:166;
                                for OBJ files without per-vertex normals (the reference host reads those out of
                                bounds, PathTracer.cpp:1191-1193) */
} PtNormalMode;

/* How the closest mesh hit is searched. Results are identical in every mode (same triangle ids, same distances). */
typedef enum PtAccelMode {
    PT_ACCEL_AUTO = 0,  /* the fastest measured choice. After pt_build_accel: the direction grid for the reference shading (rays
                           all start at the camera), the LBVH above 128 triangles for PT_SHADE_MATERIALS. Without
                           pt_build_accel, or when fewer triangles are traced than uploaded: the plain loop */
    PT_ACCEL_BRUTE = 1, /* the shader's own loop over all triangles in index order, pathtracing.fs:129 */
    PT_ACCEL_BVH = 2,   /* require the LBVH: pt_dispatch fails with PT_ERR_STATE when pt_build_accel has not run (for a mesh of at
                           most 4 triangles pt_build_accel keeps no hierarchy: it would be one leaf, and the loop is its walk) */
    PT_ACCEL_GRID = 4,  /* require the direction grid (built by pt_build_accel, rebuilt when the camera position changes): the
                           shader never moves its ray origin, so every query is "what does the camera position see in this
                           direction"; directions are binned into cube-map cells that list the triangles they can hit.
                           Reference shading only (PT_SHADE_MATERIALS moves the origin), meshes of at most 2^26 triangles
                           (pt_dispatch returns PT_ERR_INVALID_ARGUMENT beyond; PT_ACCEL_AUTO then walks the hierarchy) */
    PT_ACCEL_CULL = 3   /* all-triangles loop that skips, 64 triangles at a time, those no ray of the wave can reach;
                           exact, but measured slower than BRUTE (36 triangles) and BVH (>= 64): kept for experiments */
} PtAccelMode;

/* How a dispatch folds a sample into the target. */
typedef enum PtAccumMode {
    PT_ACCUM_MEAN = 0, /* out = prev + (1/u_Samples)*(sample - prev): the shader's mix(), pathtracing.fs:414 */
    PT_ACCUM_SUM = 1   /* out = prev + sample: running sum, like Framebuffer.h:46,51-52; what sample-split ranks
                          produce before their sums are merged and divided by the sample count */
} PtAccumMode;

/* What Shade() does with a mesh hit. */
typedef enum PtShadingMode {
    PT_SHADE_REFERENCE = 0, /* the shader as committed: the fetched material is overwritten with 0.5 (pathtracing.fs:228-229),
                               every surface is the same modified-Phong lobe, the ray origin never moves */
    PT_SHADE_MATERIALS = 1, /* build-defined: the fetched material (r,g,b,e) is kept. e > 0: emitter of radiance rgb*e, the
                               path ends; otherwise Lambert of albedo rgb, cosine-weighted about the normal with the shader's
                               own GetTangentSpace/randomHemisphere (pathtracing.fs:44-76; the form of its commented lines
                               :269-279), origin moved to the hit point (:274), emitted radiance added next to the sky term
                               (:410). Mesh programs only. Its definition is the shader text with exactly these
                               statements edited in; the tests compare against that text run by a GLSL implementation. */
    PT_SHADE_MATERIALS_UNIT = 2 /* PT_SHADE_MATERIALS with the Lambert bounce in the other form the reference carries as a commented
                               line: rd = normalize(N + randomUnitVector()) (pathtracing.fs:272), randomUnitVector() as at
                               pathtracing.fs:57-65 (z uniform in [-1, 1), then a point on the circle of that height). Everything
                               else as PT_SHADE_MATERIALS. */
} PtShadingMode;

/* The program's uniforms (pathtracing.fs:6-9,17-18), set per frame at PathTracer.cpp:1372-1382.
 * u_Samples (pathtracing.fs:19) is not here: it is the sample index that pt_dispatch() is given. */
typedef struct PtUniforms {
    float camera_position[3];          /* u_CameraPosition */
    float camera_lower_left_corner[3]; /* u_CameraLowerLeftCorner */
    float camera_horizontal[3];        /* u_CameraHorizontal */
    float camera_vertical[3];          /* u_CameraVertical */
    int32_t num_triangles;             /* u_NumTriangles: how many of the uploaded triangles are traced */
    int32_t depth;                     /* u_Depth: bounce limit, 10 in the reference */
} PtUniforms;

/* Filled by pt_get_stats(). Times are HIP-event times on the context's stream. */
typedef struct PtStats {
    uint64_t rays;             /* closest-hit queries since the last pt_reset_stats(): one per bounce-loop iteration
                                  (the reference's definition, PathTracer.cpp:947; its GLSL mode never counts, :1315) */
    uint64_t rays_traced;      /* of those, the queries actually evaluated: with primary reuse (pt_set_primary_reuse) the
                                  camera ray of a pixel is evaluated once per dispatch, not once per sample, and not at
                                  all when an earlier dispatch of the same view left its answer */
    uint64_t samples;          /* pixel-samples (shader invocations) */
    uint64_t lane_slots;       /* 64 x the loop trips of every wave: rays / lane_slots is the fraction of SIMD lanes
                                  that did useful work (paths in one wave end at different times) */
    uint64_t bvh_node_fetches; /* LBVH traversal: 64-byte node records fetched, counted once per WAVE (the walk is per wave) */
    uint64_t bvh_triangle_fetches; /* LBVH traversal: 64-byte triangle records (+4-byte id) fetched, once per wave */
    uint64_t grid_triangle_tests; /* direction grid: triangle tests, counted per ray; per-lane hierarchy walk (PT_SHADE_MATERIALS):
                                  node and triangle records a lane fetched */
    uint64_t grid_entries;     /* direction grid as currently built: (cell, triangle) entries over all cells (0 = not built) */
    uint64_t shaded_hits;      /* Shade() evaluations (pathtracing.fs:224): one per closest-hit query that hit, reused camera-ray
                                  hits included */
    uint64_t loop_triangle_visits; /* all-triangles loop: triangle records visited, counted once per WAVE and loop trip (every lane
                                  of the wave tests its ray against each) */
    uint64_t loop_full_tests;  /* all-triangles loop, measurement builds (-DPT_V_COUNT_FULL) only: of those visits, the ones that
                                  went past the wave-uniform back-face skip; 0 in the product build */
    uint64_t lane_walk_records; /* per-lane hierarchy walk (PT_SHADE_MATERIALS): of the records counted in grid_triangle_tests, the
                                  triangle records (48 bytes each in the compact form; the others are 32-byte nodes) */
    uint32_t dispatches;       /* pt_dispatch calls */
    float last_dispatch_ms;    /* device time of the most recent pt_dispatch (scheduler, render and fold kernels);
                                  0 unless timing is enabled */
    double total_dispatch_ms;  /* sum over dispatches since the last reset; 0 unless timing is enabled */
    uint32_t bvh_sweeps;       /* bottom-up sweeps the last pt_build_accel ran to box its hierarchy, a multiple of 8: an upper bound
                                  of the hierarchy's height (0 = no hierarchy) */
    uint32_t grid_lists_carried; /* direction grid: which form of the render loop the last dispatch through the grid used. 0: the
                                  wave runs its list loop as often as its longest list; 1: a lane tests one triangle per trip and
                                  carries the rest of its cell's list into the next. Same image; the library times both on a
                                  view's first long dispatches and keeps the quicker (the field was reserved before ABI 6) */
    uint32_t gathered_pixels;  /* sparse views (since ABI 7): when the last pt_dispatch rendered the pixels whose camera ray hits the
                                  mesh gathered into tiles of 64 instead of the window's 8x8 blocks (the other pixels are folded by a
                                  pass of their own; same image), how many such pixels the view has; 0 = it rendered the blocks */
    uint32_t gathered_tiles;   /* ... and how many tiles they make */
} PtStats;

typedef struct PtContext PtContext; /* opaque */

/* ---- library / context ------------------------------------------------------------------------------- */

int32_t pt_abi_version(void);
/* Message of the calling thread's most recent failing call ("" if none). Never NULL. */
const char* pt_last_error(void);
/* Number of HIP devices visible to the process. */
PtStatus pt_device_count(int32_t* count);
/* Replaces window + GL context creation (PathTracer.cpp:1072-1112): binds device `device`, creates a stream. */
PtStatus pt_create(int32_t device, PtContext** out_ctx);
/* Replaces the teardown at PathTracer.cpp:1502-1517. Frees every device allocation of the context. */
PtStatus pt_destroy(PtContext* ctx);
/* Use an existing HIP stream (hipStream_t passed as void*; NULL = back to the context's own stream). The caller
 * keeps ownership of the stream. For hosts that already order their own work on a stream (e.g. torch). */
PtStatus pt_set_stream(PtContext* ctx, void* hip_stream);

/* ---- program object: stands in for Shader (Shader.h:18-27) ---------------------------------------------- */

/* Replaces Shader::loadFragmentShader + createProgram (Shader.cpp:361-373, 86-107; called at
 * PathTracer.cpp:1248-1251): selects the kernel family. *out_program is a non-zero handle. */
PtStatus pt_program_create(PtContext* ctx, PtProgramKind kind, uint32_t* out_program);
/* Replaces Shader::destroyProgram (Shader.cpp:59-66). */
PtStatus pt_program_destroy(PtContext* ctx, uint32_t program);
/* Replaces glUseProgram(program) (PathTracer.cpp:1371). */
PtStatus pt_use_program(PtContext* ctx, uint32_t program);

/* ---- scene upload: replaces the four glTexImage2D uploads (PathTracer.cpp:1253-1288) -------------------- */

/* count spheres as (cx,cy,cz,r); the shader always loops over 10 (pathtracing.fs:87), so count must be 10.
 * Replaces the spheres texture, PathTracer.cpp:1253-1258. */
PtStatus pt_upload_spheres(PtContext* ctx, const float* xyzr, int32_t count);
/* count materials as (r,g,b,type). Kept resident for interface parity: the shader fetches the material and then
 * overwrites it with vec4(0.5) (pathtracing.fs:228-229). Replaces PathTracer.cpp:1260-1265. */
PtStatus pt_upload_materials(PtContext* ctx, const float* rgbt, int32_t count);
/* Mesh in the reference's layouts: `vertices` = num_vertices x {position xyz, normal xyz} (struct Vertex,
 * PathTracer.cpp:71-75), `indices` = num_triangles x 3 vertex indices. Replaces the index and vertex textures
 * (PathTracer.cpp:1267-1288) without their 1024-texel limit. On the device the mesh becomes per-triangle
 * structure-of-arrays records (edges, normals) in HBM; triangle id = position in `indices`. */
PtStatus pt_upload_mesh(PtContext* ctx, const float* vertices, int32_t num_vertices, const uint32_t* indices,
                        int32_t num_triangles);
PtStatus pt_set_normal_mode(PtContext* ctx, PtNormalMode mode);
/* One row index into the material table per triangle of the uploaded mesh (count must equal its triangle count; ids
 * >= 0). This is tinyobj's mesh_t::material_ids (tiny_obj_loader.h:207), which the reference's loader fills and its host
 * then drops: its shader returns 0.0 as the hit index of every triangle (pathtracing.fs:167). Used by
 * PT_SHADE_MATERIALS only. A new pt_upload_mesh discards the ids. */
PtStatus pt_upload_material_ids(PtContext* ctx, const int32_t* ids, int32_t count);
PtStatus pt_set_shading_mode(PtContext* ctx, PtShadingMode mode);
/* The shader has no pixel jitter (pathtracing.fs:381-383: the camera ray depends on gl_FragCoord only), so the first
 * closest-hit query of every sample of a pixel has the same answer. enable != 0 (the default) evaluates it once per
 * dispatch and pixel and starts every sample of that dispatch from it; 0 evaluates it for every sample like the shader
 * does. The image is identical either way; PtStats::rays counts the query for every sample in both cases,
 * PtStats::rays_traced only the evaluations. No effect on dispatches of a single sample. */
PtStatus pt_set_primary_reuse(PtContext* ctx, int32_t enable);
/* Builds a bounding volume hierarchy over the uploaded mesh (and makes the direction grid available, see PT_ACCEL_GRID). No
 * counterpart in the reference, whose shader tests every triangle for every ray; needed for scenes beyond a few thousand
 * triangles. A new pt_upload_mesh discards it. Meshes of at most 4 triangles need none (PT_OK, no-op). Synchronises. */
PtStatus pt_build_accel(PtContext* ctx);
/* Which builder the next pt_build_accel uses. Images do not depend on it (the walk only ever skips triangles the exact test
 * could not accept); the time of the walks and of the build does. */
typedef enum PtBvhBuilder {
    PT_BVH_BUILDER_SAH = 0,    /* top-down on the HOST, splits by the binned surface-area heuristic, up to 16 threads;
                                  0.09 s for a million triangles */
    PT_BVH_BUILDER_MORTON = 1, /* on the device: Morton codes, radix sort, Karras hierarchy; the quickest build, and walks that
                                  fetch 10-20 % more records than over the other two */
    PT_BVH_BUILDER_PLOC = 2,   /* on the device, the quickest build whose walks stay near the surface-area builders': triangles in Morton order merged bottom-up, each cluster with the
                                  neighbour (within 16 places along the curve) whose merge adds the least area, then re-ordered
                                  depth first; meshes on which that makes no headway (degenerate layouts) get the Morton-code
                                  hierarchy */
    PT_BVH_BUILDER_BINNED = 3  /* default, on the device: PT_BVH_BUILDER_SAH's rules (16 bins per axis, area x count), the large ranges
                                  level by level, ranges of at most 64 triangles by one wave each; the same tree as the host
                                  builder's for meshes below 131,072 triangles (above, the two sample different triangles for the
                                  first planes); 13 ms for a million triangles */
} PtBvhBuilder;
PtStatus pt_set_bvh_builder(PtContext* ctx, PtBvhBuilder builder);
PtStatus pt_set_accel_mode(PtContext* ctx, PtAccelMode mode);

/* ---- accumulation target: replaces the two RGBA32F textures + FBOs (PathTracer.cpp:1136-1152) ------------ */

/* Two width x height RGBA32F images (row 0 = bottom row, like GL), cleared to (0,0,0,1). Each pass reads one and
 * writes the other, then they swap (PathTracer.cpp:1306,1384,1398,1493). */
PtStatus pt_target_create(PtContext* ctx, int32_t width, int32_t height);
/* Same, but in caller-owned device memory (two buffers of width*height*4 floats, 16-byte aligned): lets a host
 * that manages device memory itself (torch tensors, RCCL buffers) merge results without a copy. The buffers are
 * NOT cleared. `rgba_b` may equal `rgba_a` for in-place accumulation. */
PtStatus pt_target_bind_external(PtContext* ctx, void* rgba_a, void* rgba_b, int32_t width, int32_t height);
/* glClear with the reference's clear colour (0,0,0,1) on both images (PathTracer.cpp:1108,1151); with
 * PT_ACCUM_SUM in effect the clear colour is (0,0,0,0). */
PtStatus pt_target_clear(PtContext* ctx);
/* Restricts rendering to the pixel window [x0,x0+w) x [y0,y0+h) (glScissor semantics); pixels outside keep their
 * value. Used for the image-tile partition across GPUs. w = h = 0 restores the full frame. */
PtStatus pt_set_window(PtContext* ctx, int32_t x0, int32_t y0, int32_t w, int32_t h);
/* No counterpart in the reference (one GPU, whole frame). Image-tile partition across `count` renderers in ONE dispatch
 * each: the image rows are cut into blocks of `block_rows`, dealt round-robin, and this context renders only the blocks
 * whose number is `index` modulo `count` (inside the current window); all other pixels keep their value. Interleaving
 * shares out the expensive regions of an image. Every renderer runs all samples of its pixels with the reference
 * recurrence, so its pixels are bit-identical to a single-renderer image; images that are zero outside the owned rows are
 * merged exactly by a sum. count <= 1 switches it off. */
PtStatus pt_set_row_interleave(PtContext* ctx, int32_t block_rows, int32_t index, int32_t count);
PtStatus pt_set_accum_mode(PtContext* ctx, PtAccumMode mode);
/* PT_PROGRAM_CPU_TRACER only: seed of its random number generator (default 1) and the camera ray's maxt, far - near of
 * Camera::initialize (PathTracer.cpp:709-716, 745; default 100 - 0.1 as at :1161). Stands where the reference seeds the stream
 * its render loop draws from, srand(2 ^ 24 - 1) at PathTracer.cpp:1297 (an XOR: srand(21)); the value has no correspondence to
 * the reference's, because the generator here is counter-based and not libc's. */
PtStatus pt_set_cpu_tracer(PtContext* ctx, uint32_t seed, float ray_max_t);

/* ---- per-frame: replaces PathTracer.cpp:1370-1403 ------------------------------------------------------ */

/* Replaces the glUniform* block, PathTracer.cpp:1372-1382. */
PtStatus pt_set_uniforms(PtContext* ctx, const PtUniforms* uniforms);
/* Replaces bind + glDrawArrays (PathTracer.cpp:1384-1402), asynchronously on the context's stream.
 * Renders `num_samples` consecutive passes u_Samples = first_sample, first_sample + stride, ... for every pixel of
 * the window and leaves the result in the current image, exactly as that many reference frames would (the
 * running mean is carried in registers between passes instead of through memory). num_samples = 1, stride = 1
 * is one reference frame. first_sample is 1-based (PathTracer.cpp:1305,1378).
 * Not fully asynchronous: when the work list is rebuilt (second dispatch of a view, then about every 16th; never for
 * num_samples = 1) two counters are read back, which waits for the stream, and scratch memory may be allocated. A
 * dispatch of 32 samples or more that has no tile costs to go by is issued as two launches (8 samples, then the rest).
 * The first dispatch after a change of mesh or camera position also rebuilds what depends on them (triangle records,
 * direction grid: one more read-back). With primary reuse and PT_SHADE_REFERENCE the camera ray's hit of every pixel is
 * kept (16 bytes per pixel) until uniforms, window, row partition, modes, program or mesh change, so single-sample
 * dispatches of a standing camera do not evaluate it again; the images are the same either way. */
PtStatus pt_dispatch(PtContext* ctx, int32_t first_sample, int32_t num_samples, int32_t stride);
/* pt_dispatch for callers that cannot allocate. No counterpart in the reference: it is for hosts that may not call
 * hipMalloc where they render -- tight serving loops, code that forbids allocation, harnesses that check for it. Same
 * arguments, same image bit for bit, but the call allocates nothing (no hipMalloc, hipFree or allocating library routine):
 * per-tile costs, the work list and the parked samples of split tiles live in a fixed 64 MiB array that is part of the
 * library's device image (one copy per GPU, held until the process ends) instead of the context's grown-on-demand buffers,
 * and the scheduler's counts are not read back -- launches are sized for the largest list the array holds.
 * Limits: a window of at most 131,072 tiles of 8x8 pixels (3840x2160 fits; larger windows: PT_ERR_UNSUPPORTED); the parked
 * samples get 63 MiB, so a range of n samples can split at most 63 MiB / (n x 512 B) tiles (1 KiB in PT_SHADE_MATERIALS),
 * fewer than pt_dispatch would: slower on views whose work sits in few tiles, never different. Everything the call needs
 * must exist already: target, scene, and -- for the direction grid -- arrays large enough for the current camera position
 * (one pt_dispatch of the view makes them; otherwise PT_ERR_STATE and nothing is rendered). It keeps no tile costs between
 * calls: a range of 32 samples or more is two launches (8 samples, then the rest scheduled by their costs), shorter ones one.
 * An error found after the first launch of a long range (a HIP failure) leaves the first 8 samples of the range folded into the
 * image; PtStats::dispatches counts only calls that succeeded.
 * All contexts on a GPU share that GPU's array. A call made while ANOTHER context's pt_dispatch_static is still running on the
 * same GPU returns PT_ERR_STATE and renders nothing (wait for that context with pt_finish first); calls of one context follow
 * each other on its stream. */
PtStatus pt_dispatch_static(PtContext* ctx, int32_t first_sample, int32_t num_samples, int32_t stride);
/* Replaces glFinish (PathTracer.cpp:1403): waits for the stream and reports any kernel fault. */
PtStatus pt_finish(PtContext* ctx);

/* ---- results (new: the reference never reads back, it only presents) ------------------------------------ */

/* Copies the current image (the one the last pass wrote) to host: width*height*4 floats, row 0 = bottom. */
PtStatus pt_readback_rgba32f(PtContext* ctx, float* dst);
/* The reverse of pt_readback_rgba32f: `src` (width*height*4 floats, row 0 = bottom) becomes the current image, the one the next
 * pass reads as its previous result. The reference cannot do this (its accumulation textures, PathTracer.cpp:1136-1152, are only
 * ever rendered into), but those textures and the frame counter `samples` (PathTracer.cpp:1300,1305) are all the state its
 * progressive render has: a host that saved an image after n passes continues it, in this or another context, by uploading it
 * and dispatching with first_sample = n + 1 -- the result equals an uninterrupted run bit for bit. Synchronous like the scene
 * uploads. */
PtStatus pt_upload_rgba32f(PtContext* ctx, const float* src);
/* Replaces the present quad (PathTracer.cpp:1473-1491 with data/shaders/copy.fs:12-20): display colour
 * pow(rgb, 0.454545) of the current image, converted to 8 bits per channel as an RGBA8 render target does, copied to
 * host as width*height RGBA bytes, row 0 = bottom. The reference draws this into its window; a host shows or saves it. */
PtStatus pt_present_rgba8(PtContext* ctx, uint8_t* dst);
/* Device address of the current image (for hosts that consume it on the device). */
PtStatus pt_current_image(PtContext* ctx, void** out_device_ptr);
/* Traces primary rays only and copies per-pixel closest-hit ids (int32; -1 = miss; sphere id for
 * PT_PROGRAM_WORLD, triangle id for PT_PROGRAM_MESH) and hit distances (1e38 on a miss). Either may be NULL. Only the
 * pixels of the window (and of this context's row blocks, pt_set_row_interleave) are traced; all others report a miss. */
PtStatus pt_readback_primary(PtContext* ctx, int32_t* ids, float* t);
PtStatus pt_enable_timing(PtContext* ctx, int32_t enabled);
PtStatus pt_get_stats(PtContext* ctx, PtStats* out);
PtStatus pt_reset_stats(PtContext* ctx);
/* No counterpart in the reference (its shader only has the loop). Self-check of the search structure the current modes
 * select for a dispatch -- direction grid, LBVH (per-wave or per-lane walk), or the loop's compacted record array: every
 * camera ray of the window, and `bounces` generations of the rays Shade() makes of the hits (sample index 1), are answered by
 * that structure AND by the shader's own search, the loop over every uploaded triangle in index order (pathtracing.fs:129-168),
 * on the device. result[0] = queries compared, [1] = queries whose closest triangle differs, [2] = queries whose hit distance
 * differs in any bit. [1] and [2] are 0 when the structure is exact, which is what keeps images identical to the shader's.
 * Costs one all-triangles search per query: seconds for a 4K frame over a million triangles. PT_ERR_STATE when the modes
 * select no structure (plain loop), PT_ERR_UNSUPPORTED for PT_ACCEL_CULL. */
PtStatus pt_verify_accel(PtContext* ctx, int32_t bounces, uint64_t result[3]);
/* No counterpart in the reference. The kernels form 1/x, sqrt(x) and the quotient inside their log2 with shorter
 * instruction sequences than the compiler's IEEE division and square root; the results are required to be the same bit
 * for bit (that is what keeps images identical to the shader's on llvmpipe). This runs every one of the 2^32 float
 * inputs (2^23 mantissas for the quotient) through both on the context's device and returns how many differ:
 * mismatches[0] reciprocal, [1] square root, [2] quotient. All three are 0 on a device the library supports.
 * Takes about a tenth of a second. */
PtStatus pt_selftest_arithmetic(PtContext* ctx, uint64_t mismatches[3]);

#ifdef __cplusplus
}
#endif
#endif /* PT_HIP_H */
