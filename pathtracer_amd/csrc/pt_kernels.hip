// HIP kernels for gfx950 (MI355X): everything the reference's fragment shader data/shaders/pathtracing.fs
// ("fs" below) does per pixel. Build flags matter and are part of the numerics contract (see build.py):
//   -ffp-contract=off                 a*b+c is two roundings unless written as __builtin_fmaf
//   -fgpu-flush-denormals-to-zero     the reference's software GL runs with FTZ+DAZ
//   (default) correctly rounded fp32 divide and sqrt
//
// Execution shape: one lane owns one pixel for the whole dispatch and walks through its samples and bounces in
// a single flattened loop whose trip = one closest-hit query ("ray"). A lane whose path ends (miss, depth limit)
// folds the sample into its running mean and starts its next sample in the same iteration, so the lanes of a
// wave stay busy although paths have different lengths; the running mean lives in registers and the image is
// read once and written once per dispatch, not once per sample. Triangle records are wave-uniform: the loop
// over triangles fetches each 64-byte record with scalar loads and every lane tests its own ray against it.
#include "pt_internal.h"

#include <cstdlib>

namespace {

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 V(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
// GLSL built-ins in the evaluation order of the reference's software GL:
//   dot(a,b) = a.x*b.x + (a.y*b.y + a.z*b.z);  normalize(v) = v * (1/sqrt(dot(v,v)));  reflect(I,N) = I - (2 dot(N,I)) N
__device__ __forceinline__ float dot3(v3 a, v3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ v3 cross3(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// 1.0f / x, correctly rounded like the IEEE division, in four instructions instead of the division's twelve: hardware
// reciprocal (1 ulp), one Newton step in fused arithmetic, and the division's own fix-up for zero, infinity and NaN.
// Bit-identical to `1.0f / x` for every one of the 2^32 inputs under this build's flags (pt_arith_selftest_kernel runs them all).
__device__ __forceinline__ float ieee_rcp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    const float q = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    return __builtin_amdgcn_div_fixupf(q, x, 1.0f);
}
// sqrtf(x), correctly rounded: the compiler's own refinement (reciprocal square root, coupled Newton step, residual
// correction) without its rescaling of tiny inputs and without its selects for 0 and infinity: finite inputs from 2^-96 up take
// this way (one unsigned compare decides), every other bit pattern (zeros, tiny, negative, infinite, NaN -- rare on the render
// path) takes the compiler's full sequence. Bit-identical to __builtin_sqrtf for all 2^32 inputs (pt_arith_selftest_kernel).
__device__ __forceinline__ float ieee_sqrt(float x) {
    if ((__float_as_uint(x) - 0x0f800000u) >= 0x70000000u) return __builtin_sqrtf(x);
    const float r = __builtin_amdgcn_rsqf(x);
    float s = x * r;
    float h = 0.5f * r;
    const float e = __builtin_fmaf(-h, s, 0.5f);
    s = __builtin_fmaf(s, e, s);
    h = __builtin_fmaf(h, e, h);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
__device__ __forceinline__ v3 normalize3(v3 a) { return a * ieee_rcp(ieee_sqrt(dot3(a, a))); }
__device__ __forceinline__ v3 reflect3(v3 i, v3 n) { return i - n * (2.0f * dot3(n, i)); }
__device__ __forceinline__ float fract1(float x) { return x - __builtin_floorf(x); }
__device__ __forceinline__ float clamp01(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }

// ---- sin / cos / exp2 / pow in the polynomial forms the reference's software GL uses ------------------------
// Range reduction by multiples of pi/4 with a three-part constant, then a degree-3 polynomial in x^2 for the
// sine or the cosine octant; fused multiply-adds exactly where written.
template <bool COS>
__device__ __forceinline__ float lp_sincos(float a) {
    const uint32_t ai = __float_as_uint(a);
    const float x_abs = __uint_as_float(ai & 0x7fffffffu);
    const float scale_y = x_abs * 1.27323954473516f;
    const int32_t j1 = (int32_t)scale_y + 1;
    const int32_t je = j1 & ~1;
    const float y = (float)je;
    const int32_t sel = COS ? je - 2 : je;
    const uint32_t sign = COS ? (((uint32_t)(4 & ~sel)) << 29) : ((ai ^ ((uint32_t)j1 << 29)) & 0x80000000u);
    float x = __builtin_fmaf(y, -0.78515625f, x_abs);
    x = __builtin_fmaf(y, -2.4187564849853515625e-4f, x);
    x = __builtin_fmaf(y, -3.77489497744594108e-8f, x);
    const float z = x * x;
    float c = __builtin_fmaf(z, 2.443315711809948E-005f, -1.388731625493765E-003f);
    c = __builtin_fmaf(c, z, 4.166664568298827E-002f);
    c = c * z;
    c = c * z;
    c = c - z * 0.5f;
    c = c + 1.0f;
    float s = __builtin_fmaf(z, -1.9515295891E-4f, 8.3321608736E-3f);
    s = __builtin_fmaf(s, z, -1.6666654611E-1f);
    s = s * z;
    s = __builtin_fmaf(s, x, x);
    const float r = __uint_as_float(__float_as_uint((sel & 2) == 0 ? s : c) ^ sign);
    return __builtin_fminf(__builtin_fmaxf(r, -1.0f), 1.0f);
}

// exp2: split into floor and fraction, degree-5 polynomial on the fraction evaluated as even/odd Horner chains.
// CLAMP_HIGH / CLAMP_LOW: the clamps to 128 and -126.99999; a caller whose argument cannot reach a bound leaves that clamp out
// (the result is the same for every argument inside the bound; pt_arith_selftest_kernel compares the forms over the ranges used).
template <bool CLAMP_HIGH = true, bool CLAMP_LOW = true>
__device__ __forceinline__ float lp_exp2(float x) {
    if (CLAMP_HIGH) x = __builtin_fminf(128.0f, x);
    if (CLAMP_LOW) x = __builtin_fmaxf(-126.99999f, x);
    const float ip = __builtin_floorf(x);
    const float f = x - ip;
    const float e = __uint_as_float((uint32_t)(((int32_t)ip + 127) << 23));
    const float f2 = f * f;
    float even = __builtin_fmaf(f2, 0.00898934009049466391101f, 0.240153617044375388211f);
    even = __builtin_fmaf(f2, even, 1.0f);
    float odd = __builtin_fmaf(f2, 0.00187757667519147912699f, 0.0558263180532956664775f);
    odd = __builtin_fmaf(f2, odd, 0.693153073200168932794f);
    return e * __builtin_fmaf(odd, f, even);
}

// (m - 1) / (m + 1) for a mantissa m in [1, 2): refined reciprocal, quotient, one residual correction -- equal to the IEEE
// quotient for every one of the 2^23 mantissas (pt_arith_selftest_kernel).
__device__ __forceinline__ float log2_quotient(float m) {
    const float num = m - 1.0f, den = m + 1.0f;
    float r = __builtin_amdgcn_rcpf(den);
    r = __builtin_fmaf(__builtin_fmaf(-den, r, 1.0f), r, r);
    const float q = num * r;
    return __builtin_fmaf(__builtin_fmaf(-den, q, num), r, q);
}

// log2: exponent + y*P(y^2) with y = (m-1)/(m+1), m the mantissa in [1,2); degree-4 P.
// UNIT: the argument is a rand01() result, i.e. in [0, 1) or NaN: of the three special cases only x == 0 can occur (a NaN fails
// all three comparisons either way), so the other two selects are left out.
template <bool UNIT = false>
__device__ __forceinline__ float lp_log2(float x) {
    const uint32_t i = __float_as_uint(x);
    const float e = (float)((int32_t)((i & 0x7f800000u) >> 23) - 127);
    const float m = __uint_as_float((i & 0x007fffffu) | 0x3f800000u);
    const float y = log2_quotient(m);
    const float z = y * y;
    const float z2 = z * z;
    float even = __builtin_fmaf(z2, 0.406718052498846252698f, 0.577440339438736392009f);
    even = __builtin_fmaf(z2, even, 2.88539009343309178325f);
    float odd = __builtin_fmaf(z2, 0.403343858251329912514f, 0.961791550404184197881f);
    float r = __builtin_fmaf(y, __builtin_fmaf(odd, z, even), e);
    if (!UNIT && x >= __builtin_inff()) r = __builtin_inff();
    if (x == 0.0f) r = -__builtin_inff();
    if (!UNIT && x < 0.0f) r = __builtin_nanf("");
    return r;
}
__device__ __forceinline__ float lp_pow(float x, float y) { return lp_exp2(lp_log2(x) * y); }
// pow(x, y) for x = a rand01() result and 0 < y <= 1 (the two calls of the shading code): log2(x) is -inf or a finite value
// <= 0 (for a NaN x: the finite value its bits give, below 130), so the product never comes near the upper clamp.
__device__ __forceinline__ float lp_pow_unit(float x, float y) { return lp_exp2<false, true>(lp_log2<true>(x) * y); }

// ---- the shader's random numbers (fs:21-42) -------------------------------------------------------------
struct Rng {
    float fcx, fcy;  // gl_FragCoord.xy
    float samples;   // u_Samples
    float sx;        // randState.x (randState.y is written by the shader but never read)
    // fs:27-36, then mod(r, 1.0) = r - floor(r) (fs:41)
    __device__ __forceinline__ float rand01() {
        const float ax = (fcx + sx) + samples;
        const float ay = (fcy + sx) + samples;
        const float r = fract1(52.9829189f * fract1(ax * 0.06711056f + ay * 0.00583715f));
        sx = sx + r;
        // fs:41, mod(r, 1.0) = r - floor(r), leaves r as it is: r is x - floor(x) of a product of 52.98 and a value in [0, 1],
        // which is exact and below 1 for every such x (a NaN stays the same NaN); pt_arith_selftest_kernel checks every input
        return r;
    }
};

struct Hit {
    float t;    // recT; 1e38 when nothing was hit
    int id;     // -1 = miss
    float u, v, ood;  // mesh only: the two barycentric numerators and 1/det of the accepted triangle
};

// One ray against one triangle record: fs:139-168 with the direction-independent parts taken from the record
// (see pt_internal.h). The record registers are wave-uniform (scalar).
struct TriRec { float4 a, b, c, d; };  // e1.xyz e2.x | e2.yz ap.xy | ap.z r.xyz | e2r - - -

// `full` counts the tests that went past the wave-uniform back-face skip -- only in builds with -DPT_V_COUNT_FULL
// (a measurement variant: the add costs a vector instruction per test); otherwise it stays 0.
template <bool ANY_ORDER>
__device__ __forceinline__ void test_triangle(const TriRec& t, int id, v3 rd, Hit& h, unsigned& full) {
    const v3 e1 = V(t.a.x, t.a.y, t.a.z);
    const v3 e2 = V(t.a.w, t.b.x, t.b.y);
#ifdef PT_V_VFIRST
    // v does not depend on the cross product: computed first, and when no lane has v >= 0 nothing else is needed
    const v3 r0 = V(t.c.y, t.c.z, t.c.w);
    const float v0 = dot3(rd, r0);
    if (!__builtin_amdgcn_ballot_w64(v0 >= 0.0f)) return;
#endif
    const v3 q = cross3(rd, e2);
    const float det = dot3(e1, q);
#ifndef PT_V_NO_DETSKIP
    // wave-uniform skip: when no lane sees the front of the triangle the two remaining dot products are not needed
#ifdef PT_V_VFIRST
    if (!__builtin_amdgcn_ballot_w64((det >= 0.0001f) & (v0 >= 0.0f))) return;
#else
    if (!__builtin_amdgcn_ballot_w64(det >= 0.0001f)) return;
#endif
#endif
#ifdef PT_V_COUNT_FULL
    ++full;
#endif
    const v3 ap = V(t.b.z, t.b.w, t.c.x);
    const float u = dot3(ap, q);
#ifdef PT_V_VFIRST
    const float v = v0;
#else
    const v3 r = V(t.c.y, t.c.z, t.c.w);
    const float v = dot3(rd, r);
#endif
    // fs:142,151. "u <= det" is implied by the other three (v >= 0 makes fl(u+v) >= u), so it is not evaluated.
    const bool inside = (det >= 0.0001f) & (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= det);
    if (inside) {
        const float ood = ieee_rcp(det);
        const float tt = t.d.x * ood;
        // The shader visits triangles in ascending id with a strict '<', so among equal distances the lowest id wins.
        // When triangles arrive in another order (BVH leaves) that rule is applied explicitly.
        const bool closer = ANY_ORDER ? (tt < h.t || (tt == h.t && id < h.id)) : (tt < h.t);
        if (tt > 0.0f && closer) {
            h.t = tt; h.id = id; h.u = u; h.v = v; h.ood = ood;
        }
    }
}

// The same test for a ray with its own origin (the "materials" shading mode moves the origin to each hit point, as the
// shader's commented line fs:274 does): fs:136-168 as written, from a record that holds a, e1, e2 and the triangle's
// material id:  a.xyz e1.x | e1.yz e2.xy | e2.z matid - - | -
template <bool ANY_ORDER>
__device__ __forceinline__ void test_triangle_from(const TriRec& t, int id, v3 ro, v3 rd, Hit& h, unsigned& full, bool* accepted = nullptr) {
    const v3 a = V(t.a.x, t.a.y, t.a.z);
    const v3 e1 = V(t.a.w, t.b.x, t.b.y);
    const v3 e2 = V(t.b.z, t.b.w, t.c.x);
    const v3 q = cross3(rd, e2);
    const float det = dot3(e1, q);
    if (!__builtin_amdgcn_ballot_w64(det >= 0.0001f)) return;
#ifdef PT_V_COUNT_FULL
    ++full;
#endif
    const v3 ap = ro - a;
    const v3 r = cross3(ap, e1);
    const float u = dot3(ap, q);
    const float v = dot3(rd, r);
    const bool inside = (det >= 0.0001f) & (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= det);
    if (inside) {
        const float ood = ieee_rcp(det);
        const float tt = dot3(e2, r) * ood;
        const bool closer = ANY_ORDER ? (tt < h.t || (tt == h.t && id < h.id)) : (tt < h.t);
        if (accepted && tt > 0.0f) *accepted = true;   // (measurement builds)
        if (tt > 0.0f && closer) {
            h.t = tt; h.id = id; h.u = u; h.v = v; h.ood = ood;
        }
    }
}

// Read-only scene data fetched with a wave-uniform address (triangle and node records, work items) is read through the
// constant address space: such loads become scalar loads (one fetch per wave into scalar registers) no matter what the
// kernel stores elsewhere. Nothing a render kernel writes is ever read through these pointers.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(4))) kf32x4;
typedef const uint32_t __attribute__((address_space(4))) kuint32;
struct ScalarView {
    const kf32x4* p;
    __device__ __forceinline__ float4 operator[](size_t i) const {
        const f32x4 v = p[i];
        return make_float4(v.x, v.y, v.z, v.w);
    }
};
__device__ __forceinline__ ScalarView scalar_view(const float4* p) { ScalarView s; s.p = (const kf32x4*)p; return s; }

__device__ __forceinline__ TriRec load_rec(const float4* __restrict__ recs, int i) {
    const ScalarView k = scalar_view(recs);
    TriRec t;
    t.a = k[4 * i]; t.b = k[4 * i + 1]; t.c = k[4 * i + 2]; t.d = k[4 * i + 3];
    return t;
}

// fs:125-174: closest hit over triangles 0..n-1 in ascending order. The record address is the same in every lane,
// so records arrive through scalar loads; they are requested two triangles ahead of their use (two register sets,
// alternating), which keeps the scalar-load latency off the wave's critical path. The buffer is padded with
// PT_TRI_PAD all-zero records (det = 0 fails the cull), so the loop needs no remainder handling.
__device__ __forceinline__ Hit intersect_mesh(const float* __restrict__ tris, int n, v3 rd, unsigned& full) {
    Hit h;
    h.t = 1.0e38f; h.id = -1; h.u = 0.f; h.v = 0.f; h.ood = 0.f;
    const float4* __restrict__ recs = reinterpret_cast<const float4*>(tris);
    // the triangle's index travels in the record's last word: the array may be the compacted one (see
    // pt_compact_triangles_kernel), which keeps the ascending order the strict '<' relies on
    auto id_of = [](const TriRec& t) { return (int)__float_as_uint(t.d.w); };
#ifdef PT_V_SIMPLE
    for (int i = 0; i < n; ++i) { const TriRec t = load_rec(recs, i); test_triangle<false>(t, id_of(t), rd, h, full); }
#else
    TriRec a0 = load_rec(recs, 0), a1 = load_rec(recs, 1);
    for (int i = 0; i < n; i += 4) {
        const TriRec b0 = load_rec(recs, i + 2), b1 = load_rec(recs, i + 3);
        test_triangle<false>(a0, id_of(a0), rd, h, full);
        test_triangle<false>(a1, id_of(a1), rd, h, full);
        a0 = load_rec(recs, i + 4);
        a1 = load_rec(recs, i + 5);
        test_triangle<false>(b0, id_of(b0), rd, h, full);
        test_triangle<false>(b1, id_of(b1), rd, h, full);
    }
#endif
    return h;
}

// how many records the all-triangles loop visits: all of them, or the count the compaction left on the device
__device__ __forceinline__ int live_triangles(const PtKernelParams& p) {
    return p.live_count ? (int)*(const kuint32*)p.live_count : p.num_triangles;
}

// fs:125-174 for rays with their own origins: same double-buffered scalar record stream.
__device__ __forceinline__ Hit intersect_mesh_from(const float* __restrict__ tris_abs, int n, v3 ro, v3 rd, unsigned& full) {
    Hit h;
    h.t = 1.0e38f; h.id = -1; h.u = 0.f; h.v = 0.f; h.ood = 0.f;
    const float4* __restrict__ recs = reinterpret_cast<const float4*>(tris_abs);
#ifdef PT_V_GROUP_COUNT
    // measurement build: how often NO lane of the wave has a ray that any of 12 consecutive triangles accepts (whatever its
    // distance) -- the most a wave-uniform skip of that group by a box around it could save; counted in `full`
    bool any = false;
    for (int i = 0; i < n; ++i) {
        const TriRec t = load_rec(recs, i);
        unsigned unused = 0;
        test_triangle_from<false>(t, i, ro, rd, h, unused, &any);
        if (i % 12 == 11 || i == n - 1) {
            if (!__builtin_amdgcn_ballot_w64(any)) ++full;
            any = false;
        }
    }
    return h;
#else
    TriRec a0 = load_rec(recs, 0), a1 = load_rec(recs, 1);
    for (int i = 0; i < n; i += 4) {
        const TriRec b0 = load_rec(recs, i + 2), b1 = load_rec(recs, i + 3);
        test_triangle_from<false>(a0, i, ro, rd, h, full);
        test_triangle_from<false>(a1, i + 1, ro, rd, h, full);
        a0 = load_rec(recs, i + 4);
        a1 = load_rec(recs, i + 5);
        test_triangle_from<false>(b0, i + 2, ro, rd, h, full);
        test_triangle_from<false>(b1, i + 3, ro, rd, h, full);
    }
    return h;
#endif
}

// Wave-wide sum / max of a value held by ALL 64 lanes (EXEC must be full), without LDS: two quad permutes and two row
// rotations leave every lane of a 16-lane row with the row's total, then the four row totals are combined through
// scalar registers. The result is wave-uniform.
template <bool MAX>
__device__ __forceinline__ float wave_reduce(float v) {
    auto comb = [](float a, float b) { return MAX ? __builtin_fmaxf(a, b) : a + b; };
    auto dpp = [](float x, int ctrl_tag) {
        const int xi = __float_as_int(x);
        int r;
        switch (ctrl_tag) {
            case 0: r = __builtin_amdgcn_update_dpp(xi, xi, 0xB1, 0xf, 0xf, false); break;   // quad_perm [1,0,3,2]
            case 1: r = __builtin_amdgcn_update_dpp(xi, xi, 0x4E, 0xf, 0xf, false); break;   // quad_perm [2,3,0,1]
            case 2: r = __builtin_amdgcn_update_dpp(xi, xi, 0x124, 0xf, 0xf, false); break;  // row_ror:4
            default: r = __builtin_amdgcn_update_dpp(xi, xi, 0x128, 0xf, 0xf, false); break; // row_ror:8
        }
        return __int_as_float(r);
    };
    v = comb(v, dpp(v, 0));
    v = comb(v, dpp(v, 1));
    v = comb(v, dpp(v, 2));
    v = comb(v, dpp(v, 3));
    const int vi = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
    return comb(comb(r0, r1), comb(r2, r3));
}

struct CullRec { float4 a, b, c; };  // nab.xyz nbc.x | nbc.yz nca.xy | nca.z front.xyz

__device__ __forceinline__ CullRec load_cull(const float4* __restrict__ cull, int k, int n) {
    CullRec r;
    // out-of-range lanes get a record that always rejects (every plane faces away from any axis by more than 1)
    r.a = make_float4(0.f, 0.f, 0.f, 0.f); r.b = r.a; r.c = r.a;
    if (k < n) { r.a = cull[4 * (size_t)k]; r.b = cull[4 * (size_t)k + 1]; r.c = cull[4 * (size_t)k + 2]; }
    return r;
}

// fs:125-174 for a whole wave at once, skipping triangles no ray of the wave can hit. All rays start at the camera,
// so the wave's directions are bounded by a cone (axis = mean direction, opening from the largest chord to it);
// 64 triangles are tested against that cone in parallel, one per lane, using the cull records; the survivors are
// then tested exactly, in ascending index order, like the plain loop -- so the outcome is the plain loop's.
// Must be called by all 64 lanes; `active` marks the lanes that carry a ray. `first` = this lane's cull record of
// chunk 0, loaded once per kernel (for meshes of up to 64 triangles no record is fetched inside the ray loop).
__device__ __forceinline__ Hit intersect_mesh_culled(const PtKernelParams& p, v3 rd, bool active, const CullRec& first, unsigned& tri_tests) {
    Hit h;
    h.t = 1.0e38f; h.id = -1; h.u = 0.f; h.v = 0.f; h.ood = 0.f;
    unsigned full = 0;
    const float w = active ? 1.0f : 0.0f;
    v3 axis = V(wave_reduce<false>(rd.x * w), wave_reduce<false>(rd.y * w), wave_reduce<false>(rd.z * w));
    const float len2 = dot3(axis, axis);
    axis = axis * (len2 > 0.f ? __builtin_amdgcn_rsqf(len2) : 0.f);
    const v3 dv = rd - axis;
    // |d - axis| >= sin(angle) for unit vectors; the slack covers rounding here and in the exact test
    const float spread = __builtin_sqrtf(wave_reduce<true>(active ? dot3(dv, dv) : 0.f)) + 3.0e-4f;
    const float lim = -spread;
    const int lane = threadIdx.x & 63;
    const float4* __restrict__ cull = reinterpret_cast<const float4*>(p.cull);
    const float4* __restrict__ recs = reinterpret_cast<const float4*>(p.tris);
    const int n = p.num_triangles;
    CullRec cr = first;
    for (int base = 0; base < n; base += 64) {
        // the next chunk's records are requested before this chunk's survivors are tested
        CullRec nxt_cr = cr;
        if (base + 64 < n) nxt_cr = load_cull(cull, base + 64 + lane, n);
        const float d0 = axis.x * cr.a.x + (axis.y * cr.a.y + axis.z * cr.a.z);
        const float d1 = axis.x * cr.a.w + (axis.y * cr.b.x + axis.z * cr.b.y);
        const float d2 = axis.x * cr.b.z + (axis.y * cr.b.w + axis.z * cr.c.x);
        const float d3 = axis.x * cr.c.y + (axis.y * cr.c.z + axis.z * cr.c.w);
        const bool keep = (base + lane < n) & !((d0 < lim) | (d1 < lim) | (d2 < lim) | (d3 < lim));
        unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
        tri_tests += (unsigned)__builtin_popcountll(mask);
        if (mask) {
            int i = base + __builtin_ctzll(mask);
            TriRec cur = load_rec(recs, i);
            mask &= mask - 1;
            while (true) {
                const int inext = mask ? base + __builtin_ctzll(mask) : i;
                const TriRec nxt = load_rec(recs, inext);  // requested before the current one is tested
                test_triangle<false>(cur, i, rd, h, full);
                if (!mask) break;
                mask &= mask - 1;
                cur = nxt;
                i = inext;
            }
        }
        cr = nxt_cr;
    }
    return h;
}

// Closest hit through the LBVH, one traversal per WAVE: the stack and the node being visited are wave-uniform (node
// records arrive by scalar loads), each lane tests its own ray against the two child boxes, and a child is entered when
// ANY lane may hit it. Rays of a wave are coherent (an 8x8 pixel tile and its near-mirror bounces), so the union of
// their paths through the tree is short. A lane's box test passes when its ray can reach the padded box no farther than
// its current hit (equal distance included, so a lower-id triangle at the same distance is still found); leaves run
// the exact triangle test with the explicit lowest-id rule. The result equals the all-triangles loop bit for bit.
// MOVING = rays carry their own origin `ro` (boxes are then absolute and the origin is subtracted per lane).
template <bool MOVING>
__device__ __forceinline__ Hit intersect_bvh(const PtKernelParams& p, v3 ro, v3 rd, uint32_t* __restrict__ stack, unsigned& node_visits,
                                             unsigned& tri_tests) {
    Hit h;
    h.t = 1.0e38f; h.id = -1; h.u = 0.f; h.v = 0.f; h.ood = 0.f;
    unsigned full = 0;
    // Reciprocal direction for the slab tests only (culling arithmetic is not part of the reference's numerics; it
    // only has to be conservative, which the padded boxes and the slack on `limit` guarantee). A zero component is
    // replaced by a tiny one of the same sign: the slab test then degenerates to "origin inside the slab".
    const float tiny = 1.0e-20f;  // reciprocals stay below 1e20, so centre x reciprocal cannot overflow for any sane coordinate
    const float ix = __builtin_amdgcn_rcpf(__builtin_fabsf(rd.x) > tiny ? rd.x : __builtin_copysignf(tiny, rd.x));
    const float iy = __builtin_amdgcn_rcpf(__builtin_fabsf(rd.y) > tiny ? rd.y : __builtin_copysignf(tiny, rd.y));
    const float iz = __builtin_amdgcn_rcpf(__builtin_fabsf(rd.z) > tiny ? rd.z : __builtin_copysignf(tiny, rd.z));
    const float ax = __builtin_fabsf(ix), ay = __builtin_fabsf(iy), az = __builtin_fabsf(iz);
    // origin x reciprocal, subtracted inside the fused multiply-add of each slab
    const float ox = MOVING ? ro.x * ix : 0.f, oy = MOVING ? ro.y * iy : 0.f, oz = MOVING ? ro.z * iz : 0.f;
    const float4* __restrict__ nodes = reinterpret_cast<const float4*>(MOVING ? p.bvh_nodes_abs : p.bvh_nodes);
    const float4* __restrict__ recs = reinterpret_cast<const float4*>(MOVING ? p.bvh_tris_abs : p.bvh_tris);
    uint32_t ref = 0;  // root = internal node 0
    int sp = 0;
    for (;;) {
        if (ref & 0x80000000u) {
            // A leaf is up to PT_BVH_LEAF_MAX consecutive records; all of them are requested at once (the array is
            // padded, and a record past the leaf is simply not tested), so a leaf costs one memory latency, not `count`.
            const int first = (int)((ref & 0x7fffffffu) >> 3), count = (int)(ref & 7u) + 1;
            const TriRec r0 = load_rec(recs, first), r1 = load_rec(recs, first + 1), r2 = load_rec(recs, first + 2),
                         r3 = load_rec(recs, first + 3);
            if (MOVING) {
                test_triangle_from<true>(r0, (int)__float_as_uint(r0.d.w), ro, rd, h, full);
                if (count > 1) test_triangle_from<true>(r1, (int)__float_as_uint(r1.d.w), ro, rd, h, full);
                if (count > 2) test_triangle_from<true>(r2, (int)__float_as_uint(r2.d.w), ro, rd, h, full);
                if (count > 3) test_triangle_from<true>(r3, (int)__float_as_uint(r3.d.w), ro, rd, h, full);
            } else {
                test_triangle<true>(r0, (int)__float_as_uint(r0.d.w), rd, h, full);
                if (count > 1) test_triangle<true>(r1, (int)__float_as_uint(r1.d.w), rd, h, full);
                if (count > 2) test_triangle<true>(r2, (int)__float_as_uint(r2.d.w), rd, h, full);
                if (count > 3) test_triangle<true>(r3, (int)__float_as_uint(r3.d.w), rd, h, full);
            }
            tri_tests += (unsigned)count;
        } else {
            ++node_visits;
            const ScalarView kn = scalar_view(nodes + 4 * (size_t)ref);
            const float4 n0 = kn[0], n1 = kn[1], n2 = kn[2], n3 = kn[3];
            const float limit = h.t * 1.0001f;
            // left box: centre n0.xyz, half extents (n0.w, n1.x, n1.y); right box: centre (n1.z, n1.w, n2.x), half n2.yzw.
            // Per axis: entry = c*inv - |h*inv|, exit = c*inv + |h*inv| (minus origin*inv for rays with their own origin).
            // This arithmetic only decides what is skipped; fused multiply-adds are fine here.
            float m, t;
            m = n0.w * ax; t = MOVING ? __builtin_fmaf(n0.x, ix, -ox) : n0.x * ix;
            float lnear = MOVING ? t - m : __builtin_fmaf(n0.x, ix, -m), lfar = MOVING ? t + m : __builtin_fmaf(n0.x, ix, m);
            m = n1.x * ay; t = MOVING ? __builtin_fmaf(n0.y, iy, -oy) : n0.y * iy;
            const float lny = MOVING ? t - m : __builtin_fmaf(n0.y, iy, -m), lfy = MOVING ? t + m : __builtin_fmaf(n0.y, iy, m);
            m = n1.y * az; t = MOVING ? __builtin_fmaf(n0.z, iz, -oz) : n0.z * iz;
            const float lnz = MOVING ? t - m : __builtin_fmaf(n0.z, iz, -m), lfz = MOVING ? t + m : __builtin_fmaf(n0.z, iz, m);
            lnear = __builtin_fmaxf(__builtin_fmaxf(lnear, lny), __builtin_fmaxf(lnz, 0.0f));
            lfar = __builtin_fminf(__builtin_fminf(lfar, lfy), __builtin_fminf(lfz, limit));
            m = n2.y * ax; t = MOVING ? __builtin_fmaf(n1.z, ix, -ox) : n1.z * ix;
            float rnear = MOVING ? t - m : __builtin_fmaf(n1.z, ix, -m), rfar = MOVING ? t + m : __builtin_fmaf(n1.z, ix, m);
            m = n2.z * ay; t = MOVING ? __builtin_fmaf(n1.w, iy, -oy) : n1.w * iy;
            const float rny = MOVING ? t - m : __builtin_fmaf(n1.w, iy, -m), rfy = MOVING ? t + m : __builtin_fmaf(n1.w, iy, m);
            m = n2.w * az; t = MOVING ? __builtin_fmaf(n2.x, iz, -oz) : n2.x * iz;
            const float rnz = MOVING ? t - m : __builtin_fmaf(n2.x, iz, -m), rfz = MOVING ? t + m : __builtin_fmaf(n2.x, iz, m);
            rnear = __builtin_fmaxf(__builtin_fmaxf(rnear, rny), __builtin_fmaxf(rnz, 0.0f));
            rfar = __builtin_fminf(__builtin_fminf(rfar, rfy), __builtin_fminf(rfz, limit));
            // entry clamped to 0 and exit clamped to the current hit distance (with slack, equal distances included)
            const bool hl = lnear <= lfar;
            const bool hr = rnear <= rfar;
            const unsigned long long ml = __builtin_amdgcn_ballot_w64(hl), mr = __builtin_amdgcn_ballot_w64(hr);
            const uint32_t lref = __float_as_uint(n3.x), rref = __float_as_uint(n3.y);
            if (ml && mr) {
                // both: enter the child that is nearer for most lanes first, keep the other on the stack
                const unsigned long long lf = __builtin_amdgcn_ballot_w64(hl & (!hr | (lnear <= rnear)));
                const bool left_first = 2 * __builtin_popcountll(lf) >= __builtin_popcountll(ml | mr);
                stack[sp] = left_first ? rref : lref;  // every active lane stores the same word
                ++sp;
                ref = left_first ? lref : rref;
                continue;
            }
            if (ml | mr) {
                ref = ml ? lref : rref;
                continue;
            }
        }
        if (sp == 0) break;
        --sp;
        ref = __builtin_amdgcn_readfirstlane(stack[sp]);
    }
    return h;
}

// Closest hit through the LBVH with one traversal per LANE, for rays that have little in common (diffuse bounces from
// scattered hit points): every lane keeps its own stack (LDS, PT_LANE_STACK entries, lane-interleaved) and fetches its own
// node and triangle records with vector loads, so a lane only visits what its own ray needs instead of the union over
// the wave. Same padded boxes, same slab arithmetic, same exact triangle test with the explicit lowest-id rule as the
// per-wave walk. A lane whose stack would overflow (very deep hierarchy) reports it; the caller repeats that lane's query
// with the per-wave walk, which has room for any hierarchy.
__device__ __forceinline__ Hit intersect_bvh_lanes(const PtKernelParams& p, v3 ro, v3 rd, uint32_t* __restrict__ lstack, bool& overflow,
                                                   unsigned& steps) {
    Hit h;
    h.t = 1.0e38f; h.id = -1; h.u = 0.f; h.v = 0.f; h.ood = 0.f;
    overflow = false;
    const float tiny = 1.0e-20f;
    const float ix = __builtin_amdgcn_rcpf(__builtin_fabsf(rd.x) > tiny ? rd.x : __builtin_copysignf(tiny, rd.x));
    const float iy = __builtin_amdgcn_rcpf(__builtin_fabsf(rd.y) > tiny ? rd.y : __builtin_copysignf(tiny, rd.y));
    const float iz = __builtin_amdgcn_rcpf(__builtin_fabsf(rd.z) > tiny ? rd.z : __builtin_copysignf(tiny, rd.z));
    const float ax = __builtin_fabsf(ix), ay = __builtin_fabsf(iy), az = __builtin_fabsf(iz);
    const float ox = ro.x * ix, oy = ro.y * iy, oz = ro.z * iz;
    const float4* __restrict__ nodes = reinterpret_cast<const float4*>(p.bvh_nodes_abs);
    const float4* __restrict__ recs = reinterpret_cast<const float4*>(p.bvh_tris_abs);
    uint32_t* __restrict__ my = lstack + (threadIdx.x & 63);
    uint32_t ref = 0;  // root = internal node 0
    int sp = 0;
    for (;;) {
        // descend until this lane holds a leaf (or ran out of nodes)
        while (!(ref & 0x80000000u)) {
            ++steps;
            const float4 n0 = nodes[4 * (size_t)ref], n1 = nodes[4 * (size_t)ref + 1], n2 = nodes[4 * (size_t)ref + 2], n3 = nodes[4 * (size_t)ref + 3];
            const float limit = h.t * 1.0001f;
            float m, t;
            m = n0.w * ax; t = __builtin_fmaf(n0.x, ix, -ox); float lnear = t - m, lfar = t + m;
            m = n1.x * ay; t = __builtin_fmaf(n0.y, iy, -oy); const float lny = t - m, lfy = t + m;
            m = n1.y * az; t = __builtin_fmaf(n0.z, iz, -oz); const float lnz = t - m, lfz = t + m;
            lnear = __builtin_fmaxf(__builtin_fmaxf(lnear, lny), __builtin_fmaxf(lnz, 0.0f));
            lfar = __builtin_fminf(__builtin_fminf(lfar, lfy), __builtin_fminf(lfz, limit));
            m = n2.y * ax; t = __builtin_fmaf(n1.z, ix, -ox); float rnear = t - m, rfar = t + m;
            m = n2.z * ay; t = __builtin_fmaf(n1.w, iy, -oy); const float rny = t - m, rfy = t + m;
            m = n2.w * az; t = __builtin_fmaf(n2.x, iz, -oz); const float rnz = t - m, rfz = t + m;
            rnear = __builtin_fmaxf(__builtin_fmaxf(rnear, rny), __builtin_fmaxf(rnz, 0.0f));
            rfar = __builtin_fminf(__builtin_fminf(rfar, rfy), __builtin_fminf(rfz, limit));
            const bool hl = lnear <= lfar, hr = rnear <= rfar;
            const uint32_t lref = __float_as_uint(n3.x), rref = __float_as_uint(n3.y);
            if (hl && hr) {
                const bool left_first = lnear <= rnear;
                if (sp >= p.lane_stack_limit) { overflow = true; return h; }
                my[sp * 64] = left_first ? rref : lref;
                ++sp;
                ref = left_first ? lref : rref;
            } else if (hl || hr) {
                ref = hl ? lref : rref;
            } else {
                if (sp == 0) return h;
                --sp;
                ref = my[sp * 64];
            }
        }
        // leaf: up to PT_BVH_LEAF_MAX consecutive records
        const int first = (int)((ref & 0x7fffffffu) >> 3), count = (int)(ref & 7u) + 1;
        for (int k = 0; k < count; ++k) {
            ++steps;
            const float4 ra = recs[4 * (size_t)(first + k)], rb = recs[4 * (size_t)(first + k) + 1], rc = recs[4 * (size_t)(first + k) + 2],
                         rdd = recs[4 * (size_t)(first + k) + 3];
            // fs:136-168 as written: record = a.xyz e1.x | e1.yz e2.xy | e2.z - - - | - - - id
            const v3 a = V(ra.x, ra.y, ra.z), e1 = V(ra.w, rb.x, rb.y), e2 = V(rb.z, rb.w, rc.x);
            const v3 q = cross3(rd, e2);
            const float det = dot3(e1, q);
            const v3 ap = ro - a;
            const v3 r = cross3(ap, e1);
            const float u = dot3(ap, q);
            const float v = dot3(rd, r);
            if ((det >= 0.0001f) & (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= det)) {
                const float ood = ieee_rcp(det);
                const float tt = dot3(e2, r) * ood;
                const int id = (int)__float_as_uint(rdd.w);
                if (tt > 0.0f && (tt < h.t || (tt == h.t && id < h.id))) {
                    h.t = tt; h.id = id; h.u = u; h.v = v; h.ood = ood;
                }
            }
        }
        if (sp == 0) return h;
        --sp;
        ref = my[sp * 64];
    }
}

// The cube-map face of a direction, by the hardware's cube instructions: id = face number 0..5 (+x, -x, +y, -y, +z, -z: the
// component of largest magnitude and its sign), ma = twice that component, (sc, tc) = the other two with the signs of the
// cube-map convention. cube_coords_expected() states the selections in C++; pt_arith_selftest_kernel compares the two.
struct CubeCoords { float id, sc, tc, ma; };
__device__ __forceinline__ CubeCoords cube_coords(v3 d) {
    CubeCoords c;
    c.id = __builtin_amdgcn_cubeid(d.x, d.y, d.z);
    c.sc = __builtin_amdgcn_cubesc(d.x, d.y, d.z);
    c.tc = __builtin_amdgcn_cubetc(d.x, d.y, d.z);
    c.ma = __builtin_amdgcn_cubema(d.x, d.y, d.z);
    return c;
}
__device__ __forceinline__ CubeCoords cube_coords_expected(v3 d) {
    const float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
    CubeCoords c;
    if (az >= ax && az >= ay) { c.id = d.z < 0.f ? 5.f : 4.f; c.sc = d.z < 0.f ? -d.x : d.x; c.tc = -d.y; c.ma = 2.0f * d.z; }
    else if (ay >= ax)        { c.id = d.y < 0.f ? 3.f : 2.f; c.sc = d.x; c.tc = d.y < 0.f ? -d.z : d.z; c.ma = 2.0f * d.y; }
    else                      { c.id = d.x < 0.f ? 1.f : 0.f; c.sc = d.x < 0.f ? d.z : -d.z; c.tc = -d.y; c.ma = 2.0f * d.x; }
    return c;
}

// A word of one of the direction grid's tables. The byte offset is formed in 32 bits and added to the (wave-uniform) base by
// the load itself: the tables stay below 4 GiB (6 * 2048^2 cell words; at most 2^30 list words).
__device__ __forceinline__ uint32_t grid_word(const uint32_t* __restrict__ table, uint32_t index) {
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(table) + (index << 2));
}

// Direction-grid query (see pt_grid.hip): the cell of the ray's direction lists the triangles it can hit at all. Every
// lane walks its own list (lanes of a wave look in different cells), records come through per-lane vector loads; the
// index of the next entry is fetched while the current triangle is tested. Same exact test, explicit lowest-index rule.
// `gn` <- the accepted triangle's geometric normal, which travels in its record.
// `want`: lanes without a query pass false and get "nothing hit" (callers inside a branch pass true).
__device__ __forceinline__ Hit intersect_grid(const PtKernelParams& p, v3 rd, bool want, unsigned& tests, v3& gn, bool use_occ) {
    // (One exit, and the hit record is touched by the list loop only: with early returns the compiler set the record's
    // registers to "nothing hit" again at every one of them, four times per query.)
    // Face and face coordinates from the cube-map instructions (exact selections and sign changes, no rounding): `id` = 2 * axis +
    // (negative ? 1 : 0), `ma` = twice the major component, (sc, tc) = the other two components in the orientation that
    // pt_grid.hip's to_face() lists per face. Which face a tie between two magnitudes goes to does not matter: the build widens
    // every face's pyramid.
    const CubeCoords cc = cube_coords(rd);
    const float fw2 = __builtin_fabsf(cc.ma);
    // One word per cell: empty, the cell's only triangle (then the record is the very next fetch), or the position of its
    // list, whose last entry carries the top bit.
    uint32_t entry = PT_GRID_EMPTY;
    if (want && fw2 > 0.f) {  // a zero or NaN direction hits nothing
        const float iw2 = __builtin_amdgcn_rcpf(fw2);  // 1 / (2 |major|): sc * iw2 lies in [-0.5, 0.5]
        const int res = p.grid_res;
        const float fres = (float)res;
        const uint32_t face = (uint32_t)cc.id;
        int iu = (int)__builtin_floorf((cc.sc * iw2 + 0.5f) * fres), iv = (int)__builtin_floorf((cc.tc * iw2 + 0.5f) * fres);
        // clamp to [0, res - 1] (res >= 1): the median of the three, one instruction each
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(iu) : "v"(iu), "s"(res - 1));
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(iv) : "v"(iv), "s"(res - 1));
        // < 6 * 2048^2; both factors of either product are below 2^24, so the full-rate 24-bit multiply is exact
        const uint32_t cell = __umul24(__umul24(face, (uint32_t)res) + (uint32_t)iv, (uint32_t)res) + (uint32_t)iu;
        // Open scenes: one bit per 4x4 block of cells says whether anything lies that way at all (read from a table small enough
        // to stay next to the compute units; the cell word of a fine grid comes from the fabric).
        bool look = true;
        if (use_occ) {  // (wave-uniform)
            const uint32_t edge = (uint32_t)PT_GRID_OCC_EDGE(res);
            const uint32_t blk = __umul24(__umul24(face, edge) + ((uint32_t)iv >> 2), edge) + ((uint32_t)iu >> 2);
            look = ((p.grid_occ[blk >> 5] >> (blk & 31u)) & 1u) != 0u;
        }
        if (look) entry = grid_word(p.grid_cells, cell);
    }
    bool more = entry != PT_GRID_EMPTY;
    uint32_t pos = 0u;
    if (more) {
        if (entry & 0x80000000u) {
            pos = entry & 0x7fffffffu;
            entry = grid_word(p.grid_list, pos);
        } else {
            entry |= 0x80000000u;
        }
    }
    // "Nothing hit" is distance 1e38 and index -1, set once here (the empty statement keeps the compiler from setting the two
    // again on every path around the loop). The other three words and `gn` are read by callers only after a hit has written
    // them: they start as whatever their registers hold.
    Hit h;
    h.t = 1.0e38f; h.id = -1;
    asm volatile("" : "+v"(h.t), "+v"(h.id));
    asm volatile("" : "=v"(h.u), "=v"(h.v), "=v"(h.ood));
    while (more) {
        const uint32_t tri = entry & 0x7fffffffu;
        const bool last = (entry & 0x80000000u) != 0u;
        if (!last) entry = grid_word(p.grid_list, ++pos);  // fetched while this triangle is tested
        // (byte offset in 32 bits, see grid_word: pt_capi.hip keeps direction grids to meshes of at most 2^26 triangles)
        const float4* __restrict__ rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.grid_tris) + (tri << 6));
        TriRec t;
        t.a = rec[0]; t.b = rec[1]; t.c = rec[2]; t.d = rec[3];
        const v3 e1 = V(t.a.x, t.a.y, t.a.z);
        const v3 e2 = V(t.a.w, t.b.x, t.b.y);
        const v3 q = cross3(rd, e2);
        const float det = dot3(e1, q);
        const v3 ap = V(t.b.z, t.b.w, t.c.x);
        const v3 r = V(t.c.y, t.c.z, t.c.w);
        const float u = dot3(ap, q);
        const float v = dot3(rd, r);
        // the record's last quarter (e2r, normal) is only used by a hit; without this the compiler fetches it inside the
        // two branches below, two more round trips in a row -- here it arrives together with the other three quarters
        asm volatile("" : "+v"(t.d.x), "+v"(t.d.y), "+v"(t.d.z), "+v"(t.d.w));
        if ((det >= 0.0001f) & (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= det)) {
            const float ood = ieee_rcp(det);
            const float tt = t.d.x * ood;
            if (tt > 0.0f && (tt < h.t || (tt == h.t && (int)tri < h.id))) {
                h.t = tt; h.id = (int)tri; h.u = u; h.v = v; h.ood = ood;
                gn = V(t.d.y, t.d.z, t.d.w);
            }
        }
        ++tests;
        more = !last;
    }
    return h;
}

__device__ __forceinline__ v3 mesh_normal(const PtKernelParams& p, const Hit& h);
// The direction-grid query in two pieces, for the form of the render loop in which a lane tests ONE triangle of its cell's list
// per trip and carries the rest of the list into its next trip (instead of the wave running its list loop as often as its
// longest list). grid_lookup: the first entry of the cell's list (top bit = last) and its position; false when the cell is empty.
__device__ __forceinline__ bool grid_lookup(const PtKernelParams& p, v3 rd, bool want, bool use_occ, uint32_t& entry, uint32_t& pos) {
    const CubeCoords cc = cube_coords(rd);
    const float fw2 = __builtin_fabsf(cc.ma);
    entry = PT_GRID_EMPTY;
    if (want && fw2 > 0.f) {
        const float iw2 = __builtin_amdgcn_rcpf(fw2);
        const int res = p.grid_res;
        const float fres = (float)res;
        const uint32_t face = (uint32_t)cc.id;
        int iu = (int)__builtin_floorf((cc.sc * iw2 + 0.5f) * fres), iv = (int)__builtin_floorf((cc.tc * iw2 + 0.5f) * fres);
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(iu) : "v"(iu), "s"(res - 1));
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(iv) : "v"(iv), "s"(res - 1));
        const uint32_t cell = __umul24(__umul24(face, (uint32_t)res) + (uint32_t)iv, (uint32_t)res) + (uint32_t)iu;
        bool look = true;
        if (use_occ) {
            const uint32_t edge = (uint32_t)PT_GRID_OCC_EDGE(res);
            const uint32_t blk = __umul24(__umul24(face, edge) + ((uint32_t)iv >> 2), edge) + ((uint32_t)iu >> 2);
            look = ((p.grid_occ[blk >> 5] >> (blk & 31u)) & 1u) != 0u;
        }
        if (look) entry = grid_word(p.grid_cells, cell);
    }
    const bool more = entry != PT_GRID_EMPTY;
    pos = 0u;
    if (more) {
        if (entry & 0x80000000u) {
            pos = entry & 0x7fffffffu;
            entry = grid_word(p.grid_list, pos);
        } else {
            entry |= 0x80000000u;
        }
    }
    return more;
}
// grid_test_one: the exact test of the list's current triangle against the best hit so far (t, id, normal); returns whether the
// list goes on (entry / pos then name the next triangle).
__device__ __forceinline__ bool grid_test_one(const PtKernelParams& p, v3 rd, uint32_t& entry, uint32_t& pos, float& best_t, int& best_id, v3& best_n,
                                              float* __restrict__ uvo, unsigned& tests) {
    const uint32_t tri = entry & 0x7fffffffu;
    const bool last = (entry & 0x80000000u) != 0u;
    if (!last) entry = grid_word(p.grid_list, ++pos);
    const float4* __restrict__ rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.grid_tris) + (tri << 6));
    TriRec t;
    t.a = rec[0]; t.b = rec[1]; t.c = rec[2]; t.d = rec[3];
    const v3 e1 = V(t.a.x, t.a.y, t.a.z);
    const v3 e2 = V(t.a.w, t.b.x, t.b.y);
    const v3 q = cross3(rd, e2);
    const float det = dot3(e1, q);
    const v3 ap = V(t.b.z, t.b.w, t.c.x);
    const v3 r = V(t.c.y, t.c.z, t.c.w);
    const float u = dot3(ap, q);
    const float v = dot3(rd, r);
    asm volatile("" : "+v"(t.d.x), "+v"(t.d.y), "+v"(t.d.z), "+v"(t.d.w));
    if ((det >= 0.0001f) & (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= det)) {
        const float ood = ieee_rcp(det);
        const float tt = t.d.x * ood;
        if (tt > 0.0f && (tt < best_t || (tt == best_t && (int)tri < best_id))) {
            best_t = tt; best_id = (int)tri;
            best_n = V(t.d.y, t.d.z, t.d.w);
            // what the interpolated normal needs, should this stay the hit (formed when the query is answered): parked in LDS
            if (p.normal_mode == 0) { uvo[0] = u; uvo[64] = v; uvo[128] = ood; }
        }
    }
    ++tests;
    return !last;
}

// fs:83-121. The ray origin is the camera position for every ray, so the terms of the sphere test that do not involve the
// direction are taken from a per-sphere record made once per camera position (pt_prepare_spheres_kernel: to = ro - centre,
// C = dot(to, to) - r*r, 1/r; the same fp32 operations the shader performs per ray). The normal (fs:111) is formed once,
// for the sphere that ends up closest. Returns it through `n`.
__device__ __forceinline__ Hit intersect_world(const float* __restrict__ sphere_recs, v3 ro, v3 rd, v3& n) {
    Hit h;
    h.t = 1.0e38f; h.id = -1; h.u = 0.f; h.v = 0.f; h.ood = 0.f;
    const ScalarView recs = scalar_view(reinterpret_cast<const float4*>(sphere_recs));
#pragma unroll
    for (int obj = 0; obj < 10; ++obj) {
        const float4 r0 = recs[2 * obj];  // to.xyz, C
        const float B = -dot3(V(r0.x, r0.y, r0.z), rd);
        const float disc = B * B - r0.w;
        if (disc > 0.0f) {
            const float root = ieee_sqrt(disc);
            float temp = B - root;
            temp = temp > 0.0f ? temp : (B + root);
            if (temp > 0.0f && temp < h.t) { h.t = temp; h.id = obj; }
        }
    }
    if (h.id >= 0) {
        const float4 r1 = reinterpret_cast<const float4*>(sphere_recs)[2 * h.id + 1];  // centre.xyz, 1/radius
        const v3 p = ro + rd * h.t;
        n = (p - V(r1.x, r1.y, r1.z)) * r1.w;
    }
    return h;
}

// Shading normal of a mesh hit, computed once for the accepted triangle (fs:164-166).
__device__ __forceinline__ v3 mesh_normal(const PtKernelParams& p, const Hit& h) {
    if (p.normal_mode == 0) {
        const float* __restrict__ nn = p.vnormals + (size_t)h.id * 9;
        const float ux = h.u * h.ood, uy = h.v * h.ood, uz = (1.0f - ux) - uy;
        const v3 na = V(nn[0], nn[1], nn[2]), nb = V(nn[3], nn[4], nn[5]), nc = V(nn[6], nn[7], nn[8]);
        return normalize3((na * uz + nb * ux) + nc * uy);
    }
    const float* __restrict__ g = p.gnormals + (size_t)h.id * 3;
    return V(g[0], g[1], g[2]);
}

// fs:224-361, the lines that are live: modified-Phong lobe about the mirror direction, specularity 0.94,
// exponent 1024, albedo 0.5. Replaces rd and scales the (grey) path throughput; `ro` is never moved.
__device__ __forceinline__ void shade(Rng& rng, float& energy, v3& rd, v3 N) {
    const float specularity = 0.94f;
    const float alpha = 1024.0f;
    const v3 R = reflect3(rd, N);
    const float r1 = rng.rand01();
    // GetTangentSpace(R), fs:44-55
    v3 helper = V(1.0f, 0.0f, 0.0f);
    if (__builtin_fabsf(R.x) > 0.99f) helper = V(0.0f, 0.0f, 1.0f);
    const v3 tangent = normalize3(cross3(R, helper));
    const v3 binormal = normalize3(cross3(R, tangent));
    // randomHemisphere(r1, alpha), fs:67-76
    const float cos_t = lp_pow_unit(r1, 1.0f / (alpha + 1.0f));
    const float sin_t = ieee_sqrt(__builtin_fmaxf(0.0f, 1.0f - cos_t * cos_t));
    const float phi = (rng.rand01() * 2.0f) * 3.141592653589f;
    const v3 hemi = V(lp_sincos<true>(phi) * sin_t, cos_t, lp_sincos<false>(phi) * sin_t);
    // mat3(tangent, R, binormal) * hemi = (c0*x + c1*y) + c2*z
    rd = (tangent * hemi.x + R * hemi.y) + binormal * hemi.z;
    const float diffuse = __builtin_fminf(1.0f - specularity, 0.5f);
    const float f = (alpha + 2.0f) / (alpha + 1.0f);
    // the exponent lies in [-8.66, 0] whatever dot(N, rd) is (clamp01 maps a NaN to 0): neither clamp of exp2 can bind
    const float fresnel = specularity + (1.0f - specularity) * lp_exp2<false, false>(-8.656170f * clamp01(dot3(N, rd)));
    const float specular = (fresnel * specularity) * clamp01(dot3(N, rd) * f);
    energy = energy * (diffuse + specular);
}

// Shade() of the build-defined diffuse/emissive mode (the reference's live Shade ignores materials; its Lambert and
// emissive code is only present as comments, fs:269-279, 317-358). material = (r, g, b, e): e > 0 is an emitter of
// radiance rgb*e that ends the path; otherwise a Lambert surface of albedo rgb: cosine-weighted direction about N from
// the shader's own GetTangentSpace / randomHemisphere(r1, 1.0) (fs:44-76) with its own rand01(), and the origin moves
// to the hit point as in fs:274. Pinned against llvmpipe running the shader with exactly these statements edited in.
// `unit_form` (wave-uniform: the shading mode) selects the Lambert bounce the reference carries at fs:272 instead,
// rd = normalize(N + randomUnitVector()), with randomUnitVector() as at fs:57-65.
__device__ __forceinline__ void shade_materials(Rng& rng, const float* __restrict__ materials, int mat, v3& energy, v3& emitted, v3& ro,
                                                v3& rd, v3 P, v3 N, bool unit_form) {
    const float4 m = reinterpret_cast<const float4*>(materials)[mat];
    if (m.w > 0.0f) {
        emitted = V(energy.x * (m.x * m.w), energy.y * (m.y * m.w), energy.z * (m.z * m.w));
        energy = V(0.f, 0.f, 0.f);
        return;
    }
    if (unit_form) {
        // fs:57-65: z uniform in [-1, 1), then a point on the circle of that height
        const float z = rng.rand01() * 2.0f - 1.0f;
        const float r = ieee_sqrt(1.0f - z * z);
        const float a = (rng.rand01() * 2.0f) * 3.141592653589f;
        rd = normalize3(N + V(lp_sincos<true>(a) * r, lp_sincos<false>(a) * r, z));
    } else {
        const float r1 = rng.rand01();
        v3 helper = V(1.0f, 0.0f, 0.0f);
        if (__builtin_fabsf(N.x) > 0.99f) helper = V(0.0f, 0.0f, 1.0f);
        const v3 tangent = normalize3(cross3(N, helper));
        const v3 binormal = normalize3(cross3(N, tangent));
        const float cos_t = lp_pow_unit(r1, 1.0f / (1.0f + 1.0f));
        const float sin_t = ieee_sqrt(__builtin_fmaxf(0.0f, 1.0f - cos_t * cos_t));
        const float phi = (rng.rand01() * 2.0f) * 3.141592653589f;
        const v3 hemi = V(lp_sincos<true>(phi) * sin_t, cos_t, lp_sincos<false>(phi) * sin_t);
        rd = (tangent * hemi.x + N * hemi.y) + binormal * hemi.z;
    }
    ro = P + rd * 0.0001f;
    energy = V(energy.x * m.x, energy.y * m.y, energy.z * m.z);
}

__device__ __forceinline__ bool pixel_of_lane(const PtKernelParams& p, int tile, int& x, int& y) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (PT_TILE_WAVES == 1 && p.pixel_list) {  // sparse view: the tile is 64 entries of the list of pixels that see the mesh
        const uint32_t e = p.pixel_list[(size_t)tile * 64 + lane];
        x = (int)(e & 0xffffu);
        y = (int)(e >> 16);
        return e != PT_NO_PIXEL;
    }
    // a wave covers an 8x8 pixel block; the workgroup's PT_TILE_WAVES waves tile a PT_TILE_DIM square
    const int tiles_x = PT_TILES_ACROSS(p.win_w);
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int lx = tx * PT_TILE_DIM + (wave & 1) * 8 + (lane & 7);
    const int ly = ty * PT_TILE_DIM + (wave >> 1) * 8 + (lane >> 3);
    x = p.win_x0 + lx;
    y = p.win_y0 + ly;
    bool mine = lx < p.win_w && ly < p.win_h;
    if (p.row_count > 1) mine = mine && (y / p.row_block) % p.row_count == p.row_index;  // image-tile partition by interleaved row blocks
    return mine;
}

// fs:381-383
__device__ __forceinline__ v3 primary_direction(const PtKernelParams& p, float fcx, float fcy) {
    const float ijx = fcx / (float)p.width, ijy = fcy / (float)p.height;
    const v3 ro = V(p.ro[0], p.ro[1], p.ro[2]);
    const v3 llc = V(p.llc[0], p.llc[1], p.llc[2]);
    const v3 hor = V(p.hor[0], p.hor[1], p.hor[2]);
    const v3 ver = V(p.ver[0], p.ver[1], p.ver[2]);
    return normalize3(((llc + hor * ijx) + ver * ijy) - ro);
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ unsigned __reduce_max_trips(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ void park(float2* e, float, float, float, float energy, float rdy) { *e = make_float2(energy, rdy); }
__device__ __forceinline__ void park(float4* e, float sr, float sg, float sb, float, float) { *e = make_float4(sr, sg, sb, 0.f); }

// Which entry of the work list a block takes. Blocks are handed to the eight XCDs (each with its own L2) round-robin by block index.
__device__ __forceinline__ bool work_item_index(const PtKernelParams& p, unsigned& index) {
    const unsigned n = *(const kuint32*)p.sched_counts;
    index = blockIdx.x;
    return index < n;
}

// KIND: 0 sphere scene, 1 all-triangles loop, 2 LBVH, 3 culled loop, 4 / 7 direction grid, 5 per-lane walk in step. MATERIALS: false = the reference's shading (Phong
// lobe, grey throughput, origin fixed at the camera), true = the diffuse/emissive mode (coloured throughput, emitted
// radiance, origin moves to each hit point, so the ray-origin-specific triangle records cannot be used).
template <int KIND, bool MATERIALS>
__global__ __launch_bounds__(PT_TILE_THREADS) void pt_render_kernel(const PtKernelParams p) {
    constexpr bool GRID = KIND == 4 || KIND == 7;   // direction grid: 4 = the wave runs its list loop, 7 = list remainders carried into the next trip
    __shared__ uint32_t s_stack[(KIND == 2 || KIND == 5) ? PT_TILE_WAVES : 1][(KIND == 2 || KIND == 5) ? PT_BVH_STACK : 1];
    __shared__ uint32_t s_lane_stack[KIND == 5 ? PT_TILE_WAVES : 1][KIND == 5 ? PT_LANE_STACK * 64 : 1];
    // Which tile and which of its samples this block renders. Blocks start in blockIdx order and the work list is sorted
    // most expensive first, so the long items do not start last; a tile that would be a long serial chain of samples on
    // one wave is split over several blocks (their samples are folded in order afterwards). Any list that covers every
    // (tile, sample) once gives the same image.
    int x, y;
    int tile = (int)blockIdx.x, sample_begin = 0, sample_count = p.num_samples, scratch_slot = -1;
    if (p.items) {
        unsigned index;
        if (!work_item_index(p, index)) return;  // the grid is sized for the largest possible list
        const kuint32* item = (const kuint32*)(p.items + index);
        if (item[0] == PT_NO_TILE) return;  // a position past the end of its group's part of the list
        tile = (int)item[0]; sample_begin = (int)item[1]; sample_count = (int)item[2]; scratch_slot = (int)item[3];
    }
    const bool in_window = pixel_of_lane(p, tile, x, y);
    const size_t pix = in_window ? ((size_t)y * p.width + x) : 0;
    const v3 ro0 = V(p.ro[0], p.ro[1], p.ro[2]);

    Rng rng;
    rng.fcx = (float)x + 0.5f;
    rng.fcy = (float)y + 0.5f;
    const v3 rd0 = primary_direction(p, rng.fcx, rng.fcy);

    float4 prev = make_float4(0.f, 0.f, 0.f, 1.f);
    if (in_window && scratch_slot < 0) prev = reinterpret_cast<const float4*>(p.src)[pix];
    float mr = prev.x, mg = prev.y, mb = prev.z;
    // what a parked sample is: the reference shading's three channels are functions of the throughput and the final
    // direction's y (fs:408-410), so those two are kept (8 bytes) and pt_fold_samples_kernel forms the channels with the same
    // operations; the materials mode keeps the channels themselves (16 bytes)
    using Parked = typename std::conditional<MATERIALS, float4, float2>::type;
    Parked* __restrict__ scratch = nullptr;
    if (scratch_slot >= 0)
        scratch = reinterpret_cast<Parked*>(p.sample_scratch) + ((size_t)scratch_slot * p.num_samples + sample_begin) * PT_TILE_THREADS + threadIdx.x;

    int remaining = in_window ? sample_count : 0;
    int sample = p.first_sample + sample_begin * p.sample_stride;
    // Closest-hit queries in the reference's sense (one per bounce-loop iteration, PathTracer.cpp:947) are not counted one by
    // one: they are the queries this lane evaluated plus the camera-ray queries answered by a reused result, one per sample
    // from the second on (and the first too when it came from an earlier dispatch) -- formed once, after the loop.
    unsigned traced = 0;  // queries this lane actually evaluated
    bool first_from_cache = false;
    unsigned trips = 0;   // queries evaluated by the WAVE: trips x 64 lane slots were spent for `traced` useful ones
    unsigned node_visits = 0, tri_tests = 0;  // per WAVE (the BVH walk is wave-uniform): records fetched
    unsigned lane_tests = 0;                  // per LANE (direction grid): triangles tested
    unsigned loop_full = 0;                   // per WAVE (all-triangles loop): tests that went past the wave-uniform back-face skip
    unsigned shaded = 0;                      // per LANE: hits shaded

    // direction grid: whether queries look at the block bits first (they exist, and the view's queries mostly miss)
    const bool use_occ = GRID && p.grid_occ != nullptr && *(const kuint32*)p.grid_occ_flag != 0u;
    CullRec cull0;
    if (KIND == 3) cull0 = load_cull(reinterpret_cast<const float4*>(p.cull), threadIdx.x & 63, p.num_triangles);

    auto intersect = [&](v3 o, v3 d, bool want, v3& n_out, float& t_out, int& id_out) -> bool {
        // wave-wide for KIND 3 (its cull step needs every lane), otherwise only the lanes that want a hit take part
        Hit h;
        h.id = -1; h.t = 1.0e38f; h.u = h.v = h.ood = 0.f;
        v3 n = V(0.f, 0.f, 0.f);
#ifdef PT_V_COUNT_ITERS
        const unsigned tests_before = lane_tests;
#endif
        if (KIND == 3) {
            h = intersect_mesh_culled(p, d, want, cull0, tri_tests);
        } else if (GRID) {
            n = n_out;  // (kept when nothing is hit: no register to clear, and the caller reads the normal only after a hit)
            h = intersect_grid(p, d, want, lane_tests, n, use_occ);
        } else if (want) {
            if (KIND == 0) h = intersect_world(p.spheres, o, d, n);
            else if (KIND == 1) h = MATERIALS ? intersect_mesh_from(p.tris_abs, p.num_triangles, o, d, loop_full) : intersect_mesh(p.tris, live_triangles(p), d, loop_full);
            else if (KIND == 2) h = intersect_bvh<MATERIALS>(p, o, d, s_stack[threadIdx.x >> 6], node_visits, tri_tests);
            else if (KIND == 5) {
                bool overflow;
                h = intersect_bvh_lanes(p, o, d, s_lane_stack[threadIdx.x >> 6], overflow, lane_tests);
                if (overflow) h = intersect_bvh<true>(p, o, d, s_stack[threadIdx.x >> 6], node_visits, tri_tests);
            }
        }
#ifdef PT_V_COUNT_ITERS
        // measurement variant, direction grid: trips of the list loop as the WAVE ran them (the longest list among its lanes)
        // -> node_visits; wave queries in which more than one lane had something to test -> tri_tests
        if (GRID) {
            const unsigned mine = want ? lane_tests - tests_before : 0u;
            node_visits += __builtin_amdgcn_readfirstlane(__reduce_max_trips(mine));
            tri_tests += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(mine > 1u));
        }
#endif
        const bool hit = want && h.id >= 0;
        if (hit && KIND != 0 && !(GRID && p.normal_mode != 0)) n = mesh_normal(p, h);  // (the grid's records hold the geometric normal)
        n_out = n; t_out = h.t; id_out = h.id;
        return hit;
    };

    // The shader has no pixel jitter: the camera ray of a pixel (fs:381-383) is the same for every sample, and so is what
    // it hits. With primary reuse the first closest-hit query of a sample is evaluated once per block and pixel and its
    // result (normal, distance, triangle) is what every further sample of the block starts from.
    const bool reuse = p.primary_reuse != 0 && p.depth > 0;  // u_Depth <= 0: the bounce loop (fs:388) never runs, no query at all
    v3 N0 = V(0.f, 0.f, 0.f);
    float t0 = 1.0e38f;
    int id0 = -1;
    bool hit0 = false;
    if (reuse) {
        const bool want0 = remaining > 0;
        if (!MATERIALS && p.primary_cache && p.primary_cache_valid) {
            // an earlier dispatch of this same view left the pixel's camera-ray hit (normal, triangle) behind: frame
            // loops of one sample per dispatch do not evaluate it again
            if (want0) {
                const float4 c = p.primary_cache[pix];
                N0 = V(c.x, c.y, c.z);
                id0 = (int)__float_as_uint(c.w);
                hit0 = id0 >= 0;
                first_from_cache = true;
            }
        } else {
            if (KIND == 3 || want0) hit0 = intersect(ro0, rd0, want0, N0, t0, id0);
            if (__builtin_amdgcn_ballot_w64(want0)) ++trips;
            if (want0) ++traced;
            if (!MATERIALS && p.primary_cache && want0) p.primary_cache[pix] = make_float4(N0.x, N0.y, N0.z, __uint_as_float((uint32_t)id0));
        }
    }

    // path state of the sample in flight
    v3 ro = ro0, rd = rd0;
    v3 energy = V(1.0f, 1.0f, 1.0f);      // the reference mode keeps the three channels equal and only uses .x
    v3 emitted = V(0.f, 0.f, 0.f);
    int level = 0;
    rng.samples = (float)sample;
    rng.sx = 0.0f;
    // the hit waiting to be shaded: the reused primary hit at the start of a sample, later the one found by the last query
    bool pending = reuse;           // false: the sample's first query is still to be evaluated
    bool pend_hit = hit0;
    v3 pend_n = N0;
    float pend_t = t0;
    int pend_id = id0;

    // A wave none of whose pixels sees the scene (most waves of a sparse view): every sample of such a pixel ends at its
    // camera ray with the same sky colour (fs:408-410 with energy 1), so the samples are folded in a short loop here
    // instead of one trip of the big loop each. Once the running mean equals that colour, further folds leave it as it
    // is (x + a * (y - x) with y - x = 0), so the loop stops there.
#ifndef PT_V_NO_SKY
    if (reuse && __builtin_amdgcn_ballot_w64(hit0 && remaining > 0) == 0ull) {
        if (remaining > 0) {
            const float t = 0.5f * (rd0.y + 1.0f);
            const float omt = 1.0f - t;
            const float sr = omt + 0.5f * t, sg = omt + 0.7f * t, sb = omt + 1.0f * t;
            for (int k = 0; k < sample_count; ++k) {
                if (scratch_slot >= 0) {
                    park(&scratch[(size_t)k * PT_TILE_THREADS], sr, sg, sb, 1.0f, rd0.y);
                } else if (p.accum_mode == 0) {
                    if (mr == sr && mg == sg && mb == sb) break;
                    const float a = ieee_rcp((float)sample);
                    mr = mr + a * (sr - mr);
                    mg = mg + a * (sg - mg);
                    mb = mb + a * (sb - mb);
                } else {
                    mr += sr; mg += sg; mb += sb;
                }
                sample += p.sample_stride;
            }
            remaining = 0;
        }
    }
#endif

    // One loop trip = [shade the pending hit, Shade() fs:224-361] then [closest hit of the new direction]; a sample that
    // ends (miss, throughput 0, depth reached: fs:388,402) is folded into the pixel and the lane starts its next sample
    // in the same trip, so lanes of a wave stay busy although their paths differ in length.
    // KIND 3 keeps all 64 lanes in the loop until the wave is done (its cull step is a wave-wide operation); the other
    // kinds simply let finished lanes drop out.
    // Shade() of the pending hit (fs:224-361); returns whether the sample ends with it
    auto shade_pending = [&]() -> bool {
        if (pend_hit) {
            ++shaded;
            if (MATERIALS) {
                const v3 P = ro + rd * pend_t;  // recPoint, fs:163
                shade_materials(rng, p.materials, p.material_ids[pend_id], energy, emitted, ro, rd, P, pend_n, p.shading_mode == 2);
            } else {
                shade(rng, energy.x, rd, pend_n);
            }
        }
        ++level;
        const bool spent = MATERIALS ? (energy.x == 0.0f && energy.y == 0.0f && energy.z == 0.0f) : energy.x == 0.0f;
        return !pend_hit || spent || level >= p.depth;
    };
    // the end of a sample (fs:408-415): folded into the pixel, and the lane's state set for its next sample
    auto end_sample = [&]() {
        const float t = 0.5f * (rd.y + 1.0f);
        const float omt = 1.0f - t;
        const float ex = energy.x, ey = MATERIALS ? energy.y : energy.x, ez = MATERIALS ? energy.z : energy.x;
        float sr = ex * (omt + 0.5f * t), sg = ey * (omt + 0.7f * t), sb = ez * (omt + 1.0f * t);
        if (MATERIALS) { sr = sr + emitted.x; sg = sg + emitted.y; sb = sb + emitted.z; }
        if (scratch_slot >= 0) {
            park(&scratch[(size_t)(sample_count - remaining) * PT_TILE_THREADS], sr, sg, sb, energy.x, rd.y);
        } else if (p.accum_mode == 0) {
            const float a = ieee_rcp(rng.samples);
            mr = mr + a * (sr - mr);
            mg = mg + a * (sg - mg);
            mb = mb + a * (sb - mb);
        } else {
            mr += sr; mg += sg; mb += sb;
        }
        --remaining;
        sample += p.sample_stride;
        ro = ro0; rd = rd0; energy = V(1.0f, 1.0f, 1.0f); emitted = V(0.f, 0.f, 0.f); level = 0;
        rng.samples = (float)sample;
        rng.sx = 0.0f;
        pending = reuse; pend_hit = hit0; pend_n = N0; pend_t = t0; pend_id = id0;
    };
    // Direction grid, second form of the loop (KIND 7; the host times both forms on a view's first long dispatches and keeps the
    // quicker one): one triangle test per trip; a lane whose cell lists more carries the rest of the list (entry,
    // position; best hit so far in pend_t / pend_id / pend_n; u, v, 1/det for an interpolated normal in LDS) into its next trip
    // and shades one trip later -- instead of the wave running its list loop as often as its longest list, for a few lanes.
    // Same tests, same lowest-index rule: the same image either way.
    if (KIND == 7) {
        __shared__ float s_uvo[PT_TILE_WAVES][3 * 64];
        float* __restrict__ uvo = &s_uvo[threadIdx.x >> 6][threadIdx.x & 63];
        uint32_t carry_entry = 0u, carry_pos = 0u;
        bool listing = false;
        while (remaining > 0) {
#ifdef PT_V_LIVE_COUNT
            node_visits += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true));  // measurement build: lanes still in the loop, per trip
#endif
            bool done = false;
            if (!listing) {
                if (p.depth <= 0) done = true;
                else if (pending) done = shade_pending();
            }
            const bool want = !listing && !done;
            bool more = listing;
            if (want) {
                more = grid_lookup(p, rd, true, use_occ, carry_entry, carry_pos);
                pend_t = 1.0e38f; pend_id = -1;
                ++traced;
            }
            if (__builtin_amdgcn_ballot_w64(want || listing)) ++trips;
            if (more) more = grid_test_one(p, rd, carry_entry, carry_pos, pend_t, pend_id, pend_n, uvo, lane_tests);
            if ((want || listing) && !more) {   // the query is answered
                pend_hit = pend_id >= 0;
                if (pend_hit && p.normal_mode == 0) {
                    Hit h;
                    h.t = pend_t; h.id = pend_id; h.u = uvo[0]; h.v = uvo[64]; h.ood = uvo[128];
                    pend_n = mesh_normal(p, h);
                }
                pending = true;
                done = !pend_hit;
            }
            listing = more;
            if (done) end_sample();
        }
    } else
    while (KIND == 3 ? __builtin_amdgcn_ballot_w64(remaining > 0) != 0 : remaining > 0) {
        const bool live = remaining > 0;
#ifdef PT_V_LIVE_COUNT
        if (GRID) node_visits += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true));
#endif
        bool done = false;
        if (live && p.depth <= 0) {
            done = true;
        } else if (live && pending) {
            done = shade_pending();
        }
#ifdef PT_V_RESHADE
        // measurement build: a lane whose sample ended in Shade() has no ray for this trip's closest-hit part; here it folds the
        // sample at once and shades its next sample's reused camera-ray hit, so that it enters that part with a ray
        if (KIND == 1 && MATERIALS && reuse && live && done && p.depth > 0) {
            end_sample();
            if (remaining <= 0) continue;
            done = shade_pending();
        }
#endif
        const bool want = live && !done;
        if (KIND == 3 || want) pend_hit = intersect(ro, rd, want, pend_n, pend_t, pend_id);
        if (__builtin_amdgcn_ballot_w64(want)) ++trips;
        if (want) {
            ++traced;
            pending = true;
            done = !pend_hit;  // a miss ends the sample in this same trip
        }
        if (live && done) end_sample();
    }
    if (in_window && scratch_slot < 0) {
        const float alpha = p.accum_mode == 0 ? 1.0f : prev.w + (float)p.num_samples;
        reinterpret_cast<float4*>(p.dst)[pix] = make_float4(mr, mg, mb, alpha);
    }
    unsigned rays = traced;
    if (reuse && in_window && sample_count > 0) rays += (unsigned)(sample_count - 1) + (first_from_cache ? 1u : 0u);
    const unsigned long long wave_rays = wave_sum(rays), wave_traced = wave_sum(traced);
    const unsigned long long wave_tests = (GRID || KIND == 5) ? wave_sum(lane_tests) : 0ull;
    const unsigned long long wave_shaded = wave_sum(shaded);
    const unsigned wave_trips = __builtin_amdgcn_readfirstlane(__reduce_max_trips(trips));
    if ((threadIdx.x & 63) == 0 && wave_rays) {
        unsigned long long* counters = p.counters + (size_t)(blockIdx.x % PT_COUNTER_SLOTS) * PT_COUNTER_STRIDE;
        atomicAdd(&counters[0], wave_rays);
        atomicAdd(&counters[1], (unsigned long long)wave_trips * 64ull);
        atomicAdd(&counters[4], wave_traced);
        // Scheduling cost of this tile: proportional to the time the wave took. A trip of the all-triangles loop always
        // costs the same; a trip through the hierarchy costs its node and triangle records plus a fixed share.
        if (p.tile_cost) {
            unsigned cost = wave_trips;
            if (KIND == 2) cost = 8u * wave_trips + __builtin_amdgcn_readfirstlane(node_visits) + __builtin_amdgcn_readfirstlane(tri_tests);
            if (GRID || KIND == 5) cost = 8u * wave_trips + (unsigned)(wave_tests >> 4);
            atomicAdd(&p.tile_cost[tile], cost);
        }
#ifdef PT_V_LIVE_COUNT
        if (GRID) atomicAdd(&counters[2], (unsigned long long)__builtin_amdgcn_readfirstlane(node_visits));
#endif
#ifdef PT_V_COUNT_ITERS
        if (GRID) {
            atomicAdd(&counters[2], (unsigned long long)__builtin_amdgcn_readfirstlane(node_visits));
            atomicAdd(&counters[3], (unsigned long long)__builtin_amdgcn_readfirstlane(tri_tests));
        }
#endif
        if (KIND == 2 || KIND == 3 || KIND == 5) {  // 5: what the per-wave walk did for lanes whose own stack was full
            atomicAdd(&counters[2], (unsigned long long)__builtin_amdgcn_readfirstlane(node_visits));
            atomicAdd(&counters[3], (unsigned long long)__builtin_amdgcn_readfirstlane(tri_tests));
        }
        if (GRID || KIND == 5) atomicAdd(&counters[5], wave_tests);
        if (wave_shaded) atomicAdd(&counters[6], wave_shaded);
        if (KIND == 1) {  // the loop visits every (live) record once per trip of the wave
            const unsigned n = MATERIALS ? (unsigned)p.num_triangles : (unsigned)live_triangles(p);
            atomicAdd(&counters[7], (unsigned long long)wave_trips * n);
            atomicAdd(&counters[8], (unsigned long long)__builtin_amdgcn_readfirstlane(loop_full));
        }
    }
}

// ---- diffuse/emissive mode through the LBVH, lanes decoupled -------------------------------------------------------------
// Rays that bounce diffusely from scattered points have little in common: the hierarchy walks of the 64 lanes of a wave differ
// several-fold in length, and in pt_render_kernel<5, true> every lane waits for the longest walk of its wave before any of
// them is shaded (measured: 15 % of the lanes active over the kernel). Here a lane is a small state machine and the wave's
// loop only asks "what does each lane need next":
//   WALK   one round of its own hierarchy walk: a few node steps over the four-child form of the hierarchy, then the leaf it
//          arrived at (the shader's triangle test with the lowest-index rule for equal distances, and boxes that only ever skip
//          what that test could not accept: the same hit as every other search mode);
//   SHADE  its walk is over (or its sample starts from the reused camera-ray hit): it waits until PT_SHADE_BATCH lanes are
//          waiting, or nobody walks any more, and then the wave runs Shade(), the end of the sample (fs:408-415) and the
//          start of the next query for all of them at once;
//   DONE   no samples left.
// A lane that finishes a walk early goes on to its next bounce while its neighbours are still walking. The image is the
// one pt_render_kernel<5, true> renders (and the all-triangles loop, and the shader): per pixel the samples are still folded
// in order by the lane that owns the pixel.
// lanes that wait before the wave shades (over the surface-area hierarchy, with 2 node steps per round; 1M triangles / 3,936 / 968 /
// the torus pair: 16: 8.21 / 4.73 / 2.93 / 5.07 ms, 24: 8.26 / 4.65 / 2.84 / 4.87, 32: 8.38 / 4.72 / 2.89 / 4.77)
#ifndef PT_SHADE_BATCH
#define PT_SHADE_BATCH 24
#endif
// node steps per round before the lane looks at its leaf. Over the Morton-code hierarchy the time was flat between 2 and 3 (1M
// triangles / 3,936 / 968: 2 steps 11.9 / 5.74 / 3.63 ms, 3: 11.8 / 5.74 / 3.71, 8: 13.2 / 6.90 / 4.13); over the surface-area
// hierarchy, whose walks are shorter, 2 is better (1: 8.87 / 4.80 / 2.93, 2: 8.21 / 4.73 / 2.93, 3: 8.38 / 4.99 / 3.12, 4: 8.69 /
// 5.17 / 3.33)
#ifndef PT_WALK_NODE_STEPS
#define PT_WALK_NODE_STEPS 2
#endif

struct LaneWalk {
    uint32_t ref;  // node or leaf reference the lane looks at next
    int sp;        // entries on the lane's stack
    Hit h;
    float ax, ay, az, bx, by, bz;  // a bound stored as q steps of the frame lies at ray parameter fma(q, a, b) on its axis
};

__device__ __forceinline__ void walk_begin(const PtKernelParams& p, LaneWalk& w, v3 ro, v3 rd) {
    const float tiny = 1.0e-20f;
    const float ix = __builtin_amdgcn_rcpf(__builtin_fabsf(rd.x) > tiny ? rd.x : __builtin_copysignf(tiny, rd.x));
    const float iy = __builtin_amdgcn_rcpf(__builtin_fabsf(rd.y) > tiny ? rd.y : __builtin_copysignf(tiny, rd.y));
    const float iz = __builtin_amdgcn_rcpf(__builtin_fabsf(rd.z) > tiny ? rd.z : __builtin_copysignf(tiny, rd.z));
    const float* __restrict__ f = p.bvh_qframe;  // base xyz, step xyz
    w.ax = f[3] * ix; w.ay = f[4] * iy; w.az = f[5] * iz;
    w.bx = (f[0] - ro.x) * ix; w.by = (f[1] - ro.y) * iy; w.bz = (f[2] - ro.z) * iz;
    w.ref = 0;  // root = internal node 0
    w.sp = 0;
    w.h.t = 1.0e38f; w.h.id = -1; w.h.u = 0.f; w.h.v = 0.f; w.h.ood = 0.f;
}

// entry and exit parameter of the ray in one child's box (three words lo | hi << 16), clipped to [0, limit]
__device__ __forceinline__ void qbox_span(const LaneWalk& w, uint32_t qx, uint32_t qy, uint32_t qz, float limit, float& tnear, float& tfar) {
    const float x0 = __builtin_fmaf((float)(qx & 0xffffu), w.ax, w.bx), x1 = __builtin_fmaf((float)(qx >> 16), w.ax, w.bx);
    const float y0 = __builtin_fmaf((float)(qy & 0xffffu), w.ay, w.by), y1 = __builtin_fmaf((float)(qy >> 16), w.ay, w.by);
    const float z0 = __builtin_fmaf((float)(qz & 0xffffu), w.az, w.bz), z1 = __builtin_fmaf((float)(qz >> 16), w.az, w.bz);
    tnear = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fmaxf(__builtin_fminf(z0, z1), 0.0f));
    tfar = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fminf(__builtin_fmaxf(z0, z1), limit));
}

// One round of the lanes' walks. Called by ALL lanes of the wave, `active` = this lane is walking: the loops below are the
// wave's, so the counts of records fetched (steps: nodes and triangles, records: triangles; summed over the lanes) stay in
// scalar registers -- the lanes have no registers to spare for statistics. Returns, per lane, 0 = more to do (or not active),
// 1 = the walk is over (w.h is the closest hit), 2 = the lane's stack is full (the caller repeats the query with the per-wave
// walk, which has room for any hierarchy).
__device__ __forceinline__ int walk_round(const PtKernelParams& p, LaneWalk& w, v3 ro, v3 rd, uint32_t* __restrict__ my, bool active,
                                          unsigned& steps, unsigned& records) {
    const uint4* __restrict__ nodes = reinterpret_cast<const uint4*>(p.bvh_nodes_w);
    const float4* __restrict__ recs = reinterpret_cast<const float4*>(p.bvh_tris_q);
    int result = 0;
    for (int s = 0; s < PT_WALK_NODE_STEPS; ++s) {
        const bool go = active && result == 0 && !(w.ref & 0x80000000u);
        const unsigned long long going = __builtin_amdgcn_ballot_w64(go);
        if (!going) break;
        steps += (unsigned)__builtin_popcountll(going);
        if (go) {
            const size_t at = 4 * (size_t)w.ref;
            const uint4 c0 = nodes[at], c1 = nodes[at + 1], c2 = nodes[at + 2], c3 = nodes[at + 3];
#ifdef PT_V_EXTRA_VALU
            // measurement build: vector instructions that do nothing, to see how the time follows the instruction count
#pragma unroll
            for (int k_ = 0; k_ < PT_V_EXTRA_VALU; ++k_) { int t_; asm volatile("v_mov_b32 %0, 0" : "=v"(t_)); }
#endif
            const float limit = w.h.t * 1.0001f;
            const float inf = __builtin_inff();
            float n0, n1, n2, n3, f;
            qbox_span(w, c0.x, c0.y, c0.z, limit, n0, f); if (!(n0 <= f) || c0.w == PT_BVH_NO_CHILD) n0 = inf;
            qbox_span(w, c1.x, c1.y, c1.z, limit, n1, f); if (!(n1 <= f) || c1.w == PT_BVH_NO_CHILD) n1 = inf;
            qbox_span(w, c2.x, c2.y, c2.z, limit, n2, f); if (!(n2 <= f) || c2.w == PT_BVH_NO_CHILD) n2 = inf;
            qbox_span(w, c3.x, c3.y, c3.z, limit, n3, f); if (!(n3 <= f) || c3.w == PT_BVH_NO_CHILD) n3 = inf;
            uint32_t r0 = c0.w, r1 = c1.w, r2 = c2.w, r3 = c3.w;
            // children in the order the ray enters them (those it misses last): a sorting network on four keys
#define PT_ORDER(na, ra, nb, rb) { const bool sw = nb < na; const float tn = sw ? na : nb; na = sw ? nb : na; nb = tn; \
                                   const uint32_t tr = sw ? ra : rb; ra = sw ? rb : ra; rb = tr; }
            PT_ORDER(n0, r0, n1, r1) PT_ORDER(n2, r2, n3, r3) PT_ORDER(n0, r0, n2, r2) PT_ORDER(n1, r1, n3, r3) PT_ORDER(n1, r1, n2, r2)
#undef PT_ORDER
            const int hits = (n0 < inf) + (n1 < inf) + (n2 < inf) + (n3 < inf);
            if (hits == 0) {
                if (w.sp == 0) {
                    result = 1;
                } else {
                    --w.sp;
                    w.ref = my[w.sp * 64];
                }
            } else if (w.sp + hits - 1 > p.lane_stack_limit) {
                result = 2;
            } else {
                // the nearest next, the others wait on the stack, the farthest deepest
                if (hits > 3) { my[w.sp * 64] = r3; ++w.sp; }
                if (hits > 2) { my[w.sp * 64] = r2; ++w.sp; }
                if (hits > 1) { my[w.sp * 64] = r1; ++w.sp; }
                w.ref = r0;
            }
        }
    }
    // leaf: up to PT_BVH_WIDE_LEAF_MAX consecutive records
    const bool leaf = active && result == 0 && (w.ref & 0x80000000u);
    const int first = (int)((w.ref & 0x7fffffffu) >> 3), count = leaf ? (int)(w.ref & 7u) + 1 : 0;
    auto test_record = [&](const float4 ra, const float4 rb, const float4 rc, bool live) {
        // fs:136-168 as written: record = a.xyz e1.x | e1.yz e2.xy | e2.z id - -
        const v3 a = V(ra.x, ra.y, ra.z), e1 = V(ra.w, rb.x, rb.y), e2 = V(rb.z, rb.w, rc.x);
        const v3 q = cross3(rd, e2);
        const float det = dot3(e1, q);
        const v3 ap = ro - a;
        const v3 r = cross3(ap, e1);
        const float u = dot3(ap, q);
        const float v = dot3(rd, r);
        if (live & (det >= 0.0001f) & (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= det)) {
            const float ood = ieee_rcp(det);
            const float tt = dot3(e2, r) * ood;
            const int id = (int)__float_as_uint(rc.y);
            if (tt > 0.0f && (tt < w.h.t || (tt == w.h.t && id < w.h.id))) {
                w.h.t = tt; w.h.id = id; w.h.u = u; w.h.v = v; w.h.ood = ood;
            }
        }
    };
#if PT_BVH_WIDE_LEAF_MAX == 2
    {   // both records are requested before either is looked at (one wait instead of two)
        const unsigned long long at_leaf = __builtin_amdgcn_ballot_w64(leaf);
        if (at_leaf) {
            const bool two = count > 1;
            const unsigned n = (unsigned)__builtin_popcountll(at_leaf) + (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(two));
            steps += n;
            records += n;
            if (leaf) {
                const size_t at0 = 3 * (size_t)first, at1 = 3 * (size_t)(first + (two ? 1 : 0));
                const float4 a0 = recs[at0], b0 = recs[at0 + 1], c0 = recs[at0 + 2];
                const float4 a1 = recs[at1], b1 = recs[at1 + 1], c1 = recs[at1 + 2];
                test_record(a0, b0, c0, true);
                test_record(a1, b1, c1, two);
            }
        }
    }
#else
    for (int k = 0; k < 8; ++k) {
        const bool go = k < count;
        const unsigned long long going = __builtin_amdgcn_ballot_w64(go);
        if (!going) break;
        const unsigned n = (unsigned)__builtin_popcountll(going);
        steps += n;
        records += n;
        if (go) {
            const size_t at = 3 * (size_t)(first + k);
            test_record(recs[at], recs[at + 1], recs[at + 2], true);
        }
    }
#endif
    if (leaf) {
        if (w.sp == 0) {
            result = 1;
        } else {
            --w.sp;
            w.ref = my[w.sp * 64];
        }
    }
    return result;
}

// waves per SIMD the register allocation must allow: the walk waits on dependent fetches most of the time, so a fifth wave
// is worth the few registers it costs (measured: 4 -> 5: -15 % time, 6: no further gain)
#ifndef PT_DECOUPLED_MIN_WAVES
#define PT_DECOUPLED_MIN_WAVES 5
#endif
__global__ __launch_bounds__(64, PT_DECOUPLED_MIN_WAVES) void pt_render_decoupled_kernel(const PtKernelParams p) {
    __shared__ uint32_t s_stack[PT_BVH_STACK];
    __shared__ uint32_t s_lane_stack[PT_LANE_STACK * 64];
    int x, y;
    int tile = (int)blockIdx.x, sample_begin = 0, sample_count = p.num_samples, scratch_slot = -1;
    if (p.items) {
        unsigned index;
        if (!work_item_index(p, index)) return;
        const kuint32* item = (const kuint32*)(p.items + index);
        if (item[0] == PT_NO_TILE) return;  // a position past the end of its group's part of the list
        tile = (int)item[0]; sample_begin = (int)item[1]; sample_count = (int)item[2]; scratch_slot = (int)item[3];
    }
    const bool in_window = pixel_of_lane(p, tile, x, y);
    const size_t pix = in_window ? ((size_t)y * p.width + x) : 0;
    const v3 ro0 = V(p.ro[0], p.ro[1], p.ro[2]);
    Rng rng;
    rng.fcx = (float)x + 0.5f;
    rng.fcy = (float)y + 0.5f;
    rng.samples = 0.0f;
    rng.sx = 0.0f;
    float mr = 0.f, mg = 0.f, mb = 0.f, malpha = 1.0f;
    if (in_window && scratch_slot < 0) {
        const float4 prev = reinterpret_cast<const float4*>(p.src)[pix];
        mr = prev.x; mg = prev.y; mb = prev.z;
        if (p.accum_mode != 0) malpha = prev.w + (float)p.num_samples;
    }
    float4* __restrict__ scratch = nullptr;  // where the pixel's next sample is parked (split tiles)
    if (scratch_slot >= 0)
        scratch = reinterpret_cast<float4*>(p.sample_scratch) + ((size_t)scratch_slot * p.num_samples + sample_begin) * PT_TILE_THREADS + threadIdx.x;
    uint32_t* __restrict__ my = s_lane_stack + (threadIdx.x & 63);

    int remaining = in_window ? sample_count : 0;
    int sample = p.first_sample + sample_begin * p.sample_stride;
    // statistics of the wave, counted where all its lanes are together (scalar registers: the lanes have none to spare)
    unsigned traced = 0, batches = 0, rounds = 0, shaded = 0, steps = 0, records = 0, node_visits = 0, tri_tests = 0;
    const unsigned pixels = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(remaining > 0));
#ifdef PT_V_STATE_COUNT
    unsigned st_rounds = 0, st_walk = 0, st_wait = 0, st_shade_lanes = 0;  // measurement build: lane states per loop trip
    unsigned long long tk_hand = 0, tk_shade = 0, tk_walk = 0, tk_mark = __builtin_amdgcn_s_memtime();
#define PT_PHASE_D(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - tk_mark; tk_mark = now_; }
#else
#define PT_PHASE_D(acc)
#endif

    auto fold = [&](float sr, float sg, float sb) {
        if (scratch_slot >= 0) {
            *scratch = make_float4(sr, sg, sb, 0.f);
            scratch += PT_TILE_THREADS;
        } else if (p.accum_mode == 0) {
            const float a = ieee_rcp((float)sample);
            mr = mr + a * (sr - mr);
            mg = mg + a * (sg - mg);
            mb = mb + a * (sb - mb);
        } else {
            mr += sr; mg += sg; mb += sb;
        }
        --remaining;
        sample += p.sample_stride;
    };

    enum { WALK = 0, SHADE = 1, DONE = 2, FULL = 3 };
    // The camera ray of the pixel is evaluated once (the shader has no pixel jitter): with reuse every sample starts from its
    // hit; its normal is formed by the first Shade() that needs it. It is the lane's first walk, like any other.
    const bool reuse = p.primary_reuse != 0 && p.depth > 0;
    bool primary = reuse, hit0 = false, n0_known = false;
    v3 N0 = V(0.f, 0.f, 0.f);
    float t0 = 1.0e38f;
    int id0 = -1;
    // the sample in flight; the hit it waits to shade is the camera ray's (level 0 with reuse) or the one its walk left in w.h
    v3 ro = ro0, rd = primary_direction(p, rng.fcx, rng.fcy);
    v3 energy = V(1.0f, 1.0f, 1.0f);
    int level = 0;
    LaneWalk w;
    walk_begin(p, w, ro, rd);
    int state = remaining > 0 ? (p.depth > 0 ? WALK : SHADE) : DONE;
    traced += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == WALK));

    // the walk of the lane's query is over, its result in w.h
    auto walk_over = [&]() {
        if (primary) {
            primary = false;
            hit0 = w.h.id >= 0; t0 = w.h.t; id0 = w.h.id;
            if (!hit0) {
                // the camera ray leaves the scene: every sample is the sky colour of that direction (fs:408-410 with energy 1)
                const float t = 0.5f * (rd.y + 1.0f);
                const float omt = 1.0f - t;
                const float sr = omt + 0.5f * t, sg = omt + 0.7f * t, sb = omt + 1.0f * t;
                while (remaining > 0) {
                    if (scratch_slot < 0 && p.accum_mode == 0 && mr == sr && mg == sg && mb == sb) break;  // further folds leave it as it is
                    fold(sr, sg, sb);
                }
                remaining = 0;
                state = DONE;
                return;
            }
        }
        state = SHADE;
    };

    // All camera rays share their origin, so where the direction grid of this camera position is at hand the camera ray's
    // hit comes from it instead of a walk: the same exact test and lowest-index rule, a cell and a record or two to fetch.
    if (reuse && p.grid_cells) {
        if (state == WALK) {
            unsigned grid_tests = 0;
            v3 gn;
            w.h = intersect_grid(p, rd, true, grid_tests, gn, p.grid_occ != nullptr);
            walk_over();
        }
    }

    PT_PHASE_D(tk_hand)
    for (;;) {
        const unsigned long long walking = __builtin_amdgcn_ballot_w64(state == WALK);
        const unsigned long long waiting = __builtin_amdgcn_ballot_w64(state == SHADE);
        if (!walking && !waiting) break;
#ifdef PT_V_STATE_COUNT
        st_rounds += 1; st_walk += __builtin_popcountll(walking); st_wait += __builtin_popcountll(waiting);
#endif
        if (waiting && (__builtin_popcountll(waiting) >= PT_SHADE_BATCH || !walking)) {
            ++batches;
#ifdef PT_V_STATE_COUNT
            st_shade_lanes += __builtin_popcountll(waiting);
#endif
            // Shade() of the pending hit, the end of the sample, the start of the next query: until the lane walks again or
            // has no samples left (a sample that ends starts the next one from the reused camera-ray hit at once). The loop
            // is the wave's: a lane takes part while it has something to shade.
            for (;;) {
                const bool go = state == SHADE;
                if (!__builtin_amdgcn_ballot_w64(go)) break;
                bool shades = false;
                if (go) {
                    bool done = p.depth <= 0;
                    v3 emitted = V(0.f, 0.f, 0.f);  // an emitter ends the path in the Shade() that finds it
                    if (!done) {
                        const bool from_camera = reuse && level == 0;
                        const bool hit = from_camera ? hit0 : w.h.id >= 0;
                        if (hit) {
                            shades = true;
                            v3 n;
                            if (from_camera) {
                                if (!n0_known) { N0 = mesh_normal(p, w.h); n0_known = true; }  // w.h still holds the camera ray's walk
                                n = N0;
                            } else {
                                n = mesh_normal(p, w.h);
                            }
                            rng.samples = (float)sample;
                            const v3 P = ro + rd * (from_camera ? t0 : w.h.t);  // recPoint, fs:163
                            shade_materials(rng, p.materials, p.material_ids[from_camera ? id0 : w.h.id], energy, emitted, ro, rd, P, n, p.shading_mode == 2);
                        }
                        ++level;
                        const bool spent = energy.x == 0.0f && energy.y == 0.0f && energy.z == 0.0f;
                        done = !hit || spent || level >= p.depth;
                    }
                    if (!done) {
                        walk_begin(p, w, ro, rd);
                        state = WALK;
                    } else {
                        // fs:408-415
                        const float t = 0.5f * (rd.y + 1.0f);
                        const float omt = 1.0f - t;
                        const float sr = energy.x * (omt + 0.5f * t) + emitted.x, sg = energy.y * (omt + 0.7f * t) + emitted.y,
                                    sb = energy.z * (omt + 1.0f * t) + emitted.z;
                        fold(sr, sg, sb);
                        if (remaining <= 0) {
                            state = DONE;
                        } else {
                            ro = ro0; rd = primary_direction(p, rng.fcx, rng.fcy);
                            energy = V(1.0f, 1.0f, 1.0f); level = 0;
                            rng.sx = 0.0f;
                            if (!reuse && p.depth > 0) {
                                walk_begin(p, w, ro, rd);
                                state = WALK;
                            }
                        }
                    }
                }
                shaded += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(shades));
            }
            // queries that set out in this batch: lanes that walk now and did not before it
            traced += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(state == WALK) & ~walking);
        }
        PT_PHASE_D(tk_shade)
        const int r = walk_round(p, w, ro, rd, my, state == WALK, steps, records);
        if (state == WALK) {
            if (r == 2) state = FULL;
            else if (r == 1) walk_over();
        }
        // (very deep hierarchies) a lane whose own stack is full repeats its query with the per-wave walk
        if (__builtin_amdgcn_ballot_w64(state == FULL)) {
            if (state == FULL) {
                w.h = intersect_bvh<true>(p, ro, rd, s_stack, node_visits, tri_tests);
                walk_over();
            }
        }
        ++rounds;
        PT_PHASE_D(tk_walk)
    }
    if (in_window && scratch_slot < 0) reinterpret_cast<float4*>(p.dst)[pix] = make_float4(mr, mg, mb, malpha);
    if ((threadIdx.x & 63) == 0 && pixels) {
        // closest-hit queries as the reference would count them: one per sample for the camera ray plus one per bounce; with reuse
        // the camera ray of a pixel was evaluated once
        unsigned long long rays = 0;
        if (p.depth > 0) rays = reuse ? (unsigned long long)traced - pixels + (unsigned long long)pixels * (unsigned)sample_count : (unsigned long long)traced;
        unsigned long long* counters = p.counters + (size_t)(blockIdx.x % PT_COUNTER_SLOTS) * PT_COUNTER_STRIDE;
        atomicAdd(&counters[0], rays);
        atomicAdd(&counters[1], (unsigned long long)batches * 64ull);  // lane slots of the shade batches
        atomicAdd(&counters[4], (unsigned long long)traced);
        if (p.tile_cost) atomicAdd(&p.tile_cost[tile], 2u * rounds + 8u * batches);
        atomicAdd(&counters[2], (unsigned long long)__builtin_amdgcn_readfirstlane(node_visits));
        atomicAdd(&counters[3], (unsigned long long)__builtin_amdgcn_readfirstlane(tri_tests));
        atomicAdd(&counters[5], (unsigned long long)steps);
        atomicAdd(&counters[9], (unsigned long long)records);
        if (shaded) atomicAdd(&counters[6], (unsigned long long)shaded);
#ifdef PT_V_STATE_COUNT
        atomicAdd(&counters[2], (unsigned long long)st_rounds);       // reported as bvh_node_fetches
        atomicAdd(&counters[3], (unsigned long long)st_walk);         // bvh_triangle_fetches
        atomicAdd(&counters[7], (unsigned long long)st_wait);         // loop_triangle_visits
        atomicAdd(&counters[8], (unsigned long long)st_shade_lanes);  // loop_full_tests
        atomicAdd(&counters[10], tk_hand);
        atomicAdd(&counters[11], tk_shade);
        atomicAdd(&counters[12], tk_walk);
#endif
    }
#undef PT_PHASE_D
}

// ---- the reference's CPU tracer (PT_PROGRAM_CPU_TRACER) ---------------------------------------------------------------------
// Scene::color (PathTracer.cpp:932-1052) with scatter (:830-852), Sphere::hit (:595-640), Mesh::hit (:562-586: bounding box
// test, then every triangle in order), HitRecord::hit (:101-152) and Camera::generateRay (:742-745, through the pixel's
// corner): spheres AND mesh in one scene, Lambert bounce N + randomUnitVector() from the hit point, the other material types
// pass the ray on unchanged, sky (1-t) + (1, 0.7, 0.5) t, running SUM per pixel. The reference draws its random numbers from
// one serial libc rand() stream; here they come from a counter-based generator keyed by (seed, pixel, sample), so pixels and
// samples are independent. float3's sums run left to right (:61), unlike the shader's.
__device__ __forceinline__ float dotc(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ v3 normalizec(v3 a) { return a * (1.0f / __builtin_sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z)); }
__device__ __forceinline__ uint32_t pcg_hash(uint32_t x) {
    const uint32_t state = x * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
struct CtrRng {
    uint32_t key, draw;
    // nextFloat(0, 1), PathTracer.cpp:765-768: (float)rand() / (float)RAND_MAX with a 31-bit value
    __device__ __forceinline__ float next() {
        const uint32_t h = pcg_hash(key + pcg_hash(draw++));
        return (float)(h >> 1) / 2147483648.0f;
    }
};

__global__ __launch_bounds__(64) void pt_cpu_tracer_kernel(const PtKernelParams p) {
    int x, y;
    const bool mine = pixel_of_lane(p, (int)blockIdx.x, x, y);
    const size_t pix = mine ? (size_t)y * p.width + x : 0;
    const v3 pos = V(p.ro[0], p.ro[1], p.ro[2]), llc = V(p.llc[0], p.llc[1], p.llc[2]);
    const v3 hor = V(p.hor[0], p.hor[1], p.hor[2]), ver = V(p.ver[0], p.ver[1], p.ver[2]);
    const float fx = (float)x / (float)p.width, fy = (float)y / (float)p.height;
    const v3 dir0 = normalizec(((llc + hor * fx) + ver * fy) - pos);
    const float4 prev = reinterpret_cast<const float4*>(p.src)[pix];
    float sr = prev.x, sg = prev.y, sb = prev.z;
    const float4* __restrict__ spheres = reinterpret_cast<const float4*>(p.raw_spheres);
    const float4* __restrict__ mats = reinterpret_cast<const float4*>(p.materials);
    const float4* __restrict__ recs = reinterpret_cast<const float4*>(p.tris_abs);
    unsigned rays = 0;
    for (int s = 0; mine && s < p.num_samples; ++s) {
        CtrRng rng;
        rng.key = pcg_hash(p.cpu_seed + pcg_hash((uint32_t)(y * p.width + x))) + pcg_hash(0x9E3779B9u * (uint32_t)(p.first_sample + s * p.sample_stride));
        rng.draw = 0;
        v3 ro = pos, rd = dir0;
        float mint = 0.0f, maxt = p.cpu_max_t;
        v3 energy = V(1.0f, 1.0f, 1.0f);
        for (int level = 0; level < p.depth; ++level) {
            ++rays;
            v3 rec_point = V(0.f, 0.f, 0.f), rec_normal = V(0.f, 0.f, 0.f);
            float rec_t = 0.0f;
            int hit_index = -1;
            for (int i = 0; i < 10; ++i) {
                const float4 sp = spheres[i];
                const v3 c = V(sp.x, sp.y, sp.z);
                const v3 to = ro - c;
                const float A = dotc(rd, rd), B = dotc(to, rd), C = dotc(to, to) - sp.w * sp.w;
                const float disc = B * B - A * C;
                if (disc > 0.0f) {
                    const float root = __builtin_sqrtf(disc);
                    float temp = (-B - root) / A;
                    bool ok = temp < maxt && temp > mint;
                    if (!ok) {
                        temp = (-B + root) / A;
                        ok = temp < maxt && temp > mint;
                    }
                    if (ok) {
                        rec_point = ro + rd * temp;
                        rec_normal = (rec_point - c) * (1.0f / sp.w);
                        rec_t = temp;
                        maxt = temp;
                        hit_index = i;
                    }
                }
            }
            if (p.num_triangles > 0) {
                // AABB::test with the ray as it stands (Mesh::hit works on a copy of it)
                float tmin = mint, tmax = maxt;
                bool inside = true;
                const float o3[3] = {ro.x, ro.y, ro.z}, d3[3] = {rd.x, rd.y, rd.z};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (!inside) break;
                    if (__builtin_fabsf(d3[k]) < 0.001f) {
                        if (o3[k] < p.cpu_bounds[k] || o3[k] > p.cpu_bounds[3 + k]) inside = false;
                    } else {
                        const float ood = 1.0f / d3[k];
                        float t1 = (p.cpu_bounds[k] - o3[k]) * ood, t2 = (p.cpu_bounds[3 + k] - o3[k]) * ood;
                        if (t1 > t2) { const float sw = t1; t1 = t2; t2 = sw; }
                        if (t1 > tmin) tmin = t1;
                        if (t2 < tmax) tmax = t2;
                        if (tmin > tmax) inside = false;
                    }
                }
                if (inside) {
                    float best = maxt, bu = 0.f, bv = 0.f, bood = 0.f;
                    int bid = -1;
                    for (int i = 0; i < p.num_triangles; ++i) {
                        const float4 ra = recs[4 * (size_t)i], rb = recs[4 * (size_t)i + 1], rc = recs[4 * (size_t)i + 2];
                        const v3 a = V(ra.x, ra.y, ra.z), e1 = V(ra.w, rb.x, rb.y), e2 = V(rb.z, rb.w, rc.x);
                        const v3 q = cross3(rd, e2);
                        const float det = dotc(e1, q);
                        if (det < 0.0001f) continue;
                        const v3 ap = ro - a;
                        const float u = dotc(ap, q);
                        if (u < 0.0f || u > det) continue;
                        const v3 r = cross3(ap, e1);
                        const float v = dotc(rd, r);
                        if (v < 0.0f || (u + v) > det) continue;
                        const float ood = 1.0f / det;
                        const float temp = dotc(e2, r) * ood;
                        if (temp > mint && temp < best) { best = temp; bu = u; bv = v; bood = ood; bid = i; }
                    }
                    if (bid >= 0) {
                        // point, barycentrics and normal of the accepted triangle (PathTracer.cpp:139-146)
                        const float* __restrict__ nn = p.vnormals + (size_t)bid * 9;
                        const float uy = bu * bood, uz = bv * bood, ux = (1.0f - uy) - uz;
                        rec_t = best;
                        rec_point = ro + rd * best;
                        rec_normal = normalizec((V(nn[0], nn[1], nn[2]) * ux + V(nn[3], nn[4], nn[5]) * uy) + V(nn[6], nn[7], nn[8]) * uz);
                        maxt = best;
                        hit_index = 0;   // the mesh takes the first material (PathTracer.cpp:975)
                    }
                }
            }
            if (hit_index < 0) break;
            const float4 m = mats[hit_index];
            v3 attenuation = V(1.0f, 1.0f, 1.0f);
            // scatter(), PathTracer.cpp:830-852
            ro = rec_point + rec_normal * 0.001f;
            mint = 0.0f;
            maxt = 3.402823466e+38f;
            if (m.w == 0.0f) {  // Material::LAMBERT; the other types leave the direction as it is (their code is commented out)
                const float z = rng.next() * 2.0f - 1.0f;
                const float a = (rng.next() * 2.0f) * 3.14159274f;
                const float r = __builtin_sqrtf(1.0f - z * z);
                rd = normalizec(rec_normal + V(lp_sincos<true>(a) * r, lp_sincos<false>(a) * r, z));
                attenuation = V(m.x, m.y, m.z);
            }
            energy = V(energy.x * attenuation.x, energy.y * attenuation.y, energy.z * attenuation.z);
            if ((energy.x + energy.y) + energy.z == 0.0f) break;
        }
        const float t = 0.5f * (rd.y + 1.0f);
        sr += energy.x * ((1.0f - t) + 1.0f * t);
        sg += energy.y * ((1.0f - t) + 0.7f * t);
        sb += energy.z * ((1.0f - t) + 0.5f * t);
    }
    if (mine) reinterpret_cast<float4*>(p.dst)[pix] = make_float4(sr, sg, sb, prev.w + (float)p.num_samples);
    const unsigned long long wave_rays = wave_sum(rays);
    if ((threadIdx.x & 63) == 0 && wave_rays) {
        unsigned long long* counters = p.counters + (size_t)(blockIdx.x % PT_COUNTER_SLOTS) * PT_COUNTER_STRIDE;
        atomicAdd(&counters[0], wave_rays);
        atomicAdd(&counters[4], wave_rays);
    }
}

template <int KIND>
__global__ __launch_bounds__(PT_TILE_THREADS) void pt_primary_kernel(const PtKernelParams p) {
    __shared__ uint32_t s_stack[KIND == 2 ? PT_TILE_WAVES : 1][KIND == 2 ? PT_BVH_STACK : 1];
    int x, y;
    const bool in_window = pixel_of_lane(p, (int)blockIdx.x, x, y);
    if ((KIND < 2 || KIND == 4) && !in_window) return;  // the BVH walk and the cull step are per wave: every lane stays
    const size_t pix = (size_t)y * p.width + x;
    const v3 ro = V(p.ro[0], p.ro[1], p.ro[2]);
    const v3 rd = primary_direction(p, (float)x + 0.5f, (float)y + 0.5f);
    Hit h;
    v3 N;
    if (KIND == 0) h = intersect_world(p.spheres, ro, rd, N);
    unsigned nv = 0, tt = 0;
    if (KIND == 1) h = intersect_mesh(p.tris, live_triangles(p), rd, tt);
    if (KIND == 2) h = intersect_bvh<false>(p, ro, rd, s_stack[threadIdx.x >> 6], nv, tt);
    if (KIND == 3) h = intersect_mesh_culled(p, rd, in_window, load_cull(reinterpret_cast<const float4*>(p.cull), threadIdx.x & 63, p.num_triangles), tt);
    if (KIND == 4) h = intersect_grid(p, rd, true, tt, N, p.grid_occ != nullptr);
    if (!in_window) return;
    if (p.out_ids) p.out_ids[pix] = h.id;
    if (p.out_t) p.out_t[pix] = h.t;
}

// Self-check of a search structure: every pixel's camera ray, and `verify_bounces` generations of the rays Shade() makes of
// the hits (sample index first_sample), are answered twice -- by the shader's own search, the loop over EVERY triangle record
// in index order (pathtracing.fs:129; not the compacted array), and by the structure a dispatch would use (KIND as in
// pt_render_kernel; 1 = the compacted loop). verify_out: [0] queries compared, [1] queries whose triangle differs, [2] queries
// whose distance differs in any bit. The path continues along the loop's answer.
template <int KIND, bool MATERIALS>
__global__ __launch_bounds__(PT_TILE_THREADS) void pt_verify_kernel(const PtKernelParams p) {
    __shared__ uint32_t s_stack[(KIND == 2 || KIND == 5 || KIND == 6) ? PT_TILE_WAVES : 1][(KIND == 2 || KIND == 5 || KIND == 6) ? PT_BVH_STACK : 1];
    __shared__ uint32_t s_lane_stack[(KIND == 5 || KIND == 6) ? PT_TILE_WAVES : 1][(KIND == 5 || KIND == 6) ? PT_LANE_STACK * 64 : 1];
    int x, y;
    bool live = pixel_of_lane(p, (int)blockIdx.x, x, y);
    Rng rng;
    rng.fcx = (float)x + 0.5f;
    rng.fcy = (float)y + 0.5f;
    rng.samples = (float)p.first_sample;
    rng.sx = 0.0f;
    v3 ro = V(p.ro[0], p.ro[1], p.ro[2]);
    v3 rd = primary_direction(p, rng.fcx, rng.fcy);
    v3 energy = V(1.0f, 1.0f, 1.0f), emitted = V(0.f, 0.f, 0.f);
    unsigned compared = 0, bad_id = 0, bad_t = 0, unused = 0, nv = 0, tt = 0;
    for (int g = 0; g <= p.verify_bounces; ++g) {
        if (live) {
            Hit a, b;
            v3 gn = V(0.f, 0.f, 0.f);
            a = MATERIALS ? intersect_mesh_from(p.tris_abs, p.num_triangles, ro, rd, unused) : intersect_mesh(p.verify_tris, p.num_triangles, rd, unused);
            if (KIND == 1) b = intersect_mesh(p.tris, live_triangles(p), rd, unused);
            else if (KIND == 2) b = intersect_bvh<MATERIALS>(p, ro, rd, s_stack[threadIdx.x >> 6], nv, tt);
            else if (KIND == 5) {
                bool overflow;
                b = intersect_bvh_lanes(p, ro, rd, s_lane_stack[threadIdx.x >> 6], overflow, tt);
                if (overflow) b = intersect_bvh<true>(p, ro, rd, s_stack[threadIdx.x >> 6], nv, tt);
            } else if (KIND == 6) {  // the wide hierarchy, walked in rounds as pt_render_decoupled_kernel walks it
                LaneWalk w;
                walk_begin(p, w, ro, rd);
                uint32_t* my = s_lane_stack[threadIdx.x >> 6] + (threadIdx.x & 63);
                int r;
                do { r = walk_round(p, w, ro, rd, my, true, nv, tt); } while (r == 0);
                b = w.h;
                if (r == 2) b = intersect_bvh<true>(p, ro, rd, s_stack[threadIdx.x >> 6], nv, tt);
            } else b = intersect_grid(p, rd, true, tt, gn, p.grid_occ != nullptr);  // (the block bits are checked whenever they exist)
            ++compared;
            bad_id += a.id != b.id ? 1u : 0u;
            bad_t += __float_as_uint(a.t) != __float_as_uint(b.t) ? 1u : 0u;
            if (a.id < 0) {
                live = false;
            } else {
                const v3 N = mesh_normal(p, a);
                if (MATERIALS) {
                    const v3 P = ro + rd * a.t;
                    shade_materials(rng, p.materials, p.material_ids[a.id], energy, emitted, ro, rd, P, N, p.shading_mode == 2);
                    live = !(energy.x == 0.0f && energy.y == 0.0f && energy.z == 0.0f);
                } else {
                    shade(rng, energy.x, rd, N);
                    live = energy.x != 0.0f;
                }
            }
        }
    }
    const unsigned long long c = wave_sum(compared), bi = wave_sum(bad_id), bt = wave_sum(bad_t);
    if ((threadIdx.x & 63) == 0 && c) {
        atomicAdd(&p.verify_out[0], c);
        if (bi) atomicAdd(&p.verify_out[1], bi);
        if (bt) atomicAdd(&p.verify_out[2], bt);
    }
}

// The short reciprocal, square-root and log2-quotient sequences against the compiler's IEEE division and square root, for
// every float bit pattern: bad[0] reciprocal, bad[1] square root, bad[2] quotient and the restricted-domain forms (rand01's
// tail, pow of a value in [0, 1), exp2 without clamps) and the cube-map instructions against their definition -- inputs whose results differ in any bit (NaN payloads included). All
// three must stay 0.
__global__ __launch_bounds__(256) void pt_arith_selftest_kernel(unsigned long long* __restrict__ bad) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned b0 = 0, b1 = 0, b2 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        b0 += __float_as_uint(ieee_rcp(x)) != __float_as_uint(1.0f / x);
        b1 += __float_as_uint(ieee_sqrt(x)) != __float_as_uint(__builtin_sqrtf(x));
        if (i < (1ull << 23)) {
            const float m = __uint_as_float((uint32_t)i | 0x3f800000u);
            b2 += __float_as_uint(log2_quotient(m)) != __float_as_uint((m - 1.0f) / (m + 1.0f));
        }
        // the forms that leave out what their callers' arguments cannot reach, against the full forms:
        // the tail of rand01() for every value the inner expression can take,
        const float r = fract1(52.9829189f * fract1(x));
        b2 += __float_as_uint(r - __builtin_floorf(r)) != __float_as_uint(r);
        // pow of a rand01() result (in [0, 1), or NaN) with the two exponents the shading code uses,
        if (x != x || (x >= 0.0f && x < 1.0f)) {
            b2 += __float_as_uint(lp_pow_unit(x, 1.0f / 1025.0f)) != __float_as_uint(lp_pow(x, 1.0f / 1025.0f));
            b2 += __float_as_uint(lp_pow_unit(x, 0.5f)) != __float_as_uint(lp_pow(x, 0.5f));
        }
        // the cube-map instructions against the selections written out, on directions made of this bit pattern, of two
        // scrambled ones and of their ties (equal magnitudes, zeros),
        {
            const float y = __uint_as_float((uint32_t)i * 2654435761u), z = __uint_as_float(((uint32_t)i ^ 0x5bd1e995u) * 2246822519u);
            const v3 dirs[4] = {V(x, y, z), V(x, -x, y), V(y, x, x), V(x, 0.0f, -x)};
            for (int k = 0; k < 4; ++k) {
                const CubeCoords a = cube_coords(dirs[k]), e = cube_coords_expected(dirs[k]);
                // (left out: NaN and infinite components, denormal ones -- flushed by the arithmetic here, not by a selection --
                // and the all-zero vector, whose face is the sign of a zero: none of them is a direction that can hit anything,
                // and a query with a zero or NaN major component looks nothing up)
                auto odd = [](float c) { const uint32_t b = __float_as_uint(c) & 0x7fffffffu; return b >= 0x7f800000u || (b != 0u && b < 0x00800000u); };
                const bool zero = dirs[k].x == 0.0f && dirs[k].y == 0.0f && dirs[k].z == 0.0f;
                if (!zero && !odd(dirs[k].x) && !odd(dirs[k].y) && !odd(dirs[k].z))
                    b2 += __float_as_uint(a.id) != __float_as_uint(e.id) || __float_as_uint(a.sc) != __float_as_uint(e.sc) ||
                          __float_as_uint(a.tc) != __float_as_uint(e.tc) || __float_as_uint(a.ma) != __float_as_uint(e.ma);
            }
        }
        // and exp2 without its clamps over the exponents of the Fresnel term: -8.656170 * c for every c in [0, 1]
        if (x >= 0.0f && x <= 1.0f) {
            const float a = -8.656170f * x;
            b2 += __float_as_uint(lp_exp2<false, false>(a)) != __float_as_uint(lp_exp2(a));
        }
    }
    if (b0) atomicAdd(&bad[0], (unsigned long long)b0);
    if (b1) atomicAdd(&bad[1], (unsigned long long)b1);
    if (b2) atomicAdd(&bad[2], (unsigned long long)b2);
}

// One thread per triangle: gathers the three vertices through the index buffer and writes the SoA records.
__global__ __launch_bounds__(256) void pt_prepare_triangles_kernel(const float* __restrict__ vertices,
                                                                    const uint32_t* __restrict__ indices, int n,
                                                                    float rox, float roy, float roz,
                                                                    float* __restrict__ tris, float* __restrict__ vnormals,
                                                                    float* __restrict__ gnormals, float* __restrict__ cull,
                                                                    float* __restrict__ tris_abs, float* __restrict__ tris_grid,
                                                                    int write_static) {
    // write_static = 0: only what depends on the ray origin (tris, tris_grid, cull if given) -- a camera move;
    // 1: also the arrays that depend on the mesh alone (tris_abs, vnormals, gnormals) -- after an upload
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* pa = vertices + 6 * (size_t)indices[3 * i];
    const float* pb = vertices + 6 * (size_t)indices[3 * i + 1];
    const float* pc = vertices + 6 * (size_t)indices[3 * i + 2];
    const v3 a = V(pa[0], pa[1], pa[2]), b = V(pb[0], pb[1], pb[2]), c = V(pc[0], pc[1], pc[2]);
    const v3 e1 = b - a, e2 = c - a;
    const v3 ap = V(rox, roy, roz) - a;
    const v3 r = cross3(ap, e1);
    const float e2r = dot3(e2, r);
    const v3 g = normalize3(cross3(e1, e2));
    // records are 16 floats, 64-byte aligned: four 16-byte stores each
    const float4 q0 = make_float4(e1.x, e1.y, e1.z, e2.x), q1 = make_float4(e2.y, e2.z, ap.x, ap.y), q2 = make_float4(ap.z, r.x, r.y, r.z);
    float4* t = reinterpret_cast<float4*>(tris + (size_t)i * PT_TRI_FLOATS);
    t[0] = q0; t[1] = q1; t[2] = q2;
    t[3] = make_float4(e2r, 0.f, 0.f, __uint_as_float((uint32_t)i));
    // record of the direction-grid query: the same twelve words and e2r, then the geometric normal instead of the index
    // (the query knows the index from the cell), so a hit needs no further fetch
    float4* tg = reinterpret_cast<float4*>(tris_grid + (size_t)i * PT_TRI_FLOATS);
    tg[0] = q0; tg[1] = q1; tg[2] = q2;
    tg[3] = make_float4(e2r, g.x, g.y, g.z);
    if (write_static) {
        // origin-independent record for rays with their own origins: a, e1, e2
        float4* ta = reinterpret_cast<float4*>(tris_abs + (size_t)i * PT_TRI_FLOATS);
        ta[0] = make_float4(a.x, a.y, a.z, e1.x);
        ta[1] = make_float4(e1.y, e1.z, e2.x, e2.y);
        ta[2] = make_float4(e2.z, 0.f, 0.f, 0.f);
        ta[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        float* vn = vnormals + (size_t)i * 9;
        vn[0] = pa[3]; vn[1] = pa[4]; vn[2] = pa[5];
        vn[3] = pb[3]; vn[4] = pb[4]; vn[5] = pb[5];
        vn[6] = pc[3]; vn[7] = pc[4]; vn[8] = pc[5];
        gnormals[3 * (size_t)i] = g.x; gnormals[3 * (size_t)i + 1] = g.y; gnormals[3 * (size_t)i + 2] = g.z;
    }
    if (!cull) return;

    // Cull record: four unit normals of planes through the ray origin. A direction can only pass the exact test
    // when it lies on the inner side of the three edge planes (origin, two vertices) and sees the front of the
    // triangle; a whole bundle of directions is skipped when its bounding cone lies outside one of them. Any plane
    // that cannot be oriented reliably in fp32 (origin almost in the triangle's plane, sliver seen edge-on) is
    // stored as zero, which never rejects. This arithmetic only decides what is SKIPPED; it is not part of the
    // reference's numerics.
    // Written with the short edge vectors so that small, distant triangles do not lose the planes to cancellation:
    // A x B = A x e1, B x C = B x (e2 - e1), C x A = e2 x C, and the orientation A.(B x C) = A.(e1 x e2).
    const v3 A = a - V(rox, roy, roz), B = b - V(rox, roy, roz), C = c - V(rox, roy, roz);
    const v3 e12 = e2 - e1;
    auto len = [](v3 x) { return __builtin_sqrtf(dot3(x, x)); };
    const v3 nrm = cross3(e1, e2);
    const float T = dot3(A, nrm);
    const bool oriented = __builtin_fabsf(T) > 1.0e-4f * len(A) * len(nrm);
    const float sgn = T > 0.f ? 1.f : -1.f;
    auto unit_or_zero = [&len](v3 n, float scale, bool ok, float sign) -> v3 {
        const float l = len(n);
        if (!ok || !(l > 1.0e-4f * scale)) return V(0.f, 0.f, 0.f);
        return n * (sign / l);
    };
    const v3 nab = unit_or_zero(cross3(A, e1), len(A) * len(e1), oriented, sgn);
    const v3 nbc = unit_or_zero(cross3(B, e12), len(B) * len(e12), oriented, sgn);
    const v3 nca = unit_or_zero(cross3(e2, C), len(C) * len(e2), oriented, sgn);
    const v3 front = unit_or_zero(cross3(e2, e1), len(e1) * len(e2), true, 1.f);  // det = dot(rd, e2 x e1)
    float* cr = cull + (size_t)i * PT_CULL_FLOATS;
    cr[0] = nab.x; cr[1] = nab.y; cr[2] = nab.z; cr[3] = nbc.x;
    cr[4] = nbc.y; cr[5] = nbc.z; cr[6] = nca.x; cr[7] = nca.y;
    cr[8] = nca.z; cr[9] = front.x; cr[10] = front.y; cr[11] = front.z;
}

// Per-sphere records for intersect_world: [to.xyz, C] [centre.xyz, 1/radius] with to = ro - centre, C = dot(to,to) - r*r.
__global__ void pt_prepare_spheres_kernel(const float* __restrict__ spheres, float rox, float roy, float roz, float* __restrict__ recs) {
    const int obj = threadIdx.x;
    if (obj >= 10) return;
    const v3 center = V(spheres[4 * obj], spheres[4 * obj + 1], spheres[4 * obj + 2]);
    const float radius = spheres[4 * obj + 3];
    const v3 to = V(rox, roy, roz) - center;
    float* r = recs + 8 * obj;
    r[0] = to.x; r[1] = to.y; r[2] = to.z;
    r[3] = dot3(to, to) - radius * radius;
    r[4] = center.x; r[5] = center.y; r[6] = center.z;
    r[7] = 1.0f / radius;
}

// Every ray starts at the camera (the shader never moves `ro`), so a hit's distance is e2r / det with e2r =
// dot(e2, cross(ro - a, e1)) fixed per triangle; det >= 1e-4 is positive, so a triangle with e2r <= 0 (the camera is
// behind its plane, or in it) yields t <= 0 for every ray and fails the shader's `temp > 0.0` (fs:156) always. Such
// triangles are left out of the array the all-triangles loop walks: same results, fewer tests (closed meshes lose about
// half their triangles). The ascending order is kept (the loop's strict '<' relies on it). Three launches: every workgroup
// counts the live records of its PT_COMPACT_CHUNK consecutive records; one workgroup turns the counts into offsets; every
// workgroup copies its live records to where they belong.
constexpr int PT_COMPACT_THREADS = 256, PT_COMPACT_ROUNDS = 4, PT_COMPACT_CHUNK = PT_COMPACT_THREADS * PT_COMPACT_ROUNDS;

__device__ __forceinline__ bool record_is_live(const float4* __restrict__ tris, int i, int n) {
    return i < n && tris[4 * (size_t)i + 3].x > 0.0f;
}

__global__ __launch_bounds__(PT_COMPACT_THREADS) void pt_compact_count_kernel(const float4* __restrict__ tris, int n, uint32_t* __restrict__ block_count) {
    __shared__ uint32_t wave_total[PT_COMPACT_THREADS / 64];
    const int base = (int)blockIdx.x * PT_COMPACT_CHUNK;
    uint32_t mine = 0;
    for (int r = 0; r < PT_COMPACT_ROUNDS; ++r) mine += record_is_live(tris, base + r * PT_COMPACT_THREADS + (int)threadIdx.x, n) ? 1u : 0u;
    const uint32_t w = (uint32_t)wave_sum((unsigned long long)mine);
    if ((threadIdx.x & 63) == 0) wave_total[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
}

// counts -> exclusive offsets in place; *live_count <- the total; the PT_TRI_PAD all-zero records the loop reads past the end
__global__ __launch_bounds__(1024) void pt_compact_offsets_kernel(uint32_t* __restrict__ block_count, int blocks, float4* __restrict__ live,
                                                                 uint32_t* __restrict__ live_count) {
    __shared__ uint32_t part[1024];
    // each thread owns a contiguous run of the counts
    const int per = (blocks + 1023) / 1024, begin = min(blocks, (int)threadIdx.x * per), end = min(blocks, begin + per);
    uint32_t mine = 0;
    for (int i = begin; i < end; ++i) mine += block_count[i];
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int step = 1; step < 1024; step <<= 1) {  // inclusive scan
        const uint32_t add = threadIdx.x >= (unsigned)step ? part[threadIdx.x - step] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    const uint32_t total = part[1023];
    uint32_t running = part[threadIdx.x] - mine;
    for (int i = begin; i < end; ++i) {
        const uint32_t c = block_count[i];
        block_count[i] = running;
        running += c;
    }
    if (threadIdx.x < PT_TRI_PAD)
        for (int q = 0; q < 4; ++q) live[4 * ((size_t)total + threadIdx.x) + q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x == 0) *live_count = total;
}

__global__ __launch_bounds__(PT_COMPACT_THREADS) void pt_compact_scatter_kernel(const float4* __restrict__ tris, int n, const uint32_t* __restrict__ block_offset,
                                                                               float4* __restrict__ live) {
    __shared__ uint32_t wave_total[PT_COMPACT_THREADS / 64];
    const int base = (int)blockIdx.x * PT_COMPACT_CHUNK;
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    uint32_t out = block_offset[blockIdx.x];
    for (int r = 0; r < PT_COMPACT_ROUNDS; ++r) {   // a round = 256 consecutive records, one per thread
        const int i = base + r * PT_COMPACT_THREADS + (int)threadIdx.x;
        const bool keep = record_is_live(tris, i, n);
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
        if (lane == 0) wave_total[wave] = (uint32_t)__builtin_popcountll(mask);
        __syncthreads();
        uint32_t before = (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
        uint32_t round_total = 0;
        for (int w = 0; w < PT_COMPACT_THREADS / 64; ++w) {
            if (w < wave) before += wave_total[w];
            round_total += wave_total[w];
        }
        if (keep)
            for (int q = 0; q < 4; ++q) live[4 * ((size_t)out + before) + q] = tris[4 * (size_t)i + q];
        out += round_total;
        __syncthreads();
    }
}

// Present pass: the reference's copy.fs (pow(rgb, 0.454545), copy.fs:18) followed by the float -> 8-bit conversion
// of the render target it is drawn into: clamp to [0,1], x255, round to nearest even; alpha 1 -> 255.
__global__ __launch_bounds__(256) void pt_present_kernel(const float4* __restrict__ img, size_t n, uint32_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 c = img[i];
        const float ch[3] = {c.x, c.y, c.z};
        uint32_t packed = 0xff000000u;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = clamp01(lp_pow(ch[k], 0.454545f));
            packed |= (uint32_t)__builtin_rintf(v * 255.0f) << (8 * k);
        }
        out[i] = packed;
    }
}

__global__ __launch_bounds__(256) void pt_fill_rgba_kernel(float4* img, size_t n, float4 value) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) img[i] = value;
}

// Second half of a dispatch for the tiles whose samples were rendered by several blocks: folds the samples a pixel has
// in the scratch slot into the image in sample order, with the arithmetic of the render kernel (fs:412-415), so the
// result does not depend on how the samples were distributed.
__global__ __launch_bounds__(PT_TILE_THREADS) void pt_fold_samples_kernel(const PtKernelParams p) {
    if (blockIdx.x >= p.sched_counts[1]) return;
    int x, y;
    if (!pixel_of_lane(p, (int)p.split_tiles[blockIdx.x], x, y)) return;
    const size_t pix = (size_t)y * p.width + x;
    const float4 prev = reinterpret_cast<const float4*>(p.src)[pix];
    float mr = prev.x, mg = prev.y, mb = prev.z;
    int sample = p.first_sample;
    auto fold = [&](float sr, float sg, float sb) {
        if (p.accum_mode == 0) {
            const float a = 1.0f / (float)sample;
            mr = mr + a * (sr - mr);
            mg = mg + a * (sg - mg);
            mb = mb + a * (sb - mb);
        } else {
            mr += sr; mg += sg; mb += sb;
        }
        sample += p.sample_stride;
    };
    // the folds form one dependent chain; the loads do not, so they are issued eight at a time ahead of it
    const size_t first = (size_t)blockIdx.x * p.num_samples * PT_TILE_THREADS + threadIdx.x;
    if (p.shading_mode != 0) {
        const float4* __restrict__ scratch = reinterpret_cast<const float4*>(p.sample_scratch) + first;
        int k = 0;
        for (; k + 8 <= p.num_samples; k += 8) {
            float4 s[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = scratch[(size_t)(k + j) * PT_TILE_THREADS];
#pragma unroll
            for (int j = 0; j < 8; ++j) fold(s[j].x, s[j].y, s[j].z);
        }
        for (; k < p.num_samples; ++k) { const float4 v = scratch[(size_t)k * PT_TILE_THREADS]; fold(v.x, v.y, v.z); }
    } else {
        // (throughput, final direction's y) -> fs:408-410 as in the render kernel
        auto fold_sky = [&](float2 v) {
            const float t = 0.5f * (v.y + 1.0f);
            const float omt = 1.0f - t;
            fold(v.x * (omt + 0.5f * t), v.x * (omt + 0.7f * t), v.x * (omt + 1.0f * t));
        };
        const float2* __restrict__ scratch = reinterpret_cast<const float2*>(p.sample_scratch) + first;
        int k = 0;
        for (; k + 8 <= p.num_samples; k += 8) {
            float2 s[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = scratch[(size_t)(k + j) * PT_TILE_THREADS];
#pragma unroll
            for (int j = 0; j < 8; ++j) fold_sky(s[j]);
        }
        for (; k < p.num_samples; ++k) fold_sky(scratch[(size_t)k * PT_TILE_THREADS]);
    }
    const float alpha = p.accum_mode == 0 ? 1.0f : prev.w + (float)p.num_samples;
    reinterpret_cast<float4*>(p.dst)[pix] = make_float4(mr, mg, mb, alpha);
}

// Builds the work list of the next dispatch from the per-tile costs the previous one measured (loop trips per tile).
// granule = total cost / max_split + 1: a tile above it is split into ceil(cost / granule) items of equal sample counts
// (at most num_samples), so the extra items and the split tiles both stay below max_split. Items are placed most
// expensive first by half-octave bucket of their cost: a counting sort over PT_SCHED_BLOCKS workgroups in four small
// launches (total, per-workgroup bucket counts, offsets, scatter); positions inside a workgroup's share of a bucket are
// handed out atomically, so the list is always a complete cover whatever the order. The costs are only read (the caller
// may run the scheduler again with another limit before clearing them).
// tmp: [0..1] total cost (64 bit), [3] the most expensive tile's cost, [4 + 64 * block + bucket] that workgroup's count, then start, in the bucket.
constexpr int PT_SCHED_BLOCKS = 64, PT_SCHED_THREADS = 256;

// Bucket of an item of cost c (higher = more expensive = earlier in the list): the 64 buckets cover the eight octaves below the
// most expensive tile's cost, eight to an octave; cheaper items share bucket 0. (Two buckets per octave over all 32 octaves, as
// before, put most tiles of a view into one or two buckets whenever their costs cluster -- a wave whose pixels all end after one
// query costs 8 x samples, whatever the samples --, and inside a bucket the order is raster order: the 1M triangles at 4K, 4,096
// samples per dispatch, took 70 ms per dispatch with such a list and 57 ms with one that happened to cut the cluster in two.)
#ifndef PT_SCHED_SUB_LOG
#define PT_SCHED_SUB_LOG 3
#endif
#define PT_SCHED_SUB (1 << PT_SCHED_SUB_LOG)   // buckets per octave
__device__ __forceinline__ int sched_bucket(uint32_t c, int sort, uint32_t cmax) {
    if (!sort || c == 0u) return 0;
    const int octave = __builtin_clz(c) - __builtin_clz(cmax | 1u);   // octaves below the top one (never negative: c <= cmax)
    if (octave < 0) return 63;
    if (octave >= 64 / PT_SCHED_SUB) return 0;
    const int lg = 31 - __builtin_clz(c);
    const int sub = (int)((lg >= PT_SCHED_SUB_LOG ? c >> (lg - PT_SCHED_SUB_LOG) : c << (PT_SCHED_SUB_LOG - lg)) & (uint32_t)(PT_SCHED_SUB - 1));
    return 63 - (octave * PT_SCHED_SUB + (PT_SCHED_SUB - 1 - sub));
}
// pieces of a tile and samples per piece
__device__ __forceinline__ int sched_split(uint32_t c, unsigned long long granule, int num_samples, int& per_piece) {
    if ((unsigned long long)c <= granule) {  // the usual tile: one item (and none of the divisions below)
        per_piece = num_samples;
        return 1;
    }
    unsigned long long k = ((unsigned long long)c + granule - 1) / granule;
    if (k < 1) k = 1;
    if (k > (unsigned long long)num_samples) k = (unsigned long long)num_samples;
    per_piece = (num_samples + (int)k - 1) / (int)k;
    return (num_samples + per_piece - 1) / per_piece;
}
__device__ __forceinline__ unsigned long long sched_granule(const uint32_t* tmp, int max_split) {
    const unsigned long long total = *reinterpret_cast<const unsigned long long*>(tmp);
    return max_split > 0 ? total / (unsigned long long)max_split + 1ull : ~0ull;
}
// the tiles of workgroup b: one contiguous share
__device__ __forceinline__ void sched_share(int n, int& begin, int& end) {
    const int per = (n + PT_SCHED_BLOCKS - 1) / PT_SCHED_BLOCKS;
    begin = min(n, (int)blockIdx.x * per);
    end = begin + per < n ? begin + per : n;
}
// Groups: with `groups` = 8 the list is dealt to the eight XCDs of the device. Workgroups go to the XCDs round-robin by block
// index, every XCD has an L2 of its own, and the records a tile's rays fetch are those of its part of the scene: so the
// tiles are cut, in raster order, into `groups` runs of equal cost (image bands), group g's items stand at the list positions
// g, g + groups, g + 2 groups, ... (most expensive first within the group), and the XCD that takes those blocks keeps reading
// the same part of the scene. Groups differ in length; the unused positions of the shorter ones hold PT_NO_TILE.
// tmp: [0..1] total cost (64 bit), [2] items of the longest group, [3] the most expensive tile's cost, [4 + g] items of group g, [16 + 2 w] cost of workgroup w's
// share of the tiles (64 bit), [PT_SCHED_COUNTS + ((w * 8 + g) * 64 + bucket)] that workgroup's count, then start, in the bucket.
constexpr int PT_SCHED_SHARES = 16, PT_SCHED_COUNTS = PT_SCHED_SHARES + 2 * PT_SCHED_BLOCKS, PT_SCHED_MAX_GROUPS = 8;
static_assert(PT_SCHED_COUNTS + PT_SCHED_BLOCKS * PT_SCHED_MAX_GROUPS * 64 == PT_SCHED_TMP_WORDS, "scratch of the scheduler");

__device__ __forceinline__ int sched_counter(int wg, int group, int bucket) { return PT_SCHED_COUNTS + ((wg * PT_SCHED_MAX_GROUPS + group) * 64 + bucket); }

// Exclusive prefix sum of `v` over the workgroup's PT_SCHED_THREADS threads (all of them call); total <- the sum.
__device__ __forceinline__ unsigned long long sched_block_scan(unsigned long long v, unsigned long long* wave_total, unsigned long long& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    __syncthreads();   // (the previous call's totals have been read)
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    unsigned long long before = 0;
    total = 0;
    for (int w = 0; w < PT_SCHED_THREADS / 64; ++w) {
        if (w < wave) before += wave_total[w];
        total += wave_total[w];
    }
    return before + incl - v;
}

// Calls body(tile, cost, group) for every tile of this workgroup's share, in chunks of PT_SCHED_THREADS consecutive tiles.
template <typename Body>
__device__ __forceinline__ void sched_for_tiles(const uint32_t* __restrict__ cost, int n, int groups, const uint32_t* __restrict__ tmp, Body body) {
    __shared__ unsigned long long wave_total[PT_SCHED_THREADS / 64];
    int begin, end;
    sched_share(n, begin, end);
    const unsigned long long total = *reinterpret_cast<const unsigned long long*>(tmp);
    unsigned long long running = 0;
    if (groups > 1)
        for (int w = 0; w < (int)blockIdx.x; ++w) running += *reinterpret_cast<const unsigned long long*>(tmp + PT_SCHED_SHARES + 2 * w);
    for (int base = begin; base < end; base += PT_SCHED_THREADS) {
        const int i = base + (int)threadIdx.x;
        const uint32_t c = i < end ? cost[i] : 0u;
        int group = 0;
        if (groups > 1) {   // (uniform)
            unsigned long long chunk;
            const unsigned long long before = running + sched_block_scan(c, wave_total, chunk);
            running += chunk;
            group = total ? (int)min((unsigned long long)(groups - 1), before * (unsigned long long)groups / total) : 0;
        }
        if (i < end) body(i, c, group);
    }
}

// total cost, and the cost of every workgroup's share
__global__ __launch_bounds__(PT_SCHED_THREADS) void pt_sched_total_kernel(const uint32_t* __restrict__ cost, int n, uint32_t* __restrict__ tmp) {
    __shared__ unsigned long long part[PT_SCHED_THREADS / 64];
    int begin, end;
    sched_share(n, begin, end);
    unsigned long long mine = 0;
    uint32_t highest = 0;
    for (int i = begin + (int)threadIdx.x; i < end; i += PT_SCHED_THREADS) { mine += cost[i]; highest = max(highest, cost[i]); }
    if (highest) atomicMax(&tmp[3], highest);   // the most expensive tile's cost (the buckets are laid out below it)
    mine = wave_sum(mine);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long share = 0;
        for (int w = 0; w < PT_SCHED_THREADS / 64; ++w) share += part[w];
        *reinterpret_cast<unsigned long long*>(tmp + PT_SCHED_SHARES + 2 * blockIdx.x) = share;
        if (share) atomicAdd(reinterpret_cast<unsigned long long*>(tmp), share);
    }
}

__global__ __launch_bounds__(PT_SCHED_THREADS) void pt_sched_count_kernel(const uint32_t* __restrict__ cost, int n, int num_samples, int max_split,
                                                                          int sort, int groups, uint32_t* __restrict__ tmp) {
    __shared__ uint32_t hist[PT_SCHED_MAX_GROUPS][64];
    for (int k = threadIdx.x; k < PT_SCHED_MAX_GROUPS * 64; k += PT_SCHED_THREADS) hist[k >> 6][k & 63] = 0;
    __syncthreads();
    const unsigned long long granule = sched_granule(tmp, max_split);
    const uint32_t cmax = tmp[3];
    sched_for_tiles(cost, n, groups, tmp, [&](int, uint32_t c, int group) {
        int per_piece;
        const int k = sched_split(c, granule, num_samples, per_piece);
        atomicAdd(&hist[group][sched_bucket(c / (uint32_t)k, sort, cmax)], (uint32_t)k);
    });
    __syncthreads();
    for (int k = threadIdx.x; k < groups * 64; k += PT_SCHED_THREADS) tmp[sched_counter(blockIdx.x, k >> 6, k & 63)] = hist[k >> 6][k & 63];
}

// One workgroup of 64 threads, thread = bucket: counts -> rank of every workgroup's first item of the bucket within its group
// (most expensive bucket first); counts[0] <- list positions in use (groups x the longest group), counts[1] <- 0 (the scatter
// counts the split tiles there).
__global__ __launch_bounds__(64) void pt_sched_offsets_kernel(uint32_t* __restrict__ tmp, uint32_t* __restrict__ counts, int groups) {
    __shared__ uint32_t bucket_total[PT_SCHED_MAX_GROUPS][64], bucket_base[PT_SCHED_MAX_GROUPS][64], group_items[PT_SCHED_MAX_GROUPS];
    const int b = threadIdx.x;
    for (int g = 0; g < groups; ++g) {
        uint32_t running = 0;
        for (int blk = 0; blk < PT_SCHED_BLOCKS; ++blk) {
            const uint32_t v = tmp[sched_counter(blk, g, b)];
            tmp[sched_counter(blk, g, b)] = running;
            running += v;
        }
        bucket_total[g][b] = running;
    }
    __syncthreads();
    if (b < groups) {
        uint32_t acc = 0;
        for (int k = 63; k >= 0; --k) { bucket_base[b][k] = acc; acc += bucket_total[b][k]; }
        group_items[b] = acc;
        tmp[4 + b] = acc;
    }
    __syncthreads();
    if (b == 0) {
        uint32_t longest = 0;
        for (int g = 0; g < groups; ++g) longest = max(longest, group_items[g]);
        tmp[2] = longest;
        counts[0] = longest * (uint32_t)groups;
        counts[1] = 0;
    }
    for (int g = 0; g < groups; ++g) {
        const uint32_t base = bucket_base[g][b];
        for (int blk = 0; blk < PT_SCHED_BLOCKS; ++blk) tmp[sched_counter(blk, g, b)] += base;
    }
}

__global__ __launch_bounds__(PT_SCHED_THREADS) void pt_sched_scatter_kernel(const uint32_t* __restrict__ cost, int n, int num_samples, int max_split,
                                                                            int sort, int groups, const uint32_t* __restrict__ tmp, PtWorkItem* __restrict__ items,
                                                                            uint32_t* __restrict__ counts, uint32_t* __restrict__ split_tiles) {
    __shared__ uint32_t cursor[PT_SCHED_MAX_GROUPS][64];
    for (int k = threadIdx.x; k < groups * 64; k += PT_SCHED_THREADS) cursor[k >> 6][k & 63] = tmp[sched_counter(blockIdx.x, k >> 6, k & 63)];
    __syncthreads();
    if (groups > 1) {
        // positions past the end of a group that is shorter than the longest one: nothing to render there
        const uint32_t positions = tmp[2] * (uint32_t)groups, per = (positions + PT_SCHED_BLOCKS - 1) / PT_SCHED_BLOCKS;
        const uint32_t first = blockIdx.x * per, last = min(positions, first + per);
        for (uint32_t q = first + threadIdx.x; q < last; q += PT_SCHED_THREADS)
            if (q / (uint32_t)groups >= tmp[4 + q % (uint32_t)groups]) {
                PtWorkItem none;
                none.tile = PT_NO_TILE; none.sample_begin = 0; none.sample_count = 0; none.scratch_slot = -1;
                items[q] = none;
            }
    }
    const unsigned long long granule = sched_granule(tmp, max_split);
    const uint32_t cmax = tmp[3];
    sched_for_tiles(cost, n, groups, tmp, [&](int i, uint32_t c, int group) {
        int per_piece;
        const int k = sched_split(c, granule, num_samples, per_piece);
        const uint32_t first = atomicAdd(&cursor[group][sched_bucket(c / (uint32_t)k, sort, cmax)], (uint32_t)k);
        int slot = -1;
        if (k > 1) {
            slot = (int)atomicAdd(&counts[1], 1u);
            split_tiles[slot] = (uint32_t)i;
        }
        for (int j = 0; j < k; ++j) {
            PtWorkItem it;
            it.tile = (uint32_t)i;
            it.sample_begin = (uint32_t)(j * per_piece);
            it.sample_count = (uint32_t)(num_samples - j * per_piece < per_piece ? num_samples - j * per_piece : per_piece);
            it.scratch_slot = slot;
            items[(first + (uint32_t)j) * (uint32_t)groups + (uint32_t)group] = it;
        }
    });
}

// ---- sparse views: the pixels that see the mesh, gathered -------------------------------------------------------------------
// In a view of small objects most 8x8 blocks that contain a hit pixel also contain sky pixels, whose lanes have nothing to do
// while the wave works (1M triangles at 4K: a third of the lane slots). All camera rays start at the camera, so whether a
// pixel sees the mesh is one query of the direction grid per pixel and camera position: the hit pixels are listed block after
// block (neighbours stay neighbours), 64 consecutive entries are a wave's tile, and the other pixels -- every sample the sky
// colour of the camera ray -- are folded by pt_sky_fold_kernel with the render kernels' own operations.
__global__ __launch_bounds__(64) void pt_pixel_mask_kernel(const PtKernelParams p, unsigned long long* __restrict__ mask, uint32_t* __restrict__ counts) {
    int x, y;
    const bool mine = pixel_of_lane(p, (int)blockIdx.x, x, y);
    bool hit = false;
    if (mine) {
        unsigned tests = 0;
        v3 n = V(0.f, 0.f, 0.f);
        hit = intersect_grid(p, primary_direction(p, (float)x + 0.5f, (float)y + 0.5f), true, tests, n, p.grid_occ != nullptr).id >= 0;
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
    if (threadIdx.x == 0) {
        mask[blockIdx.x] = m;
        counts[blockIdx.x] = (uint32_t)__builtin_popcountll(m);
    }
}

// offsets <- exclusive prefix sums of counts[0..n), total[0] <- their sum, total[1] <- how many of them are not zero (one
// workgroup: a few hundred thousand blocks at most)
__global__ __launch_bounds__(1024) void pt_pixel_offsets_kernel(const uint32_t* __restrict__ counts, int n, uint32_t* __restrict__ offsets, uint32_t* __restrict__ total) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t nonzero;
    if (threadIdx.x == 0) nonzero = 0u;
    __syncthreads();
    const int per = (n + 1023) / 1024;
    const int begin = (int)threadIdx.x * per, end = begin + per < n ? begin + per : n;
    uint32_t sum = 0, some = 0;
    for (int i = begin; i < end; ++i) {
        sum += counts[i];
        some += counts[i] != 0u;
    }
    part[threadIdx.x] = sum;
    if (some) atomicAdd(&nonzero, some);
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (int i = begin; i < end; ++i) {
        offsets[i] = run;
        run += counts[i];
    }
    if (threadIdx.x == 1023) {
        total[0] = part[1023];
        total[1] = nonzero;
    }
}

__global__ __launch_bounds__(64) void pt_pixel_scatter_kernel(const PtKernelParams p, const unsigned long long* __restrict__ mask,
                                                              const uint32_t* __restrict__ offsets, uint32_t* __restrict__ list) {
    int x, y;
    (void)pixel_of_lane(p, (int)blockIdx.x, x, y);
    const unsigned long long m = mask[blockIdx.x];
    const unsigned lane = threadIdx.x;
    if ((m >> lane) & 1ull) list[offsets[blockIdx.x] + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = ((uint32_t)y << 16) | (uint32_t)x;
}

__global__ __launch_bounds__(64) void pt_sky_fold_kernel(const PtKernelParams p, const unsigned long long* __restrict__ mask) {
    const unsigned long long m = mask[blockIdx.x];
    if (m == ~0ull) return;
    int x, y;
    const bool mine = pixel_of_lane(p, (int)blockIdx.x, x, y) && !((m >> threadIdx.x) & 1ull);
    if (mine) {
        const size_t pix = (size_t)y * p.width + x;
        const v3 rd0 = primary_direction(p, (float)x + 0.5f, (float)y + 0.5f);
        const float4 prev = reinterpret_cast<const float4*>(p.src)[pix];
        float mr = prev.x, mg = prev.y, mb = prev.z;
        // fs:408-410 with energy 1, folded as the render kernels fold it (pt_render_kernel: the wave that sees only sky)
        const float t = 0.5f * (rd0.y + 1.0f);
        const float omt = 1.0f - t;
        const float sr = omt + 0.5f * t, sg = omt + 0.7f * t, sb = omt + 1.0f * t;
        int sample = p.first_sample;
        for (int k = 0; k < p.num_samples; ++k) {
            if (p.accum_mode == 0) {
                if (mr == sr && mg == sg && mb == sb) break;
                const float a = ieee_rcp((float)sample);
                mr = mr + a * (sr - mr);
                mg = mg + a * (sg - mg);
                mb = mb + a * (sb - mb);
            } else {
                mr += sr; mg += sg; mb += sb;
            }
            sample += p.sample_stride;
        }
        const float alpha = p.accum_mode == 0 ? 1.0f : prev.w + (float)p.num_samples;
        reinterpret_cast<float4*>(p.dst)[pix] = make_float4(mr, mg, mb, alpha);
        // what a render kernel leaves for a pixel whose camera ray misses
        if (p.primary_cache && !p.primary_cache_valid) p.primary_cache[pix] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xffffffffu));
    }
    // queries in the reference's sense: the camera ray's, once per sample (none of them evaluated here)
    const unsigned long long lanes = (unsigned long long)__builtin_popcountll(__builtin_amdgcn_ballot_w64(mine));
    if (threadIdx.x == 0 && lanes) atomicAdd(&p.counters[(size_t)(blockIdx.x % PT_COUNTER_SLOTS) * PT_COUNTER_STRIDE], lanes * (unsigned long long)p.num_samples);
}

int window_tiles(const PtKernelParams& p) { return p.pixel_list ? p.pixel_tiles : PT_TILES_ACROSS(p.win_w) * PT_TILES_ACROSS(p.win_h); }
}  // namespace

hipError_t pt_launch_prepare_triangles(const float* vertices, const uint32_t* indices, int32_t num_triangles,
                                       const float ro[3], float* tris, float* vnormals, float* gnormals, float* cull,
                                       float* tris_abs, float* tris_grid, int write_static, hipStream_t stream) {
    if (num_triangles <= 0) return hipSuccess;
    hipLaunchKernelGGL(pt_prepare_triangles_kernel, dim3((num_triangles + 255) / 256), dim3(256), 0, stream, vertices,
                       indices, num_triangles, ro[0], ro[1], ro[2], tris, vnormals, gnormals, cull, tris_abs, tris_grid, write_static);
    return hipGetLastError();
}

hipError_t pt_launch_prepare_spheres(const float* spheres, const float ro[3], float* recs, hipStream_t stream) {
    hipLaunchKernelGGL(pt_prepare_spheres_kernel, dim3(1), dim3(64), 0, stream, spheres, ro[0], ro[1], ro[2], recs);
    return hipGetLastError();
}

size_t pt_compact_temp_words(int32_t num_triangles) {
    return (size_t)(num_triangles > 0 ? (num_triangles + PT_COMPACT_CHUNK - 1) / PT_COMPACT_CHUNK : 0) + 1;
}

hipError_t pt_launch_compact_triangles(const float* tris, int32_t num_triangles, float* live, uint32_t* live_count, uint32_t* tmp, hipStream_t stream) {
    if (num_triangles <= 0) return hipSuccess;
    const int blocks = (num_triangles + PT_COMPACT_CHUNK - 1) / PT_COMPACT_CHUNK;
    const float4* in = reinterpret_cast<const float4*>(tris);
    hipLaunchKernelGGL(pt_compact_count_kernel, dim3(blocks), dim3(PT_COMPACT_THREADS), 0, stream, in, num_triangles, tmp);
    hipLaunchKernelGGL(pt_compact_offsets_kernel, dim3(1), dim3(1024), 0, stream, tmp, blocks, reinterpret_cast<float4*>(live), live_count);
    hipLaunchKernelGGL(pt_compact_scatter_kernel, dim3(blocks), dim3(PT_COMPACT_THREADS), 0, stream, in, num_triangles, tmp, reinterpret_cast<float4*>(live));
    return hipGetLastError();
}

hipError_t pt_launch_fill_rgba(float* img, size_t num_pixels, float r, float g, float b, float a, hipStream_t stream) {
    if (num_pixels == 0) return hipSuccess;
    const size_t blocks = (num_pixels + 255) / 256;
    hipLaunchKernelGGL(pt_fill_rgba_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream,
                       reinterpret_cast<float4*>(img), num_pixels, make_float4(r, g, b, a));
    return hipGetLastError();
}

hipError_t pt_launch_present(const float* img, size_t num_pixels, uint32_t* out_rgba8, hipStream_t stream) {
    if (num_pixels == 0) return hipSuccess;
    const size_t blocks = (num_pixels + 255) / 256;
    hipLaunchKernelGGL(pt_present_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream,
                       reinterpret_cast<const float4*>(img), num_pixels, out_rgba8);
    return hipGetLastError();
}

hipError_t pt_launch_arith_selftest(unsigned long long* bad, hipStream_t stream) {
    hipLaunchKernelGGL(pt_arith_selftest_kernel, dim3(256 * 16), dim3(256), 0, stream, bad);
    return hipGetLastError();
}

hipError_t pt_launch_schedule_items(const uint32_t* tile_cost, PtWorkItem* items, uint32_t* sched_counts, uint32_t* split_tiles, uint32_t* sched_tmp,
                                    int num_tiles, int num_samples, int max_split_tiles, int sort, int groups, hipStream_t stream) {
    if (num_tiles <= 0) return hipSuccess;
    if (groups < 1 || groups > PT_SCHED_MAX_GROUPS) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(sched_tmp, 0, PT_SCHED_SHARES * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const dim3 grid(PT_SCHED_BLOCKS), block(PT_SCHED_THREADS);
    hipLaunchKernelGGL(pt_sched_total_kernel, grid, block, 0, stream, tile_cost, num_tiles, sched_tmp);
    hipLaunchKernelGGL(pt_sched_count_kernel, grid, block, 0, stream, tile_cost, num_tiles, num_samples, max_split_tiles, sort, groups, sched_tmp);
    hipLaunchKernelGGL(pt_sched_offsets_kernel, dim3(1), dim3(64), 0, stream, sched_tmp, sched_counts, groups);
    hipLaunchKernelGGL(pt_sched_scatter_kernel, grid, block, 0, stream, tile_cost, num_tiles, num_samples, max_split_tiles, sort, groups, sched_tmp, items,
                       sched_counts, split_tiles);
    return hipGetLastError();
}

// flag <- 1 when the queries counted so far end the sample at the first or second query on average (an open scene: most
// bounce rays leave), 0 when paths are longer (a closed scene); unchanged while fewer than 65,536 samples are counted.
__global__ __launch_bounds__(PT_COUNTER_SLOTS) void pt_grid_occ_flag_kernel(const unsigned long long* __restrict__ counters, unsigned long long samples,
                                                                            uint32_t* __restrict__ flag) {
    __shared__ unsigned long long part[PT_COUNTER_SLOTS / 64];
    unsigned long long rays = counters[(size_t)threadIdx.x * PT_COUNTER_STRIDE];
    rays = wave_sum(rays);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = rays;
    __syncthreads();
    if (threadIdx.x == 0 && samples >= 65536ull) {
        unsigned long long total = 0;
        for (int w = 0; w < PT_COUNTER_SLOTS / 64; ++w) total += part[w];
        *flag = total < 2ull * samples ? 1u : 0u;
    }
}

hipError_t pt_launch_grid_occ_flag(const unsigned long long* counters, unsigned long long samples, uint32_t* flag, hipStream_t stream) {
    hipLaunchKernelGGL(pt_grid_occ_flag_kernel, dim3(1), dim3(PT_COUNTER_SLOTS), 0, stream, counters, samples, flag);
    return hipGetLastError();
}

hipError_t pt_launch_pixel_list(const PtKernelParams& p, unsigned long long* mask, uint32_t* counts, uint32_t* offsets, uint32_t* list, uint32_t* total,
                                hipStream_t stream) {
    if (p.pixel_list != nullptr || p.grid_cells == nullptr || PT_TILE_WAVES != 1 || p.win_w <= 0 || p.win_h <= 0 || p.width > 65535 || p.height > 65535)
        return hipErrorInvalidValue;
    const int blocks = window_tiles(p);
    const size_t words = ((size_t)p.win_w * p.win_h + 63) / 64 * 64;
    hipError_t e = hipMemsetAsync(list, 0xff, words * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pt_pixel_mask_kernel, dim3(blocks), dim3(64), 0, stream, p, mask, counts);
    hipLaunchKernelGGL(pt_pixel_offsets_kernel, dim3(1), dim3(1024), 0, stream, counts, blocks, offsets, total);
    hipLaunchKernelGGL(pt_pixel_scatter_kernel, dim3(blocks), dim3(64), 0, stream, p, mask, offsets, list);
    return hipGetLastError();
}

hipError_t pt_launch_sky_fold(const PtKernelParams& p, const unsigned long long* mask, hipStream_t stream) {
    if (p.pixel_list != nullptr || PT_TILE_WAVES != 1) return hipErrorInvalidValue;
    if (p.win_w <= 0 || p.win_h <= 0 || p.num_samples <= 0) return hipSuccess;
    hipLaunchKernelGGL(pt_sky_fold_kernel, dim3(window_tiles(p)), dim3(64), 0, stream, p, mask);
    return hipGetLastError();
}

hipError_t pt_launch_render(const PtKernelParams& p, int program_kind, hipStream_t stream) {
    if (p.win_w <= 0 || p.win_h <= 0 || p.num_samples <= 0) return hipSuccess;
    const dim3 grid(p.items ? p.max_items : window_tiles(p)), block(PT_TILE_THREADS);
    if (program_kind == 2) {
        if (PT_TILE_WAVES != 1 || p.items) return hipErrorInvalidValue;
        hipLaunchKernelGGL(pt_cpu_tracer_kernel, dim3(window_tiles(p)), dim3(64), 0, stream, p);
        return hipGetLastError();
    }
    if (program_kind == 0) hipLaunchKernelGGL((pt_render_kernel<0, false>), grid, block, 0, stream, p);
    else if (p.shading_mode != 0) {
        // lane_walk: 1 = lanes decoupled (default), 2 = the per-lane walk with all lanes in step, 0 = one walk per wave
        if (p.bvh_nodes_abs && p.lane_walk == 1 && PT_TILE_WAVES == 1) hipLaunchKernelGGL(pt_render_decoupled_kernel, grid, block, 0, stream, p);
        else if (p.bvh_nodes_abs && p.lane_walk) hipLaunchKernelGGL((pt_render_kernel<5, true>), grid, block, 0, stream, p);
        else if (p.bvh_nodes_abs) hipLaunchKernelGGL((pt_render_kernel<2, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((pt_render_kernel<1, true>), grid, block, 0, stream, p);
    }
    else if (p.grid_cells && p.grid_carry) hipLaunchKernelGGL((pt_render_kernel<7, false>), grid, block, 0, stream, p);
    else if (p.grid_cells) hipLaunchKernelGGL((pt_render_kernel<4, false>), grid, block, 0, stream, p);
    else if (p.bvh_nodes) hipLaunchKernelGGL((pt_render_kernel<2, false>), grid, block, 0, stream, p);
    else if (p.cull) hipLaunchKernelGGL((pt_render_kernel<3, false>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((pt_render_kernel<1, false>), grid, block, 0, stream, p);
    if (p.items && p.max_split_tiles > 0 && p.num_samples > 1)
        hipLaunchKernelGGL(pt_fold_samples_kernel, dim3(p.max_split_tiles), dim3(PT_TILE_THREADS), 0, stream, p);
    return hipGetLastError();
}

hipError_t pt_launch_verify(const PtKernelParams& p, hipStream_t stream) {
    if (p.win_w <= 0 || p.win_h <= 0) return hipSuccess;
    const dim3 grid(window_tiles(p)), block(PT_TILE_THREADS);
    if (p.shading_mode != 0) {
        if (p.bvh_nodes_abs && p.lane_walk == 1 && PT_TILE_WAVES == 1) hipLaunchKernelGGL((pt_verify_kernel<6, true>), grid, block, 0, stream, p);
        else if (p.bvh_nodes_abs && p.lane_walk) hipLaunchKernelGGL((pt_verify_kernel<5, true>), grid, block, 0, stream, p);
        else if (p.bvh_nodes_abs) hipLaunchKernelGGL((pt_verify_kernel<2, true>), grid, block, 0, stream, p);
        else return hipErrorInvalidValue;  // the materials mode without a hierarchy IS the all-triangles loop
    }
    else if (p.grid_cells) hipLaunchKernelGGL((pt_verify_kernel<4, false>), grid, block, 0, stream, p);
    else if (p.bvh_nodes) hipLaunchKernelGGL((pt_verify_kernel<2, false>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((pt_verify_kernel<1, false>), grid, block, 0, stream, p);
    return hipGetLastError();
}

hipError_t pt_launch_primary(const PtKernelParams& p, int program_kind, hipStream_t stream) {
    if (p.win_w <= 0 || p.win_h <= 0) return hipSuccess;
    if (program_kind == 0) hipLaunchKernelGGL(pt_primary_kernel<0>, dim3(window_tiles(p)), dim3(PT_TILE_THREADS), 0, stream, p);
    else if (p.grid_cells) hipLaunchKernelGGL(pt_primary_kernel<4>, dim3(window_tiles(p)), dim3(PT_TILE_THREADS), 0, stream, p);
    else if (p.bvh_nodes) hipLaunchKernelGGL(pt_primary_kernel<2>, dim3(window_tiles(p)), dim3(PT_TILE_THREADS), 0, stream, p);
    else if (p.cull) hipLaunchKernelGGL(pt_primary_kernel<3>, dim3(window_tiles(p)), dim3(PT_TILE_THREADS), 0, stream, p);
    else hipLaunchKernelGGL(pt_primary_kernel<1>, dim3(window_tiles(p)), dim3(PT_TILE_THREADS), 0, stream, p);
    return hipGetLastError();
}
