/*
 * TEST INFRASTRUCTURE ONLY -- this file is the parity oracle. It is never linked into, imported by or
 * shipped with the product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it,
 * and only as the checker / the CPU figure printed beside the GPU one.
 *
 * CPU restatement (plain C, fp32, one pixel at a time) of what the reference's fragment shader
 * /root/reference/data/shaders/pathtracing.fs ("fs" below) computes, with every GLSL built-in written out
 * in the exact IEEE-754 operation order that Mesa 23.2.1 llvmpipe (LLVM 15, x86-64) uses for it. Those orders,
 * and the polynomial forms of sin/cos/exp2/pow, were pinned by running probe shaders on llvmpipe in the build
 * container and comparing bit patterns (oracle/llvmpipe/, tests/golden/lp_math_*.npz); with them this file
 * reproduces the reference shader's RGBA32F output BIT FOR BIT on the golden images (tests/test_oracle_golden.py).
 *
 * PARITY PINNED: against goldens rendered by the reference shader itself on llvmpipe
 * (oracle/gen_golden.py -> tests/golden/).
 *
 * Must be compiled with -ffp-contract=off (see Makefile): a*b+c below means two roundings; a fused
 * multiply-add is used only where written as fmaf(), which is where llvmpipe emits llvm.fmuladd on an
 * FMA-capable host.
 *
 * llvmpipe runs its shader threads with MXCSR flush-to-zero + denormals-are-zero; entry points here do the same.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <xmmintrin.h>
#include <pmmintrin.h>

#define PTO_API __attribute__((visibility("default")))

typedef struct { float x, y, z; } v3;

/* Layouts follow the reference host's uploads (PathTracer.cpp:1253-1288). */
typedef struct PtoScene {
    int32_t num_triangles;
    int32_t num_vertices;
    const uint32_t* indices;  /* 3 per triangle: RGB32UI texel i (PathTracer.cpp:1274) */
    const float* vertices;    /* 6 per vertex: position at texel 2i, normal at texel 2i+1 (PathTracer.cpp:71-75,1287) */
    const float* spheres;     /* 10 x (cx,cy,cz,r) (PathTracer.cpp:589-592,1258) */
    const float* materials;   /* 10 x (r,g,b,type); fetched then overwritten by the shader (fs:228-229) */
    const int32_t* material_ids; /* per triangle, 0..9; only read in the build-defined "materials" shading mode */
} PtoScene;

typedef struct PtoUniforms {
    float cam_pos[3], cam_llc[3], cam_hor[3], cam_ver[3]; /* fs:6-9, set at PathTracer.cpp:1372-1375 */
    int32_t num_triangles;                                /* fs:17 */
    int32_t depth;                                        /* fs:18 (reference passes 10) */
    int32_t width, height;                                /* textureSize(u_Texture,0) (fs:381) */
    int32_t variant;      /* 0 = "world": IntersectWorld call (fs:395, as committed); 1 = "mesh": IntersectMesh call (fs:396) */
    int32_t normal_mode;  /* 0 = interpolated vertex normals (fs:165); 1 = geometric normal, the shader's alternative (fs:166) */
    int32_t shading_mode; /* 0 = the shader as it is (Phong lobe, material ignored, origin fixed);
                             1 = build-defined diffuse/emissive mode = the shader with the edits of
                                 oracle/llvmpipe/edits/mesh_materials_geo.txt (pinned against llvmpipe running exactly that);
                             2 = the same with the Lambert bounce in the form of fs:272, rd = normalize(N + randomUnitVector())
                                 (fs:57-65): oracle/llvmpipe/edits/mesh_materials_unit_geo.txt */
} PtoUniforms;

/* ------------------------------------------------------------------------------------------------ */
/* GLSL built-ins, in llvmpipe's evaluation order                                                    */

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
/* dot(a,b) = a.x*b.x + (a.y*b.y + a.z*b.z) */
static inline float dot3(v3 a, v3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
static inline v3 cross3(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* normalize(v) = v * (1.0f / sqrtf(dot(v,v))): inversesqrt is a correctly rounded sqrt followed by a division */
static inline v3 normalize3(v3 a) { return scale(a, 1.0f / sqrtf(dot3(a, a))); }
/* reflect(I,N) = I - (2*dot(N,I))*N */
static inline v3 reflect3(v3 i, v3 n) { return sub(i, scale(n, 2.0f * dot3(n, i))); }
static inline float fract1(float x) { return x - floorf(x); }
static inline float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

static inline float as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* Even/odd split Horner evaluation with fused steps: llvmpipe's polynomial helper. */
static float lp_poly(float x, const float* c, int n) {
    float x2 = x * x, even = 0.f, odd = 0.f;
    int have_even = 0, have_odd = 0;
    for (int i = n - 1; i >= 0; i--) {
        if ((i & 1) == 0) { even = have_even ? fmaf(x2, even, c[i]) : c[i]; have_even = 1; }
        else              { odd  = have_odd  ? fmaf(x2, odd,  c[i]) : c[i]; have_odd = 1; }
    }
    return fmaf(odd, x, even);
}

/* sin/cos: Cephes-style range reduction to [0,pi/4] + two degree-3 polynomials in x^2. */
static float lp_sincos(float a, int want_cos) {
    const uint32_t ai = as_uint(a);
    const float x_abs = as_float(ai & 0x7fffffffu);
    const float scale_y = x_abs * 1.27323954473516f;
    const int32_t j = (int32_t)scale_y;            /* truncation */
    const int32_t j1 = j + 1;
    const int32_t je = j1 & ~1;
    const float y = (float)je;
    const int32_t sel = want_cos ? je - 2 : je;
    const uint32_t sign = want_cos ? (((uint32_t)(4 & ~sel)) << 29)
                                   : ((ai ^ ((uint32_t)j1 << 29)) & 0x80000000u);
    const int use_sin_poly = (sel & 2) == 0;
    float x = fmaf(y, -0.78515625f, x_abs);
    x = fmaf(y, -2.4187564849853515625e-4f, x);
    x = fmaf(y, -3.77489497744594108e-8f, x);
    const float z = x * x;
    float c = fmaf(z, 2.443315711809948E-005f, -1.388731625493765E-003f);
    c = fmaf(c, z, 4.166664568298827E-002f);
    c = c * z;
    c = c * z;
    c = c - z * 0.5f;
    c = c + 1.0f;
    float s = fmaf(z, -1.9515295891E-4f, 8.3321608736E-3f);
    s = fmaf(s, z, -1.6666654611E-1f);
    s = s * z;
    s = fmaf(s, x, x);
    float r = as_float(as_uint(use_sin_poly ? s : c) ^ sign);
    return fminf(fmaxf(r, -1.0f), 1.0f);
}

static float lp_exp2(float x) {
    static const float k[6] = {1.000000000000000000000f, 0.693153073200168932794f, 0.240153617044375388211f,
                               0.0558263180532956664775f, 0.00898934009049466391101f, 0.00187757667519147912699f};
    x = fminf(128.0f, x);
    x = fmaxf(-126.99999f, x);
    const float ip = floorf(x);
    const float fp = x - ip;
    const float e = as_float((uint32_t)(((int32_t)ip + 127) << 23));
    return e * lp_poly(fp, k, 6);
}

static float lp_log2(float x) {
    static const float k[5] = {2.88539009343309178325f, 0.961791550404184197881f, 0.577440339438736392009f,
                               0.403343858251329912514f, 0.406718052498846252698f};
    const uint32_t i = as_uint(x);
    const float e = (float)((int32_t)((i & 0x7f800000u) >> 23) - 127);
    const float m = as_float((i & 0x007fffffu) | 0x3f800000u);
    const float y = (m - 1.0f) / (m + 1.0f);
    const float z = y * y;
    float r = fmaf(y, lp_poly(z, k, 5), e);
    if (x >= INFINITY) r = INFINITY;
    if (x == 0.0f) r = -INFINITY;
    if (x < 0.0f) r = NAN;
    return r;
}

static float lp_pow(float x, float y) { return lp_exp2(lp_log2(x) * y); }

/* Array form for the known-answer tests: kind 0 = sin(a), 1 = cos(a), 2 = exp2(a), 3 = pow(a, b). */
PTO_API void pto_lp_math(int kind, int n, const float* a, const float* b, float* out) {
    unsigned csr = _mm_getcsr();
    _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);
    _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
    for (int i = 0; i < n; i++) {
        switch (kind) {
            case 0: out[i] = lp_sincos(a[i], 0); break;
            case 1: out[i] = lp_sincos(a[i], 1); break;
            case 2: out[i] = lp_exp2(a[i]); break;
            default: out[i] = lp_pow(a[i], b[i]); break;
        }
    }
    _mm_setcsr(csr);
}

/* ------------------------------------------------------------------------------------------------ */
/* Per-pixel state: gl_FragCoord.xy, u_Samples and the shader's global randState (fs:21)             */

typedef struct {
    float fcx, fcy;    /* gl_FragCoord.xy = pixel centre (x+0.5, y+0.5), row 0 at the bottom */
    float samples;     /* u_Samples: 1-based pass index as float (PathTracer.cpp:1305,1378) */
    float rs_x, rs_y;  /* randState */
} PixelCtx;

/* fs:27 */
static float ign(float x, float y) {
    return fract1(52.9829189f * fract1(x * 0.06711056f + y * 0.00583715f));
}
/* fs:29-36: vec2 + float + float is evaluated left to right, per component */
static float rng_rand(PixelCtx* p) {
    float ax = (p->fcx + p->rs_x) + p->samples;
    float ay = (p->fcy + p->rs_x) + p->samples;
    float r = ign(ax, ay);
    p->rs_x = p->rs_x + r;
    p->rs_y = p->rs_y + r * 2.0f;
    return r;
}
/* fs:37-42: mod(x, 1.0) = x - 1.0*floor(x/1.0) */
static float rng_rand01(PixelCtx* p) {
    float r = rng_rand(p);
    return r - floorf(r);
}

/* fs:44-55; returns the three columns */
static void tangent_space(v3 n, v3* c0, v3* c1, v3* c2) {
    v3 helper = V(1.0f, 0.0f, 0.0f);
    if (fabsf(n.x) > 0.99f) helper = V(0.0f, 0.0f, 1.0f);
    v3 t = normalize3(cross3(n, helper));
    v3 b = normalize3(cross3(n, t));
    *c0 = t; *c1 = n; *c2 = b;
}

/* fs:67-76 */
static v3 random_hemisphere(PixelCtx* p, float r1, float distrib) {
    float cos_t = lp_pow(r1, 1.0f / (distrib + 1.0f));
    float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t));
    float phi = (rng_rand01(p) * 2.0f) * 3.141592653589f;
    return V(lp_sincos(phi, 1) * sin_t, cos_t, lp_sincos(phi, 0) * sin_t);
}

/* fs:83-121 */
static float intersect_world(const PtoScene* sc, v3 ro, v3 rd, v3* rec_p, v3* rec_n, float* rec_t) {
    float hit = -1.0f;
    for (int obj = 0; obj < 10; obj++) {
        const float* d = sc->spheres + 4 * obj;  /* texture(u_SpheresTexture, (0.1001*id, 0)) NEAREST -> texel id */
        v3 center = V(d[0], d[1], d[2]);
        float radius = d[3];
        float inv_radius = 1.0f / radius;
        v3 to = sub(ro, center);
        float B = -dot3(to, rd);
        float C = dot3(to, to) - radius * radius;
        float disc = B * B - C;
        if (disc > 0.0f) {
            float root = sqrtf(disc);
            float temp = B - root;
            temp = temp > 0.0f ? temp : (B + root);
            if (temp > 0.0f && temp < *rec_t) {
                *rec_p = add(ro, scale(rd, temp));
                *rec_n = scale(sub(*rec_p, center), inv_radius);
                *rec_t = temp;
                hit = (float)obj;
            }
        }
    }
    return hit;
}

/* fs:125-174. Returns the index of the closest triangle (the shader returns 0.0 for any hit, fs:167; the
 * instrumented shader used for the id goldens returns float(i)), or -1. */
static int intersect_mesh(const PtoScene* sc, const PtoUniforms* u, v3 ro, v3 rd, v3* rec_p, v3* rec_n, float* rec_t) {
    int hit = -1;
    for (int i = 0; i < u->num_triangles; i++) {
        const uint32_t* ix = sc->indices + 3 * i;
        const float* pa = sc->vertices + 6 * (size_t)ix[0];
        const float* pb = sc->vertices + 6 * (size_t)ix[1];
        const float* pc = sc->vertices + 6 * (size_t)ix[2];
        v3 a = V(pa[0], pa[1], pa[2]), b = V(pb[0], pb[1], pb[2]), c = V(pc[0], pc[1], pc[2]);
        v3 e1 = sub(b, a), e2 = sub(c, a);
        v3 q = cross3(rd, e2);
        float det = dot3(e1, q);
        if (det >= 0.0001f) {
            v3 ap = sub(ro, a);
            v3 r = cross3(ap, e1);
            float uu = dot3(ap, q);
            float vv = dot3(rd, r);
            if (uu >= 0.0f && uu <= det && vv >= 0.0f && (uu + vv) <= det) {
                float ood = 1.0f / det;
                float temp = dot3(e2, r) * ood;
                if (temp > 0.0f && temp < *rec_t) {
                    *rec_t = temp;
                    *rec_p = add(ro, scale(rd, temp));
                    if (u->normal_mode == 0) {
                        float ux = uu * ood, uy = vv * ood, uz = (1.0f - ux) - uy;
                        v3 na = V(pa[3], pa[4], pa[5]), nb = V(pb[3], pb[4], pb[5]), nc = V(pc[3], pc[4], pc[5]);
                        *rec_n = normalize3(add(add(scale(na, uz), scale(nb, ux)), scale(nc, uy)));
                    } else {
                        *rec_n = normalize3(cross3(e1, e2));
                    }
                    hit = i;
                }
            }
        }
    }
    return hit;
}

/* fs:218-222 for a grey SpecularColor (all three channels equal) */
static float fresnel_schlick(float spec, v3 e, v3 h) {
    return spec + (1.0f - spec) * lp_exp2(-8.656170f * clamp01(dot3(e, h)));
}

/* fs:224-361, active lines only. `ro` is never written (every "ro = ..." is commented out). */
static void shade(PixelCtx* p, v3* attenuation, v3* rd, v3 N) {
    const float specularity = 0.94f;
    const float alpha = 1024.0f;           /* pow(1024.0, 1.0*1.0), fs:254-256 */
    const float albedo = 0.5f;             /* material = vec4(0.5), fs:229 */
    v3 R = reflect3(*rd, N);
    float r1 = rng_rand01(p);
    v3 c0, c1, c2;
    tangent_space(R, &c0, &c1, &c2);
    v3 h = random_hemisphere(p, r1, alpha);
    /* mat3 * vec3 = (c0*h.x + c1*h.y) + c2*h.z */
    *rd = add(add(scale(c0, h.x), scale(c1, h.y)), scale(c2, h.z));
    float diffuse = fminf(1.0f - specularity, albedo);
    float f = (alpha + 2.0f) / (alpha + 1.0f);
    float specular = (fresnel_schlick(specularity, N, *rd) * specularity) * clamp01(dot3(N, *rd) * f);
    float k = diffuse + specular;
    *attenuation = scale(*attenuation, k);
}

/* Shade() of the "materials" variant (edits/mesh_materials_geo.txt, replacing fs:229): material = (r,g,b,e);
 * e > 0: emitter of radiance rgb*e, path ends; else Lambert of albedo rgb, cosine-weighted direction about N from the
 * shader's own GetTangentSpace/randomHemisphere(r1, 1.0) (fs:44-76), and the origin moves to the hit point (fs:274). */
/* fs:57-65 randomUnitVector(): z uniform in [-1, 1), then a point on the circle of that height. */
static v3 random_unit_vector(PixelCtx* p) {
    float z = rng_rand01(p) * 2.0f - 1.0f;
    float r = sqrtf(1.0f - z * z);
    float a = (rng_rand01(p) * 2.0f) * 3.141592653589f;  /* M_PI as the shader defines it, fs:25 */
    return V(lp_sincos(a, 1) * r, lp_sincos(a, 0) * r, z);
}

static void shade_materials(PixelCtx* p, const float* m, v3* attenuation, v3* ro, v3* rd, v3 P, v3 N, v3* emitted, int unit_form) {
    if (m[3] > 0.0f) {
        *emitted = mul(*attenuation, V(m[0] * m[3], m[1] * m[3], m[2] * m[3]));
        *attenuation = V(0.0f, 0.0f, 0.0f);
        return;
    }
    if (unit_form) {
        /* the Lambert bounce in the form of fs:272: rd = normalize(N + randomUnitVector()) */
        *rd = normalize3(add(N, random_unit_vector(p)));
    } else {
        float r1 = rng_rand01(p);
        v3 c0, c1, c2;
        tangent_space(N, &c0, &c1, &c2);
        v3 h = random_hemisphere(p, r1, 1.0f);
        *rd = add(add(scale(c0, h.x), scale(c1, h.y)), scale(c2, h.z));
    }
    *ro = add(P, scale(*rd, 0.0001f));
    *attenuation = mul(*attenuation, V(m[0], m[1], m[2]));
}

typedef struct { int id; float t; v3 rd; } Primary;

/* fs:376-410: one invocation of main() up to "accum += energy * skyLight"; returns accum. */
static v3 trace_pixel(const PtoScene* sc, const PtoUniforms* u, int px, int py, float samples,
                      uint64_t* rays, Primary* prim) {
    PixelCtx p;
    p.fcx = (float)px + 0.5f; p.fcy = (float)py + 0.5f;
    p.samples = samples; p.rs_x = 0.0f; p.rs_y = 0.0f;

    float ijx = p.fcx / (float)u->width, ijy = p.fcy / (float)u->height;
    v3 ro = V(u->cam_pos[0], u->cam_pos[1], u->cam_pos[2]);
    v3 llc = V(u->cam_llc[0], u->cam_llc[1], u->cam_llc[2]);
    v3 hor = V(u->cam_hor[0], u->cam_hor[1], u->cam_hor[2]);
    v3 ver = V(u->cam_ver[0], u->cam_ver[1], u->cam_ver[2]);
    /* ((LLC + ij.x*H) + ij.y*V) - ro */
    v3 rd = normalize3(sub(add(add(llc, scale(hor, ijx)), scale(ver, ijy)), ro));
    if (prim) { prim->id = -1; prim->t = 0.0f; prim->rd = rd; }

    v3 energy = V(1.0f, 1.0f, 1.0f);
    v3 emitted = V(0.0f, 0.0f, 0.0f);
    for (int level = 0; level < u->depth; level++) {
        float rec_t = 1.0e38f;
        v3 rec_p = V(0, 0, 0), rec_n = V(0, 0, 0);
        int hit;
        if (u->variant == 0) hit = (int)intersect_world(sc, ro, rd, &rec_p, &rec_n, &rec_t);
        else hit = intersect_mesh(sc, u, ro, rd, &rec_p, &rec_n, &rec_t);
        if (rays) (*rays)++;
        if (prim && level == 0) { prim->id = hit; prim->t = hit >= 0 ? rec_t : 1.0e38f; }
        if (hit >= 0) {
            if (u->shading_mode == 0) shade(&p, &energy, &rd, rec_n);
            else shade_materials(&p, sc->materials + 4 * sc->material_ids[hit], &energy, &ro, &rd, rec_p, rec_n, &emitted, u->shading_mode == 2);
        }
        int any_energy = (energy.x != 0.0f) || (energy.y != 0.0f) || (energy.z != 0.0f);
        if (!any_energy || hit < 0) break;
    }
    float t = 0.5f * (rd.y + 1.0f);
    v3 sky = add(V(1.0f - t, 1.0f - t, 1.0f - t), scale(V(0.5f, 0.7f, 1.0f), t));
    if (u->shading_mode != 0) return add(mul(energy, sky), emitted);  /* vec3(0) + (energy*sky + g_emitted) */
    return mul(energy, sky);  /* vec3(0) + energy*sky */
}

/* fs:412-415: mix(x, y, a) = x + a*(y - x) with a = 1.0/u_Samples; `px` = this pixel in the other ping-pong
 * target, overwritten with the new value (alpha = 1). */
static void blend_pixel(float* px, v3 accum, float samples) {
    float a = 1.0f / samples;
    px[0] = px[0] + a * (accum.x - px[0]);
    px[1] = px[1] + a * (accum.y - px[1]);
    px[2] = px[2] + a * (accum.z - px[2]);
    px[3] = 1.0f;
}

/* ------------------------------------------------------------------------------------------------ */
/* Entry points                                                                                       */

typedef struct { unsigned csr; } FpGuard;
static FpGuard fp_enter(void) {
    FpGuard g; g.csr = _mm_getcsr();
    _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);
    _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
    return g;
}
static void fp_leave(FpGuard g) { _mm_setcsr(g.csr); }

/* Progressive render of the window [x0,x0+w) x [y0,y0+h) of a width x height frame: passes
 * first_sample .. first_sample+num_samples-1 (1-based u_Samples), applied in place to `rgba`
 * (full-frame, row 0 = bottom, 4 floats per pixel; alpha is set to 1). Because a pixel only ever reads its
 * own previous value, updating in place equals the reference's ping-pong (PathTracer.cpp:1384,1398,1493).
 * Returns the number of rays (bounce-loop iterations, PathTracer.cpp:947) traced. */
PTO_API uint64_t pto_render(const PtoScene* sc, const PtoUniforms* u, int first_sample, int num_samples,
                            float* rgba, int x0, int y0, int w, int h) {
    FpGuard g = fp_enter();
    uint64_t rays = 0;
    for (int y = y0; y < y0 + h; y++) {
        for (int x = x0; x < x0 + w; x++) {
            float* px = rgba + 4 * ((size_t)y * u->width + x);
            for (int s = 0; s < num_samples; s++) {
                float samples = (float)(first_sample + s);
                blend_pixel(px, trace_pixel(sc, u, x, y, samples, &rays, 0), samples);
            }
        }
    }
    fp_leave(g);
    return rays;
}

/* ------------------------------------------------------------------------------------------------ */
/* The reference's CPU tracer (PathTracer.cpp:932-1052 Scene::color, :830-852 scatter, :595-640 Sphere::hit, :562-586  */
/* Mesh::hit, :170-190 AABB::test, :101-152 HitRecord::hit, :742-745 Camera::generateRay), restated with a             */
/* counter-based random number generator in place of its serial libc rand() stream, so that pixels and samples are      */
/* independent and a parallel implementation can be compared bit for bit. PARITY of this restatement against the        */
/* reference itself is STATISTICAL (tests/golden/cpu_tracer_stats.npz: per-pixel mean and variance from the reference's */
/* own code, compiled from where it lies: oracle/_ref/cpu_tracer). Deviations, all forced by the generator: (1) the     */
/* random numbers; (2) cos/sin of randomUnitVector() are the polynomial forms above instead of libm's cosf/sinf.        */

/* float3::dot / length / normalize (PathTracer.cpp:58-62): sums left to right */
static inline float dotc(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 normalizec(v3 a) { return scale(a, 1.0f / sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z)); }

static inline uint32_t pcg_hash(uint32_t x) {
    uint32_t state = x * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
typedef struct { uint32_t key, draw; } CtrRng;
/* nextFloat(0, 1) (PathTracer.cpp:765-768): (float)rand() / (float)RAND_MAX with a 31-bit value; (float)RAND_MAX = 2^31 */
static inline float ctr_next(CtrRng* r) {
    uint32_t h = pcg_hash(r->key + pcg_hash(r->draw++));
    return (float)(h >> 1) / 2147483648.0f;
}

typedef struct { v3 o, d; float mint, maxt; } RayC;
typedef struct { v3 point, normal; float t; } RecC;

static int cpu_sphere_hit(const float* s, const RayC* ray, RecC* rec) {
    v3 c = V(s[0], s[1], s[2]);
    float radius = s[3];
    v3 to = sub(ray->o, c);
    float A = dotc(ray->d, ray->d), B = dotc(to, ray->d), C = dotc(to, to) - radius * radius;
    float disc = B * B - A * C;
    if (disc > 0.0f) {
        float root = sqrtf(disc);
        float temp = (-B - root) / A;
        if (temp < ray->maxt && temp > ray->mint) {
            rec->point = add(ray->o, scale(ray->d, temp));
            rec->normal = scale(sub(rec->point, c), 1.0f / radius);
            rec->t = temp;
            return 1;
        }
        float temp2 = (-B + root) / A;
        if (temp2 < ray->maxt && temp2 > ray->mint) {
            rec->point = add(ray->o, scale(ray->d, temp2));
            rec->normal = scale(sub(rec->point, c), 1.0f / radius);
            rec->t = temp2;
            return 1;
        }
    }
    return 0;
}

static int cpu_aabb_test(const float* lo, const float* hi, const RayC* ray) {
    float tmin = ray->mint, tmax = ray->maxt;
    const float o[3] = {ray->o.x, ray->o.y, ray->o.z}, d[3] = {ray->d.x, ray->d.y, ray->d.z};
    for (int i = 0; i < 3; i++) {
        if (fabsf(d[i]) < 0.001f) {
            if (o[i] < lo[i] || o[i] > hi[i]) return 0;
        } else {
            float ood = 1.0f / d[i];
            float t1 = (lo[i] - o[i]) * ood, t2 = (hi[i] - o[i]) * ood;
            if (t1 > t2) { float x = t1; t1 = t2; t2 = x; }
            if (t1 > tmin) tmin = t1;
            if (t2 < tmax) tmax = t2;
            if (tmin > tmax) return 0;
        }
    }
    return 1;
}

static v3 cpu_color(const PtoScene* sc, int num_triangles, const float* lo, const float* hi, RayC ray, int depth, CtrRng* rng, uint64_t* rays) {
    v3 energy = V(1.0f, 1.0f, 1.0f);
    for (int level = 0; level < depth; ++level) {
        RecC rec;
        rec.point = V(0, 0, 0); rec.normal = V(0, 0, 0); rec.t = 0.0f;
        int hit_index = -1;
        (*rays)++;
        for (int i = 0; i < 10; i++)
            if (cpu_sphere_hit(sc->spheres + 4 * i, &ray, &rec)) { ray.maxt = rec.t; hit_index = i; }
        if (num_triangles > 0) {
            RayC r2 = ray;   /* Mesh::hit takes the ray by value */
            int has = 0;
            if (cpu_aabb_test(lo, hi, &r2)) {
                for (int i = 0; i < num_triangles; i++) {
                    const float* pa = sc->vertices + 6 * sc->indices[3 * i];
                    const float* pb = sc->vertices + 6 * sc->indices[3 * i + 1];
                    const float* pc = sc->vertices + 6 * sc->indices[3 * i + 2];
                    v3 a = V(pa[0], pa[1], pa[2]);
                    v3 e1 = sub(V(pb[0], pb[1], pb[2]), a), e2 = sub(V(pc[0], pc[1], pc[2]), a);
                    v3 q = cross3(r2.d, e2);
                    float det = dotc(e1, q);
                    if (det < 0.0001f) continue;
                    v3 ap = sub(r2.o, a);
                    float u = dotc(ap, q);
                    if (u < 0.0f || u > det) continue;
                    v3 r = cross3(ap, e1);
                    float v = dotc(r2.d, r);
                    if (v < 0.0f || (u + v) > det) continue;
                    float ood = 1.0f / det;
                    float temp = dotc(e2, r) * ood;
                    if (temp > r2.mint && temp < r2.maxt) {
                        r2.maxt = temp;
                        rec.t = temp;
                        rec.point = add(r2.o, scale(r2.d, temp));
                        float uy = u * ood, uz = v * ood, ux = (1.0f - uy) - uz;
                        v3 n = add(add(scale(V(pa[3], pa[4], pa[5]), ux), scale(V(pb[3], pb[4], pb[5]), uy)), scale(V(pc[3], pc[4], pc[5]), uz));
                        rec.normal = normalizec(n);
                        has = 1;
                    }
                }
            }
            if (has) { ray.maxt = rec.t; hit_index = 0; }
        }
        if (hit_index < 0) break;
        const float* m = sc->materials + 4 * hit_index;
        v3 attenuation = V(1.0f, 1.0f, 1.0f);
        /* scatter(), PathTracer.cpp:830-852 */
        ray.o = add(rec.point, scale(rec.normal, 0.001f));
        ray.mint = 0.0f;
        ray.maxt = 3.402823466e+38f;
        if (m[3] == 0.0f) {  /* Material::LAMBERT */
            float z = ctr_next(rng) * 2.0f - 1.0f;
            float a = (ctr_next(rng) * 2.0f) * 3.14159274f;
            float r = sqrtf(1.0f - z * z);
            v3 unit = V(lp_sincos(a, 1) * r, lp_sincos(a, 0) * r, z);
            ray.d = normalizec(add(rec.normal, unit));
            attenuation = V(m[0], m[1], m[2]);
        }
        energy = mul(energy, attenuation);
        if ((energy.x + energy.y) + energy.z == 0.0f) break;
    }
    float t = 0.5f * (ray.d.y + 1.0f);
    v3 sky = V((1.0f - t) + 1.0f * t, (1.0f - t) + 0.7f * t, (1.0f - t) + 0.5f * t);
    return mul(energy, sky);
}

/* Running SUM of the samples first_sample, first_sample + stride, ... (count of them) of every pixel of the window, added
 * to rgba_sum (4 floats per pixel, row 0 = bottom; the 4th is the number of samples folded in). Pixel-order layout; the
 * reference's accumulationBuffer holds the same sums in its tile-traversal order (PathTracer.cpp:1326-1357). */
PTO_API uint64_t pto_cpu_tracer(const PtoScene* sc, const PtoUniforms* u, float max_t, uint32_t seed, int first_sample, int stride,
                                int count, float* rgba_sum, int x0, int y0, int w, int h) {
    FpGuard g = fp_enter();
    uint64_t rays = 0;
    float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (int i = 0; i < sc->num_vertices; i++)
        for (int k = 0; k < 3; k++) {
            float c = sc->vertices[6 * i + k];
            if (c < lo[k]) lo[k] = c;
            if (c > hi[k]) hi[k] = c;
        }
    v3 pos = V(u->cam_pos[0], u->cam_pos[1], u->cam_pos[2]), llc = V(u->cam_llc[0], u->cam_llc[1], u->cam_llc[2]);
    v3 hor = V(u->cam_hor[0], u->cam_hor[1], u->cam_hor[2]), ver = V(u->cam_ver[0], u->cam_ver[1], u->cam_ver[2]);
    for (int y = y0; y < y0 + h; y++) {
        for (int x = x0; x < x0 + w; x++) {
            float* px = rgba_sum + 4 * ((size_t)y * u->width + x);
            /* Camera::generateRay(float(x), float(y)): through the pixel's CORNER, not its centre */
            float fx = (float)x / (float)u->width, fy = (float)y / (float)u->height;
            RayC ray0;
            ray0.o = pos;
            ray0.d = normalizec(sub(add(add(llc, scale(hor, fx)), scale(ver, fy)), pos));
            ray0.mint = 0.0f;
            ray0.maxt = max_t;
            for (int s = 0; s < count; s++) {
                CtrRng rng;
                rng.key = pcg_hash(seed + pcg_hash((uint32_t)(y * u->width + x))) + pcg_hash(0x9E3779B9u * (uint32_t)(first_sample + s * stride));
                rng.draw = 0;
                v3 c = cpu_color(sc, u->num_triangles, lo, hi, ray0, u->depth, &rng, &rays);
                px[0] += c.x; px[1] += c.y; px[2] += c.z; px[3] += 1.0f;
            }
        }
    }
    fp_leave(g);
    return rays;
}

/* Primary-hit map of the same window: ids (int32, -1 = miss), t (1e38 on a miss) and the primary direction. */
PTO_API void pto_primary(const PtoScene* sc, const PtoUniforms* u, int32_t* ids, float* ts, float* rds,
                         int x0, int y0, int w, int h) {
    FpGuard g = fp_enter();
    PtoUniforms u1 = *u;
    u1.depth = 1;
    for (int y = y0; y < y0 + h; y++) {
        for (int x = x0; x < x0 + w; x++) {
            size_t o = (size_t)y * u->width + x;
            Primary pr;
            (void)trace_pixel(sc, &u1, x, y, 1.0f, 0, &pr);
            if (ids) ids[o] = pr.id;
            if (ts) ts[o] = pr.t;
            if (rds) { rds[3 * o] = pr.rd.x; rds[3 * o + 1] = pr.rd.y; rds[3 * o + 2] = pr.rd.z; }
        }
    }
    fp_leave(g);
}

/* Sample-split form: adds the radiance of passes first_sample, first_sample+stride, ... (count of them)
 * into a running fp32 SUM (3 floats per pixel used, 4 stored; alpha untouched). Not something the reference
 * does: it is the checker for the product's multi-GPU sample partition, whose ranks each own one residue class
 * of the sample index and merge sums. */
PTO_API uint64_t pto_render_sum(const PtoScene* sc, const PtoUniforms* u, int first_sample, int stride, int count,
                                float* rgba_sum, int x0, int y0, int w, int h) {
    FpGuard g = fp_enter();
    uint64_t rays = 0;
    for (int y = y0; y < y0 + h; y++) {
        for (int x = x0; x < x0 + w; x++) {
            float* px = rgba_sum + 4 * ((size_t)y * u->width + x);
            for (int s = 0; s < count; s++) {
                v3 a = trace_pixel(sc, u, x, y, (float)(first_sample + s * stride), &rays, 0);
                px[0] += a.x; px[1] += a.y; px[2] += a.z;
            }
        }
    }
    fp_leave(g);
    return rays;
}

/* Present pass, /root/reference/data/shaders/copy.fs:12-20 as llvmpipe renders it into an 8-bit target:
 * pow(rgb, 0.454545) with the polynomial pow above, then float -> UNORM8 = clamp to [0,1], times 255, round to nearest
 * even; alpha 1 -> 255. (The unused u_InvNumSamples uniform has no effect, copy.fs:8,16.) */
PTO_API void pto_present(const float* rgba, int num_pixels, uint8_t* out_rgba8) {
    FpGuard g = fp_enter();
    for (int i = 0; i < num_pixels; i++) {
        for (int c = 0; c < 3; c++) {
            float v = lp_pow(rgba[4 * i + c], 0.454545f);
            v = fminf(fmaxf(v, 0.0f), 1.0f);
            out_rgba8[4 * i + c] = (uint8_t)nearbyintf(v * 255.0f);
        }
        out_rgba8[4 * i + 3] = 255;
    }
    fp_leave(g);
}

/* n successive rand01() draws (fs:29-42) at pixel (px,py) in pass `samples`. */
PTO_API void pto_rng_stream(int px, int py, float samples, int n, float* out) {
    FpGuard g = fp_enter();
    PixelCtx p;
    p.fcx = (float)px + 0.5f; p.fcy = (float)py + 0.5f; p.samples = samples; p.rs_x = 0.f; p.rs_y = 0.f;
    for (int i = 0; i < n; i++) out[i] = rng_rand01(&p);
    fp_leave(g);
}

/* Camera set-up as the reference host computes it (PathTracer.cpp:691-716):
 * out12 = position, lowerLeftCorner, horizontal, vertical. */
PTO_API void pto_camera(float fov_deg, float width, float height, const float eye[3], const float target[3],
                        const float up[3], float out12[12]) {
    float aspect = width / height;                                   /* :712 */
    float ex = eye[0], ey = eye[1], ez = eye[2];
    float wx = target[0] - ex, wy = target[1] - ey, wz = target[2] - ez;
    float il = 1.f / sqrtf(wx * wx + wy * wy + wz * wz);             /* float3::length :59, normalize :68 */
    wx *= il; wy *= il; wz *= il;
    float ux = up[1] * wz - up[2] * wy, uy = up[2] * wx - up[0] * wz, uz = up[0] * wy - up[1] * wx; /* cross(up,w) :62 */
    il = 1.f / sqrtf(ux * ux + uy * uy + uz * uz);
    ux *= il; uy *= il; uz *= il;
    float vx = wy * uz - wz * uy, vy = wz * ux - wx * uz, vz = wx * uy - wy * ux;                 /* cross(w,u) */
    float theta = (float)tan((double)fov_deg * M_PI / (double)360.0f);   /* :699 */
    float hh = theta, hw = theta * aspect;
    /* position - hw*u*1 - hh*v*1 + w*1, left to right (:704) */
    out12[0] = ex; out12[1] = ey; out12[2] = ez;
    out12[3] = ((ex - hw * ux * 1.f) - hh * vx * 1.f) + wx * 1.f;
    out12[4] = ((ey - hw * uy * 1.f) - hh * vy * 1.f) + wy * 1.f;
    out12[5] = ((ez - hw * uz * 1.f) - hh * vz * 1.f) + wz * 1.f;
    float h2 = 2.f * hw * 1.f, v2 = 2.f * hh * 1.f;                  /* :705-706 */
    out12[6] = h2 * ux; out12[7] = h2 * uy; out12[8] = h2 * uz;
    out12[9] = v2 * vx; out12[10] = v2 * vy; out12[11] = v2 * vz;
}
