// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or shipped with the product.
//
// Headless runner for the reference's GLSL path tracer on Mesa llvmpipe (CPU), used in the
// build container to produce golden vectors and the "reference on CPU" timing baseline.
//
// What it does: opens Mesa's software rasteriser (swrast_dri.so) directly through the DRI
// SWRast interface -- no X server, no GLX, no EGL -- creates an OpenGL core context, loads the
// reference's own vertex + fragment shader text from the reference checkout AT RUN TIME
// (nothing of it is stored in this repository), optionally applies a small, recorded list of
// line edits (see edits/*.txt) and then repeats the reference host's GL object set-up and
// per-frame dispatch:
//   context state          /root/reference/PathTracer.cpp:1108-1112
//   version string prefix  /root/reference/Shader.cpp:46-55, 148-162
//   ping-pong RGBA32F FBOs /root/reference/PathTracer.cpp:1136-1152
//   scene textures         /root/reference/PathTracer.cpp:1253-1288
//   per-frame dispatch     /root/reference/PathTracer.cpp:1370-1403
// and finally reads the accumulation texture back with glReadPixels(GL_RGBA, GL_FLOAT).
//
// The job is described by a small "key value..." text file; see parse_job() below.
//
// Build: see ../Makefile (g++ lp_harness.cpp -ldl). Output binary goes to oracle/_ref/.

#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>
#include <dlfcn.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

// ---------------------------------------------------------------- GL entry points
#define GL_FUNCS(X)                                                   \
    X(PFNGLGETSTRINGPROC, glGetString)                                \
    X(PFNGLGETERRORPROC, glGetError)                                  \
    X(PFNGLGETINTEGERVPROC, glGetIntegerv)                            \
    X(PFNGLCLEARCOLORPROC, glClearColor)                              \
    X(PFNGLCLEARPROC, glClear)                                        \
    X(PFNGLDISABLEPROC, glDisable)                                    \
    X(PFNGLENABLEPROC, glEnable)                                      \
    X(PFNGLSCISSORPROC, glScissor)                                    \
    X(PFNGLDEPTHMASKPROC, glDepthMask)                                \
    X(PFNGLVIEWPORTPROC, glViewport)                                  \
    X(PFNGLFINISHPROC, glFinish)                                      \
    X(PFNGLREADPIXELSPROC, glReadPixels)                              \
    X(PFNGLREADBUFFERPROC, glReadBuffer)                              \
    X(PFNGLDRAWARRAYSPROC, glDrawArrays)                              \
    X(PFNGLGENTEXTURESPROC, glGenTextures)                            \
    X(PFNGLBINDTEXTUREPROC, glBindTexture)                            \
    X(PFNGLTEXPARAMETERIPROC, glTexParameteri)                        \
    X(PFNGLTEXIMAGE2DPROC, glTexImage2D)                              \
    X(PFNGLACTIVETEXTUREPROC, glActiveTexture)                        \
    X(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers)                    \
    X(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer)                    \
    X(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D)          \
    X(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus)      \
    X(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays)                    \
    X(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray)                    \
    X(PFNGLCREATESHADERPROC, glCreateShader)                          \
    X(PFNGLSHADERSOURCEPROC, glShaderSource)                          \
    X(PFNGLCOMPILESHADERPROC, glCompileShader)                        \
    X(PFNGLGETSHADERIVPROC, glGetShaderiv)                            \
    X(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog)                  \
    X(PFNGLCREATEPROGRAMPROC, glCreateProgram)                        \
    X(PFNGLATTACHSHADERPROC, glAttachShader)                          \
    X(PFNGLLINKPROGRAMPROC, glLinkProgram)                            \
    X(PFNGLGETPROGRAMIVPROC, glGetProgramiv)                          \
    X(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog)                \
    X(PFNGLUSEPROGRAMPROC, glUseProgram)                              \
    X(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation)              \
    X(PFNGLUNIFORM3FVPROC, glUniform3fv)                              \
    X(PFNGLUNIFORM1IPROC, glUniform1i)                                \
    X(PFNGLUNIFORM1FPROC, glUniform1f)

#define DECL(T, n) static T n = nullptr;
GL_FUNCS(DECL)
#undef DECL

static void die(const char* msg) {
    fprintf(stderr, "lp_harness: %s\n", msg);
    exit(2);
}

// ---------------------------------------------------------------- swrast loader callbacks
// The window-system drawable is never shown; all rendering goes to FBOs. The loader only has
// to answer "how big is the drawable" and swallow image transfers.
static void cb_getDrawableInfo(__DRIdrawable*, int* x, int* y, int* w, int* h, void*) {
    *x = 0; *y = 0; *w = 16; *h = 16;
}
static void cb_putImage(__DRIdrawable*, int, int, int, int, int, char*, void*) {}
static void cb_getImage(__DRIdrawable*, int, int, int w, int h, char* data, void*) {
    memset(data, 0, (size_t)w * h * 4);
}
static void cb_putImage2(__DRIdrawable*, int, int, int, int, int, int, char*, void*) {}
static void cb_getImage2(__DRIdrawable*, int, int, int, int h, int stride, char* data, void*) {
    memset(data, 0, (size_t)stride * h);
}

struct GLContext {
    void* dri = nullptr;
    void* glapi = nullptr;
    const __DRIcoreExtension* core = nullptr;
    const __DRIswrastExtension* swrast = nullptr;
    __DRIscreen* screen = nullptr;
    __DRIcontext* ctx = nullptr;
    __DRIdrawable* drawable = nullptr;
};

static GLContext open_llvmpipe() {
    GLContext g;
    const char* dri_path = getenv("LP_SWRAST_DRI");
    if (!dri_path) dri_path = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
    g.glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!g.glapi) die("cannot dlopen libglapi.so.0");
    g.dri = dlopen(dri_path, RTLD_NOW | RTLD_GLOBAL);
    if (!g.dri) { fprintf(stderr, "%s\n", dlerror()); die("cannot dlopen swrast_dri.so"); }

    typedef const __DRIextension** (*GetExtFn)(void);
    GetExtFn getext = (GetExtFn)dlsym(g.dri, "__driDriverGetExtensions_swrast");
    if (!getext) die("no __driDriverGetExtensions_swrast");
    const __DRIextension** drv = getext();
    for (int i = 0; drv[i]; i++) {
        if (!strcmp(drv[i]->name, __DRI_CORE)) g.core = (const __DRIcoreExtension*)drv[i];
        if (!strcmp(drv[i]->name, __DRI_SWRAST)) g.swrast = (const __DRIswrastExtension*)drv[i];
    }
    if (!g.core || !g.swrast || g.swrast->base.version < 4) die("DRI_Core / DRI_SWRast v4 not found");

    static __DRIswrastLoaderExtension loader;
    memset(&loader, 0, sizeof(loader));
    loader.base.name = __DRI_SWRAST_LOADER;
    loader.base.version = 3;
    loader.getDrawableInfo = cb_getDrawableInfo;
    loader.putImage = cb_putImage;
    loader.getImage = cb_getImage;
    loader.putImage2 = cb_putImage2;
    loader.getImage2 = cb_getImage2;
    static const __DRIextension* loader_exts[] = {&loader.base, nullptr};

    const __DRIconfig** configs = nullptr;
    g.screen = g.swrast->createNewScreen2(0, loader_exts, drv, &configs, nullptr);
    if (!g.screen || !configs || !configs[0]) die("createNewScreen2 failed");

    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 1};
    unsigned err = 0;
    g.ctx = g.swrast->createContextAttribs(g.screen, __DRI_API_OPENGL_CORE, configs[0], nullptr, 2,
                                           attribs, &err, nullptr);
    if (!g.ctx) die("createContextAttribs failed");
    g.drawable = g.swrast->createNewDrawable(g.screen, configs[0], nullptr);
    if (!g.drawable) die("createNewDrawable failed");
    if (!g.core->bindContext(g.ctx, g.drawable, g.drawable)) die("bindContext failed");

    typedef void* (*GetProc)(const char*);
    GetProc gpa = (GetProc)dlsym(g.glapi, "_glapi_get_proc_address");
    if (!gpa) die("no _glapi_get_proc_address");
#define LOAD(T, n) n = (T)gpa(#n); if (!n) die("missing GL entry point " #n);
    GL_FUNCS(LOAD)
#undef LOAD
    return g;
}

// ---------------------------------------------------------------- job description
struct Job {
    std::string shader_dir = "/root/reference/data/shaders";
    std::string vs_name = "postprocess.vs";
    std::string fs_name = "pathtracing.fs";
    std::string fs_override;            // full path to a non-reference fragment shader (probes)
    std::string edits;                  // path of an edit script, may be empty
    int width = 0, height = 0;
    int depth = 10;
    int first_sample = 1;               // u_Samples of the first pass (1-based, PathTracer.cpp:1305,1378)
    int num_samples = 1;
    int warmup = 0;                     // untimed, discarded passes (JIT)
    float camera[12] = {0};             // position, lowerLeftCorner, horizontal, vertical
    int num_triangles = 0;
    int tex_width = 1024;               // reference: 1024 (PathTracer.cpp:1274,1287)
    int index_components = 3;           // reference: RGB32UI (PathTracer.cpp:1274); 4 = RGBA32UI carrying a material id
    std::string spheres, materials, vertices, indices, init_accum, out;
    std::string present_out;            // if set: also run the reference's present pass (copy.fs) into an RGBA8 target
    std::vector<int> dump_after;        // also dump out.<k>.f32 after these 1-based pass counts
    int crop[4] = {0, 0, 0, 0};         // x0 y0 w h; w==0 means full frame
    bool print_source = false;
};

static std::vector<char> read_file(const std::string& p, bool must = true) {
    std::ifstream f(p, std::ios::binary);
    if (!f) {
        if (must) { fprintf(stderr, "lp_harness: cannot read %s\n", p.c_str()); exit(2); }
        return {};
    }
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static Job parse_job(const char* path) {
    Job j;
    std::ifstream f(path);
    if (!f) die("cannot open job file");
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string k;
        if (!(ss >> k) || k[0] == '#') continue;
        if (k == "shader_dir") ss >> j.shader_dir;
        else if (k == "vs_name") ss >> j.vs_name;
        else if (k == "fs_name") ss >> j.fs_name;
        else if (k == "fs_override") ss >> j.fs_override;
        else if (k == "edits") ss >> j.edits;
        else if (k == "width") ss >> j.width;
        else if (k == "height") ss >> j.height;
        else if (k == "depth") ss >> j.depth;
        else if (k == "first_sample") ss >> j.first_sample;
        else if (k == "num_samples") ss >> j.num_samples;
        else if (k == "warmup") ss >> j.warmup;
        else if (k == "camera") { for (int i = 0; i < 12; i++) { std::string t; ss >> t; j.camera[i] = strtof(t.c_str(), nullptr); } }
        else if (k == "num_triangles") ss >> j.num_triangles;
        else if (k == "tex_width") ss >> j.tex_width;
        else if (k == "index_components") ss >> j.index_components;
        else if (k == "spheres") ss >> j.spheres;
        else if (k == "materials") ss >> j.materials;
        else if (k == "vertices") ss >> j.vertices;
        else if (k == "indices") ss >> j.indices;
        else if (k == "init_accum") ss >> j.init_accum;
        else if (k == "out") ss >> j.out;
        else if (k == "present_out") ss >> j.present_out;
        else if (k == "dump_after") { int v; while (ss >> v) j.dump_after.push_back(v); }
        else if (k == "crop") { for (int i = 0; i < 4; i++) ss >> j.crop[i]; }
        else if (k == "print_source") j.print_source = true;
        else { fprintf(stderr, "lp_harness: unknown key %s\n", k.c_str()); exit(2); }
    }
    if (j.width <= 0 || j.height <= 0) die("job needs width/height");
    return j;
}

// ---------------------------------------------------------------- recorded source edits
// Edit script: one edit per line, TAB-separated:  <1-based line number> <old> <new>
// <old> must occur in that line of the shader file as loaded; it is replaced once by <new>.
// "\n" inside <new> becomes a newline. Lines starting with '#' are comments. A mismatch is fatal,
// so a changed reference cannot silently produce a different oracle.
static std::string apply_edits(const std::string& src, const std::string& script_path) {
    std::vector<std::string> lines;
    {
        std::string cur;
        for (char c : src) { if (c == '\n') { lines.push_back(cur); cur.clear(); } else cur.push_back(c); }
        lines.push_back(cur);
    }
    std::ifstream f(script_path);
    if (!f) die("cannot open edit script");
    std::string e;
    while (std::getline(f, e)) {
        if (e.empty() || e[0] == '#') continue;
        size_t t1 = e.find('\t');
        size_t t2 = e.find('\t', t1 + 1);
        if (t1 == std::string::npos || t2 == std::string::npos) die("malformed edit line");
        int ln = atoi(e.substr(0, t1).c_str());
        std::string olds = e.substr(t1 + 1, t2 - t1 - 1);
        std::string news = e.substr(t2 + 1);
        for (size_t p = 0; (p = news.find("\\n", p)) != std::string::npos;) news.replace(p, 2, "\n");
        if (ln < 1 || ln > (int)lines.size()) die("edit line number out of range");
        size_t pos = lines[ln - 1].find(olds);
        if (pos == std::string::npos) {
            fprintf(stderr, "lp_harness: edit mismatch at line %d: expected to find [%s]\n", ln, olds.c_str());
            exit(2);
        }
        lines[ln - 1].replace(pos, olds.size(), news);
    }
    std::string out;
    for (size_t i = 0; i < lines.size(); i++) { out += lines[i]; if (i + 1 < lines.size()) out += '\n'; }
    return out;
}

static GLuint compile(GLenum type, const std::string& src, const char* what) {
    GLuint s = glCreateShader(type);
    const char* p = src.c_str();
    glShaderSource(s, 1, &p, nullptr);
    glCompileShader(s);
    GLint ok = 0, len = 0;
    glGetShaderiv(s, GL_COMPILE_STATUS, &ok);
    glGetShaderiv(s, GL_INFO_LOG_LENGTH, &len);
    if (!ok) {
        std::vector<char> log(len + 1);
        glGetShaderInfoLog(s, len, nullptr, log.data());
        fprintf(stderr, "lp_harness: %s failed to compile:\n%s\n", what, log.data());
        exit(3);
    }
    return s;
}

static std::vector<float> load_f32(const std::string& p, size_t min_count) {
    std::vector<float> v;
    if (!p.empty()) {
        std::vector<char> b = read_file(p);
        v.resize(b.size() / 4);
        memcpy(v.data(), b.data(), v.size() * 4);
    }
    if (v.size() < min_count) v.resize(min_count, 0.0f);  // zero padding instead of the reference's over-read
    return v;
}
static std::vector<uint32_t> load_u32(const std::string& p, size_t min_count) {
    std::vector<uint32_t> v;
    if (!p.empty()) {
        std::vector<char> b = read_file(p);
        v.resize(b.size() / 4);
        memcpy(v.data(), b.data(), v.size() * 4);
    }
    if (v.size() < min_count) v.resize(min_count, 0u);
    return v;
}

static void write_f32(const std::string& p, const std::vector<float>& v) {
    FILE* f = fopen(p.c_str(), "wb");
    if (!f) die("cannot write output");
    fwrite(v.data(), 4, v.size(), f);
    fclose(f);
}

#define GLCHECK(where) do { GLenum e_ = glGetError(); if (e_ != GL_NO_ERROR) { fprintf(stderr, "lp_harness: GL error 0x%x at %s\n", e_, where); exit(4); } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) die("usage: lp_harness <job.txt>");
    Job job = parse_job(argv[1]);
    GLContext g = open_llvmpipe();

    const char* renderer = (const char*)glGetString(GL_RENDERER);
    const char* version = (const char*)glGetString(GL_VERSION);
    const char* glsl = (const char*)glGetString(GL_SHADING_LANGUAGE_VERSION);

    // PathTracer.cpp:1108-1112
    glClearColor(0.f, 0.f, 0.f, 1.f);
    glDisable(GL_BLEND);
    glDisable(GL_CULL_FACE);
    glDisable(GL_DEPTH_TEST);
    glDepthMask(GL_FALSE);

    // PathTracer.cpp:1117-1120 + Shader.cpp:46-55: "#version <100*GLSL version>\n" is prepended.
    float glsl_ver = 0.f;
    sscanf(glsl, "%f", &glsl_ver);
    char verstr[32];
    snprintf(verstr, sizeof verstr, "#version %d\n", (int)(unsigned)(100.f * glsl_ver));

    std::vector<char> vs_raw = read_file(job.shader_dir + "/" + job.vs_name);
    std::string fs_path = job.fs_override.empty() ? job.shader_dir + "/" + job.fs_name : job.fs_override;
    std::vector<char> fs_raw = read_file(fs_path);
    std::string vs_src = std::string(verstr) + std::string(vs_raw.begin(), vs_raw.end());
    std::string fs_body(fs_raw.begin(), fs_raw.end());
    if (!job.edits.empty()) fs_body = apply_edits(fs_body, job.edits);
    std::string fs_src = std::string(verstr) + fs_body;
    if (job.print_source) { fputs(fs_src.c_str(), stderr); }

    GLuint vs = compile(GL_VERTEX_SHADER, vs_src, "vertex shader");
    GLuint fs = compile(GL_FRAGMENT_SHADER, fs_src, "fragment shader");
    GLuint prog = glCreateProgram();
    glAttachShader(prog, vs);
    glAttachShader(prog, fs);
    glLinkProgram(prog);
    GLint linked = 0;
    glGetProgramiv(prog, GL_LINK_STATUS, &linked);
    if (!linked) {
        char log[4096]; GLsizei n = 0;
        glGetProgramInfoLog(prog, sizeof log, &n, log);
        fprintf(stderr, "lp_harness: link failed:\n%s\n", log);
        return 3;
    }

    const int W = job.width, H = job.height;
    // PathTracer.cpp:1136-1152 -- two RGBA32F targets, NEAREST, CLAMP_TO_EDGE, cleared to (0,0,0,1)
    GLuint fbo[2], tex[2];
    glGenFramebuffers(2, fbo);
    glGenTextures(2, tex);
    std::vector<float> init;
    if (!job.init_accum.empty()) init = load_f32(job.init_accum, (size_t)W * H * 4);
    for (int i = 0; i < 2; i++) {
        glBindTexture(GL_TEXTURE_2D, tex[i]);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
        glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, W, H, 0, GL_RGBA, GL_FLOAT, init.empty() ? nullptr : init.data());
        glBindFramebuffer(GL_FRAMEBUFFER, fbo[i]);
        glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex[i], 0);
        if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) die("FBO incomplete");
        if (init.empty()) glClear(GL_COLOR_BUFFER_BIT);
    }
    GLuint vao;
    glGenVertexArrays(1, &vao);
    GLCHECK("targets");

    // PathTracer.cpp:1253-1288 -- scene textures
    const int TW = job.tex_width;
    std::vector<float> spheres = load_f32(job.spheres, 40);
    std::vector<float> materials = load_f32(job.materials, 40);
    std::vector<float> vertices = load_f32(job.vertices, (size_t)TW * 3);
    const int IC = job.index_components == 4 ? 4 : 3;
    std::vector<uint32_t> indices = load_u32(job.indices, (size_t)TW * IC);
    if ((size_t)job.num_triangles > (size_t)TW) die("num_triangles exceeds tex_width");
    if (vertices.size() > (size_t)TW * 3) die("vertex data exceeds tex_width texels");

    GLuint t_sph, t_mat, t_idx, t_vtx;
    glGenTextures(1, &t_sph);
    glBindTexture(GL_TEXTURE_2D, t_sph);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, 10, 1, 0, GL_RGBA, GL_FLOAT, spheres.data());
    glGenTextures(1, &t_mat);
    glBindTexture(GL_TEXTURE_2D, t_mat);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, 10, 1, 0, GL_RGBA, GL_FLOAT, materials.data());
    glGenTextures(1, &t_idx);
    glBindTexture(GL_TEXTURE_RECTANGLE, t_idx);
    glTexParameteri(GL_TEXTURE_RECTANGLE, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_RECTANGLE, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_RECTANGLE, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_RECTANGLE, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    glTexImage2D(GL_TEXTURE_RECTANGLE, 0, IC == 4 ? GL_RGBA32UI : GL_RGB32UI, TW, 1, 0, IC == 4 ? GL_RGBA_INTEGER : GL_RGB_INTEGER,
                 GL_UNSIGNED_INT, indices.data());
    GLCHECK("indices texture");
    glGenTextures(1, &t_vtx);
    glBindTexture(GL_TEXTURE_RECTANGLE, t_vtx);
    glTexParameteri(GL_TEXTURE_RECTANGLE, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_RECTANGLE, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_RECTANGLE, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_RECTANGLE, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    glTexImage2D(GL_TEXTURE_RECTANGLE, 0, GL_RGB32F, TW, 1, 0, GL_RGB, GL_FLOAT, vertices.data());
    GLCHECK("vertices texture");

    glViewport(0, 0, W, H);
    int cx = job.crop[0], cy = job.crop[1], cw = job.crop[2], ch = job.crop[3];
    if (cw > 0) { glEnable(GL_SCISSOR_TEST); glScissor(cx, cy, cw, ch); }
    else { cx = 0; cy = 0; cw = W; ch = H; }

    int cur = 0;
    std::vector<double> pass_ms;
    std::vector<float> out((size_t)cw * ch * 4);
    auto read_back = [&](int which) {
        glBindFramebuffer(GL_FRAMEBUFFER, fbo[which]);
        glReadBuffer(GL_COLOR_ATTACHMENT0);
        glReadPixels(cx, cy, cw, ch, GL_RGBA, GL_FLOAT, out.data());
        GLCHECK("readback");
    };

    const int total = job.warmup + job.num_samples;
    for (int p = 0; p < total; p++) {
        const bool timed = p >= job.warmup;
        // warm-up passes render pass `first_sample` into a scratch target state and are discarded:
        // they run with the same uniforms but their result is overwritten because cur is not flipped.
        const int samples = job.first_sample + (timed ? p - job.warmup : 0);
        const int nxt = cur ^ 1;
        auto t1 = std::chrono::steady_clock::now();
        // PathTracer.cpp:1370-1403
        glUseProgram(prog);
        glUniform3fv(glGetUniformLocation(prog, "u_CameraPosition"), 1, &job.camera[0]);
        glUniform3fv(glGetUniformLocation(prog, "u_CameraLowerLeftCorner"), 1, &job.camera[3]);
        glUniform3fv(glGetUniformLocation(prog, "u_CameraHorizontal"), 1, &job.camera[6]);
        glUniform3fv(glGetUniformLocation(prog, "u_CameraVertical"), 1, &job.camera[9]);
        glUniform1i(glGetUniformLocation(prog, "u_Depth"), job.depth);
        glUniform1f(glGetUniformLocation(prog, "u_Samples"), (float)samples);
        glUniform1i(glGetUniformLocation(prog, "u_NumTriangles"), job.num_triangles);
        glBindFramebuffer(GL_FRAMEBUFFER, fbo[cur]);
        glActiveTexture(GL_TEXTURE4);
        glBindTexture(GL_TEXTURE_2D, t_mat);
        glUniform1i(glGetUniformLocation(prog, "u_MaterialsTexture"), 4);
        glActiveTexture(GL_TEXTURE3);
        glBindTexture(GL_TEXTURE_RECTANGLE, t_idx);
        glUniform1i(glGetUniformLocation(prog, "u_IndicesTexture"), 3);
        glActiveTexture(GL_TEXTURE2);
        glBindTexture(GL_TEXTURE_RECTANGLE, t_vtx);
        glUniform1i(glGetUniformLocation(prog, "u_VerticesTexture"), 2);
        glActiveTexture(GL_TEXTURE1);
        glBindTexture(GL_TEXTURE_2D, t_sph);
        glUniform1i(glGetUniformLocation(prog, "u_SpheresTexture"), 1);
        glActiveTexture(GL_TEXTURE0);
        glBindTexture(GL_TEXTURE_2D, tex[nxt]);
        glUniform1i(glGetUniformLocation(prog, "u_Texture"), 0);
        glBindVertexArray(vao);
        glDrawArrays(GL_TRIANGLE_STRIP, 0, 4);
        glFinish();
        auto t2 = std::chrono::steady_clock::now();
        if (!timed) continue;
        pass_ms.push_back(std::chrono::duration<double, std::milli>(t2 - t1).count());
        const int done = p - job.warmup + 1;
        for (int d : job.dump_after) {
            if (d == done && !job.out.empty()) {
                read_back(cur);
                write_f32(job.out + "." + std::to_string(d) + ".f32", out);
            }
        }
        cur ^= 1;  // PathTracer.cpp:1493
    }
    GLCHECK("render");
    if (!job.out.empty()) {
        read_back(cur ^ 1);  // the target written by the last pass
        write_f32(job.out, out);
    }

    // Present pass as the reference host issues it (PathTracer.cpp:1473-1491): copy.fs over a full-screen quad with the
    // last result bound as u_Texture. The reference draws it into the window's back buffer; here the target is an
    // RGBA8 texture of the frame's size, read back as bytes.
    if (!job.present_out.empty()) {
        std::vector<char> cfs_raw = read_file(job.shader_dir + "/copy.fs");
        GLuint cfs = compile(GL_FRAGMENT_SHADER, std::string(verstr) + std::string(cfs_raw.begin(), cfs_raw.end()), "copy.fs");
        GLuint cprog = glCreateProgram();
        glAttachShader(cprog, vs);
        glAttachShader(cprog, cfs);
        glLinkProgram(cprog);
        GLint ok = 0;
        glGetProgramiv(cprog, GL_LINK_STATUS, &ok);
        if (!ok) die("copy program failed to link");
        GLuint ptex, pfbo;
        glGenTextures(1, &ptex);
        glBindTexture(GL_TEXTURE_2D, ptex);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
        glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
        glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA8, W, H, 0, GL_RGBA, GL_UNSIGNED_BYTE, nullptr);
        glGenFramebuffers(1, &pfbo);
        glBindFramebuffer(GL_FRAMEBUFFER, pfbo);
        glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, ptex, 0);
        if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) die("present FBO incomplete");
        glDisable(GL_SCISSOR_TEST);
        glUseProgram(cprog);
        glActiveTexture(GL_TEXTURE0);
        glBindTexture(GL_TEXTURE_2D, tex[cur ^ 1]);
        glUniform1i(glGetUniformLocation(cprog, "u_Texture"), 0);
        glUniform1f(glGetUniformLocation(cprog, "u_InvNumSamples"), (float)(job.first_sample + job.num_samples - 1));
        glBindVertexArray(vao);
        glDrawArrays(GL_TRIANGLE_STRIP, 0, 4);
        glFinish();
        std::vector<unsigned char> bytes((size_t)W * H * 4);
        glReadBuffer(GL_COLOR_ATTACHMENT0);
        glReadPixels(0, 0, W, H, GL_RGBA, GL_UNSIGNED_BYTE, bytes.data());
        GLCHECK("present");
        FILE* pf = fopen(job.present_out.c_str(), "wb");
        if (!pf) die("cannot write present_out");
        fwrite(bytes.data(), 1, bytes.size(), pf);
        fclose(pf);
    }

    // one JSON line on stdout
    printf("{\"renderer\": \"%s\", \"gl_version\": \"%s\", \"glsl\": \"%s\", \"width\": %d, \"height\": %d, "
           "\"crop\": [%d, %d, %d, %d], \"passes\": %d, \"pass_ms\": [",
           renderer, version, glsl, W, H, cx, cy, cw, ch, job.num_samples);
    for (size_t i = 0; i < pass_ms.size(); i++) printf("%s%.4f", i ? ", " : "", pass_ms[i]);
    printf("]}\n");
    fflush(stdout);

    g.core->unbindContext(g.ctx);
    g.core->destroyDrawable(g.drawable);
    g.core->destroyContext(g.ctx);
    g.core->destroyScreen(g.screen);
    return 0;
}
