#include "Shader.h"

#include <atomic>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>

#include "pt_hip.h"

char Shader::g_VersionString[16];
uint32_t Shader::g_VersionStringSize = 0;

namespace {
// the device context calls are made on: per thread, like a GL context that is current on the calling thread (the reference makes
// its context current on the main thread, PathTracer.cpp:1095); a host that drives several GPUs uses a thread per context
thread_local PtContext* g_context = nullptr;
std::atomic<Shader::Stage> g_default_kind{Shader::STAGE_PATHTRACE_WORLD};

const char* base_name(const char* path) {
    const char* b = path;
    for (const char* p = path; *p; ++p)
        if (*p == '/' || *p == '\\') b = p + 1;
    return b;
}

bool ends_with(const char* s, const char* suffix) {
    const size_t n = strlen(s), m = strlen(suffix);
    return n >= m && strcmp(s + n - m, suffix) == 0;
}

// Removes // and /* */ comments so that only live code is scanned.
std::string strip_comments(const char* code) {
    std::string out;
    for (const char* p = code; *p;) {
        if (p[0] == '/' && p[1] == '/') {
            while (*p && *p != '\n') ++p;
        } else if (p[0] == '/' && p[1] == '*') {
            p += 2;
            while (*p && !(p[0] == '*' && p[1] == '/')) ++p;
            if (*p) p += 2;
        } else {
            out.push_back(*p++);
        }
    }
    return out;
}
}  // namespace

void Shader::SetVersionString(int32_t version) {
    snprintf(g_VersionString, sizeof g_VersionString, "#version %d\n", version);
    g_VersionStringSize = static_cast<uint32_t>(strnlen(g_VersionString, sizeof g_VersionString));
}

void Shader::SetContext(PtContext* ctx) { g_context = ctx; }
PtContext* Shader::GetContext() { return g_context; }
void Shader::SetDefaultKind(Stage stage) { g_default_kind = stage; }

char* Shader::LoadFileToMemory(const char* filename) {
    FILE* file = fopen(filename, "rb");
    if (!file) return nullptr;
    fseek(file, 0, SEEK_END);
    const long len = ftell(file);
    rewind(file);
    const size_t total = g_VersionStringSize + static_cast<size_t>(len < 0 ? 0 : len);
    char* buffer = new char[total + 1];
    memcpy(buffer, g_VersionString, g_VersionStringSize);
    const size_t got = len > 0 ? fread(buffer + g_VersionStringSize, 1, static_cast<size_t>(len), file) : 0;
    buffer[g_VersionStringSize + got] = '\0';
    fclose(file);
    return buffer;
}

Shader::Stage Shader::ClassifySource(const char* code) {
    const std::string live = strip_comments(code);
    // the live call inside main(): search after the last "void main"
    const size_t m = live.rfind("void main");
    const std::string body = m == std::string::npos ? live : live.substr(m);
    const bool mesh = body.find("IntersectMesh(") != std::string::npos;
    const bool world = body.find("IntersectWorld(") != std::string::npos;
    if (mesh && !world) return STAGE_PATHTRACE_MESH;
    if (world && !mesh) return STAGE_PATHTRACE_WORLD;
    return STAGE_NONE;
}

void Shader::compileShader(uint32_t type, const char* code) {
    if (type == VERTEX_STAGE) {
        m_VertexShader = 1;  // the full-screen quad is implicit in the kernel launch
        return;
    }
    if (type != FRAGMENT_STAGE) return;
    Stage s = code ? ClassifySource(code) : STAGE_NONE;
    if (s == STAGE_NONE && code && strstr(code, "pow(source")) s = STAGE_PRESENT;
    m_FragmentShader = static_cast<uint32_t>(s);
    if (s == STAGE_NONE) std::cout << "Shader: fragment source is not a stage this library implements" << std::endl;
}

bool Shader::loadVertexShader(const char* filename) {
    (void)filename;
    m_VertexShader = 1;
    return true;
}

bool Shader::loadFragmentShader(const char* filename) {
    const char* base = base_name(filename);
    m_FragmentShader = STAGE_NONE;
    if (strcmp(base, "pathtracing.fs") == 0) {
        char* text = LoadFileToMemory(filename);
        Stage s = text ? ClassifySource(text) : STAGE_NONE;
        delete[] text;
        m_FragmentShader = static_cast<uint32_t>(s != STAGE_NONE ? s : g_default_kind.load());
    } else if (strcmp(base, "copy.fs") == 0) {
        m_FragmentShader = STAGE_PRESENT;
    } else if (ends_with(base, ".fs")) {
        std::cout << filename << std::endl << "Shader: no HIP implementation for this fragment shader" << std::endl;
    }
    return m_FragmentShader != 0;
}

bool Shader::compileVertexShader(char* buffer) {
    compileShader(VERTEX_STAGE, buffer);
    return m_VertexShader != 0;
}

bool Shader::compileFragmentShader(char* buffer) {
    compileShader(FRAGMENT_STAGE, buffer);
    return m_FragmentShader != 0;
}

bool Shader::createProgram(int signature) {
    (void)signature;
    m_ProgramObject = 0;
    const Stage s = getStage();
    if (s == STAGE_PATHTRACE_WORLD || s == STAGE_PATHTRACE_MESH) {
        if (!g_context) {
            std::cerr << "Shader::createProgram: no device context (call Shader::SetContext)" << std::endl;
            return true;
        }
        uint32_t handle = 0;
        const PtStatus st = pt_program_create(g_context, s == STAGE_PATHTRACE_MESH ? PT_PROGRAM_MESH : PT_PROGRAM_WORLD, &handle);
        if (st != PT_OK) std::cerr << pt_last_error() << std::endl;
        m_ProgramObject = handle;
    }
    // Like the reference, the call returns true whatever happened; failures go to stderr, and getProgram() stays 0.
    return true;
}

void Shader::destroyProgram() {
    if (m_ProgramObject && g_context) pt_program_destroy(g_context, m_ProgramObject);
    m_ProgramObject = 0;
    m_VertexShader = 0;
    m_FragmentShader = 0;
}
