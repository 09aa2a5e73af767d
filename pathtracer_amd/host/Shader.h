// Program object with the reference's `Shader` surface (Shader.h:5-41 in the reference), so that host code written
// against it -- loadVertexShader / loadFragmentShader / createProgram / getProgram / destroyProgram -- keeps working
// when the GL program behind it is replaced by the HIP kernels of libpt_hip.so.
//
// What changes underneath:
//   * there is no GLSL compiler. loadFragmentShader("…/pathtracing.fs") selects the path-tracing kernel family.
//     If the named file is readable its text is scanned for which closest-hit call is live in main() -- the
//     reference switches between its sphere scene and its mesh by commenting one of two lines -- so editing the
//     shader still switches the program. If the file is absent the kind set by SetDefaultKind() is used.
//   * "copy.fs" names the present pass and "*.vs" the full-screen-quad vertex stage; both are accepted and need no
//     device code here.
//   * getProgram() returns the handle from pt_program_create(); pass it to pt_use_program() where the reference
//     called glUseProgram().
// The device context plays the role of the current GL context: set it with Shader::SetContext() on the thread that makes the
// calls (it is kept per thread, as a GL context is current per thread).
#pragma once

#include <cstdint>

struct PtContext;

class Shader
{
public:
    enum Signature {
        ATTRIBUTE_LESS = -1,
        STANDARD = 0,
        COMPATIBLE = 1,
        MAX
    };
    // What a loaded fragment stage stands for.
    enum Stage { STAGE_NONE = 0, STAGE_PATHTRACE_WORLD, STAGE_PATHTRACE_MESH, STAGE_PRESENT };

    Shader() : m_ProgramObject(0), m_VertexShader(0), m_FragmentShader(0) {}
    ~Shader() {}

    bool loadVertexShader(const char* filename);
    bool loadFragmentShader(const char* filename);
    bool compileVertexShader(char* buffer);
    bool compileFragmentShader(char* buffer);
    void compileShader(uint32_t type, const char* code);

    bool createProgram(int signature = STANDARD);
    void destroyProgram();

    inline uint32_t getProgram() { return m_ProgramObject; }
    inline Stage getStage() const { return static_cast<Stage>(m_FragmentShader); }

    static void SetVersionString(int32_t version);
    // The device context programs are created in (the "current GL context" of this interface).
    static void SetContext(PtContext* ctx);
    static PtContext* GetContext();
    // Kind used when the fragment shader file cannot be read: STAGE_PATHTRACE_WORLD (the reference shader as
    // committed) unless changed.
    static void SetDefaultKind(Stage stage);

public:
    static char g_VersionString[16];
    static uint32_t g_VersionStringSize;

    // Returns a new[]-allocated, NUL-terminated copy of the file with the version string in front; the caller
    // delete[]s it. Returns nullptr when the file cannot be opened.
    static char* LoadFileToMemory(const char* filename);
    // Classifies path-tracing shader text: which closest-hit call is live in main().
    static Stage ClassifySource(const char* code);

    // type values accepted by compileShader (numerically the GL enums, so existing callers keep working)
    static const uint32_t FRAGMENT_STAGE = 0x8B30;
    static const uint32_t VERTEX_STAGE = 0x8B31;

private:
    uint32_t m_ProgramObject;
    uint32_t m_VertexShader;    // non-zero once a vertex stage was accepted
    uint32_t m_FragmentShader;  // Stage of the accepted fragment stage
};
