#include "ObjLoader.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace ptobj {
namespace {

inline bool is_blank(char c) { return c == ' ' || c == '\t'; }
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool at_line_end(char c) { return c == '\r' || c == '\n' || c == '\0'; }

// A cursor over one line of text.
struct Cursor {
    const char* p;
    void skip_blanks() { while (is_blank(*p)) ++p; }
    // end of the current whitespace-delimited word
    const char* word_end() const { return p + strcspn(p, " \t\r"); }
    bool starts(const char* kw) const {
        const size_t n = strlen(kw);
        return strncmp(p, kw, n) == 0 && is_blank(p[n]);
    }
    std::string word() {
        skip_blanks();
        const char* e = word_end();
        std::string s(p, e);
        p = e;
        return s;
    }
    float number(double fallback = 0.0) {
        skip_blanks();
        const char* e = word_end();
        const float f = ParseFloatText(p, e, fallback);
        p = e;
        return f;
    }
    void numbers(float* out, int n, double fallback = 0.0) {
        for (int i = 0; i < n; ++i) out[i] = number(fallback);
    }
    int integer() {
        skip_blanks();
        const int v = atoi(p);
        p = word_end();
        return v;
    }
};

// Reads a line ending in \n, \r\n, \r or end of file.
bool read_line(std::istream& in, std::string& line) {
    line.clear();
    if (in.peek() == std::char_traits<char>::eof()) return false;
    std::streambuf* sb = in.rdbuf();
    for (;;) {
        const int c = sb->sbumpc();
        if (c == '\n') return true;
        if (c == '\r') {
            if (sb->sgetc() == '\n') sb->sbumpc();
            return true;
        }
        if (c == std::char_traits<char>::eof()) {
            in.setstate(std::ios::eofbit);
            return true;
        }
        line.push_back(static_cast<char>(c));
    }
}

inline int resolve_index(int idx, int count) {
    if (idx > 0) return idx - 1;  // 1-based
    if (idx == 0) return 0;
    return count + idx;           // relative to the elements read so far
}

struct Corner {
    int v = -1, vt = -1, vn = -1;
};

// i, i/j, i//k or i/j/k
Corner parse_corner(Cursor& c, int nv, int nvn, int nvt) {
    Corner k;
    auto advance = [&c]() { c.p += strcspn(c.p, "/ \t\r"); };
    k.v = resolve_index(atoi(c.p), nv);
    advance();
    if (*c.p != '/') return k;
    ++c.p;
    if (*c.p == '/') {
        ++c.p;
        k.vn = resolve_index(atoi(c.p), nvn);
        advance();
        return k;
    }
    k.vt = resolve_index(atoi(c.p), nvt);
    advance();
    if (*c.p != '/') return k;
    ++c.p;
    k.vn = resolve_index(atoi(c.p), nvn);
    advance();
    return k;
}

void default_material(material_t& m) {
    m = material_t();
    for (int i = 0; i < 3; ++i) m.ambient[i] = m.diffuse[i] = m.specular[i] = m.transmittance[i] = m.emission[i] = 0.f;
    m.illum = 0;
    m.dissolve = 1.f;
    m.shininess = 1.f;
    m.ior = 1.f;
    m.roughness = m.metallic = m.sheen = m.clearcoat_thickness = m.clearcoat_roughness = 0.f;
    m.anisotropy = m.anisotropy_rotation = 0.f;
}

// The texture statements may carry "-option value..." groups before the file name; the name is the last word
// that is not part of an option.
std::string texture_name(Cursor c) {
    std::string name;
    while (!at_line_end(*c.p)) {
        c.skip_blanks();
        if (at_line_end(*c.p)) break;
        if (*c.p == '-') {
            const std::string opt = c.word();
            int nargs = 1;
            if (opt == "-o" || opt == "-s" || opt == "-t") nargs = 3;
            else if (opt == "-mm") nargs = 2;
            for (int i = 0; i < nargs; ++i) (void)c.word();
        } else {
            name = c.word();
        }
    }
    return name;
}

void load_mtl(std::istream& in, std::map<std::string, int>& by_name, std::vector<material_t>& out, std::string& warn) {
    material_t cur;
    default_material(cur);
    bool has_d = false, has_tr = false;
    std::string line;
    while (read_line(in, line)) {
        const size_t last = line.find_last_not_of(" \t");
        line = last == std::string::npos ? std::string() : line.substr(0, last + 1);
        if (line.empty()) continue;
        Cursor c{line.c_str()};
        c.skip_blanks();
        if (*c.p == '\0' || *c.p == '#') continue;

        if (c.starts("newmtl")) {
            if (!cur.name.empty()) {
                by_name.insert(std::make_pair(cur.name, static_cast<int>(out.size())));
                out.push_back(cur);
            }
            default_material(cur);
            has_d = has_tr = false;
            c.p += 7;
            cur.name = c.word();
            continue;
        }
        auto two = [&c](char a, char b) { return c.p[0] == a && c.p[1] == b && is_blank(c.p[2]); };
        if (two('K', 'a')) { c.p += 2; c.numbers(cur.ambient, 3); continue; }
        if (two('K', 'd')) { c.p += 2; c.numbers(cur.diffuse, 3); continue; }
        if (two('K', 's')) { c.p += 2; c.numbers(cur.specular, 3); continue; }
        if (two('K', 't') || two('T', 'f')) { c.p += 2; c.numbers(cur.transmittance, 3); continue; }
        if (two('N', 'i')) { c.p += 2; cur.ior = c.number(); continue; }
        if (two('K', 'e')) { c.p += 2; c.numbers(cur.emission, 3); continue; }
        if (two('N', 's')) { c.p += 2; cur.shininess = c.number(); continue; }
        if (c.starts("illum")) { c.p += 6; cur.illum = c.integer(); continue; }
        if (c.p[0] == 'd' && is_blank(c.p[1])) {
            c.p += 1;
            cur.dissolve = c.number();
            if (has_tr) warn += "WARN: Both `d` and `Tr` parameters defined for \"" + cur.name + "\". Use the value of `d` for dissolve.\n";
            has_d = true;
            continue;
        }
        if (two('T', 'r')) {
            c.p += 2;
            if (has_d) warn += "WARN: Both `d` and `Tr` parameters defined for \"" + cur.name + "\". Use the value of `d` for dissolve.\n";
            else cur.dissolve = 1.0f - c.number();
            has_tr = true;
            continue;
        }
        if (two('P', 'r')) { c.p += 2; cur.roughness = c.number(); continue; }
        if (two('P', 'm')) { c.p += 2; cur.metallic = c.number(); continue; }
        if (two('P', 's')) { c.p += 2; cur.sheen = c.number(); continue; }
        if (two('P', 'c')) { c.p += 2; cur.clearcoat_thickness = c.number(); continue; }
        if (c.starts("Pcr")) { c.p += 4; cur.clearcoat_roughness = c.number(); continue; }
        if (c.starts("aniso")) { c.p += 6; cur.anisotropy = c.number(); continue; }
        if (c.starts("anisor")) { c.p += 7; cur.anisotropy_rotation = c.number(); continue; }
        struct { const char* kw; std::string material_t::*field; } maps[] = {
            {"map_Ka", &material_t::ambient_texname},   {"map_Kd", &material_t::diffuse_texname},
            {"map_Ks", &material_t::specular_texname},  {"map_Ns", &material_t::specular_highlight_texname},
            {"map_bump", &material_t::bump_texname},    {"bump", &material_t::bump_texname},
            {"map_d", &material_t::alpha_texname},      {"disp", &material_t::displacement_texname},
            {"map_Ke", &material_t::emissive_texname},  {"norm", &material_t::normal_texname},
        };
        for (auto& m : maps) {
            if (c.starts(m.kw)) {
                Cursor t{c.p + strlen(m.kw) + 1};
                cur.*(m.field) = texture_name(t);
                break;
            }
        }
        // anything else is ignored
    }
    // the material being assembled is always emitted at end of file, even when it was never named
    by_name.insert(std::make_pair(cur.name, static_cast<int>(out.size())));
    out.push_back(cur);
}

// Appends the faces gathered since the last flush to `shape`. Returns false when there were none.
bool flush_faces(shape_t& shape, const std::vector<std::vector<Corner>>& faces, int material_id, const std::string& name,
                 bool triangulate) {
    if (faces.empty()) return false;
    auto push = [&shape](const Corner& k) {
        index_t i;
        i.vertex_index = k.v;
        i.normal_index = k.vn;
        i.texcoord_index = k.vt;
        shape.mesh.indices.push_back(i);
    };
    for (const std::vector<Corner>& f : faces) {
        const size_t n = f.size();
        if (triangulate) {
            for (size_t k = 2; k < n; ++k) {  // fan around corner 0
                push(f[0]);
                push(f[k - 1]);
                push(f[k]);
                shape.mesh.num_face_vertices.push_back(3);
                shape.mesh.material_ids.push_back(material_id);
            }
        } else {
            for (size_t k = 0; k < n; ++k) push(f[k]);
            shape.mesh.num_face_vertices.push_back(static_cast<unsigned char>(n));
            shape.mesh.material_ids.push_back(material_id);
        }
    }
    shape.name = name;
    return true;
}

}  // namespace

float ParseFloatText(const char* s, const char* end, double fallback) {
    // Accumulates digits in a double exactly in this order, so that the float obtained equals the reference
    // reader's for every input (a library strtod can differ in the last place of the double, and then rarely
    // in the float): integer digits m = m*10 + d; fraction digits m += d * 10^-k (table for k < 8, pow above);
    // decimal exponent applied as ldexp(m * 5^e, e).
    if (s >= end) return static_cast<float>(fallback);
    const char* p = s;
    double sign = 1.0;
    if (*p == '+' || *p == '-') {
        if (*p == '-') sign = -1.0;
        ++p;
    } else if (!is_digit(*p)) {
        return static_cast<float>(fallback);
    }
    double m = 0.0;
    int ndig = 0;
    while (p != end && is_digit(*p)) { m = m * 10 + (*p - '0'); ++p; ++ndig; }
    if (ndig == 0) return static_cast<float>(fallback);
    int exponent = 0;
    bool done = (p == end);
    if (!done) {
        if (*p == '.') {
            static const double tenth[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
            ++p;
            int k = 1;
            while (p != end && is_digit(*p)) {
                m += (*p - '0') * (k < 8 ? tenth[k] : std::pow(10.0, -k));
                ++k;
                ++p;
            }
        } else if (*p != 'e' && *p != 'E') {
            done = true;
        }
    }
    if (!done && p != end && (*p == 'e' || *p == 'E')) {
        ++p;
        int esign = 1;
        if (p != end && (*p == '+' || *p == '-')) {
            if (*p == '-') esign = -1;
            ++p;
        } else if (!is_digit(*p)) {
            return static_cast<float>(fallback);
        }
        int nexp = 0;
        // (an exponent beyond a few hundred already gives infinity or zero: further digits are read but no longer accumulated,
        // so that an endless run of them cannot overflow the int)
        while (p != end && is_digit(*p)) { if (exponent < 100000) exponent = exponent * 10 + (*p - '0'); ++p; ++nexp; }
        exponent *= esign;
        if (nexp == 0) return static_cast<float>(fallback);
    }
    const double value = sign * (exponent ? std::ldexp(m * std::pow(5.0, exponent), exponent) : m);
    return static_cast<float>(value);
}

bool LoadObj(attrib_t* attrib, std::vector<shape_t>* shapes, std::vector<material_t>* materials, std::string* err,
             const char* filename, const char* mtl_basedir, bool triangulate) {
    attrib->vertices.clear();
    attrib->normals.clear();
    attrib->texcoords.clear();
    shapes->clear();

    std::ifstream in(filename);
    if (!in) {
        if (err) *err = std::string("Cannot open file [") + filename + "]\n";
        return false;
    }
    const std::string basedir = mtl_basedir ? mtl_basedir : "";

    std::vector<float> v, vn, vt;
    std::vector<std::vector<Corner>> faces;
    std::map<std::string, int> material_by_name;
    int material = -1;
    std::string name;
    shape_t shape;

    std::string line;
    while (read_line(in, line)) {
        if (line.empty()) continue;
        Cursor c{line.c_str()};
        c.skip_blanks();
        if (*c.p == '\0' || *c.p == '#') continue;

        if (c.p[0] == 'v' && is_blank(c.p[1])) {
            c.p += 2;
            float xyz[3];
            c.numbers(xyz, 3);
            v.insert(v.end(), xyz, xyz + 3);
            continue;
        }
        if (c.p[0] == 'v' && c.p[1] == 'n' && is_blank(c.p[2])) {
            c.p += 3;
            float xyz[3];
            c.numbers(xyz, 3);
            vn.insert(vn.end(), xyz, xyz + 3);
            continue;
        }
        if (c.p[0] == 'v' && c.p[1] == 't' && is_blank(c.p[2])) {
            c.p += 3;
            float uv[2];
            c.numbers(uv, 2);
            vt.insert(vt.end(), uv, uv + 2);
            continue;
        }
        if (c.p[0] == 'f' && is_blank(c.p[1])) {
            c.p += 2;
            c.skip_blanks();
            std::vector<Corner> face;
            while (!at_line_end(*c.p)) {
                face.push_back(parse_corner(c, static_cast<int>(v.size() / 3), static_cast<int>(vn.size() / 3),
                                            static_cast<int>(vt.size() / 2)));
                c.p += strspn(c.p, " \t\r");
            }
            faces.push_back(std::move(face));
            continue;
        }
        if (c.starts("usemtl")) {
            c.p += 7;
            const std::string mname = c.word();
            const auto it = material_by_name.find(mname);
            const int id = it == material_by_name.end() ? -1 : it->second;
            if (id != material) {
                // faces so far keep the previous material; the shape itself stays open
                flush_faces(shape, faces, material, name, triangulate);
                faces.clear();
                material = id;
            }
            continue;
        }
        if (c.starts("mtllib")) {
            c.p += 7;
            std::vector<std::string> names;
            {
                std::stringstream ss(c.p);
                std::string item;
                while (std::getline(ss, item, ' ')) names.push_back(item);
            }
            bool found = false;
            for (const std::string& n : names) {
                const std::string path = basedir + n;
                std::ifstream min(path.c_str());
                if (!min) {
                    if (err) *err += "WARN: Material file [ " + path + " ] not found.\n";
                    continue;
                }
                std::string warn;
                load_mtl(min, material_by_name, *materials, warn);
                if (err) *err += warn;
                found = true;
                break;
            }
            if (!found && err) {
                *err += names.empty() ? "WARN: Looks like empty filename for mtllib. Use default material. \n"
                                      : "WARN: Failed to load material file(s). Use default material.\n";
            }
            continue;
        }
        const bool is_group = c.p[0] == 'g' && is_blank(c.p[1]);
        const bool is_object = c.p[0] == 'o' && is_blank(c.p[1]);
        if (is_group || is_object) {
            // A new group/object closes the open shape. The shape is only kept when faces were pending at this
            // point: triangles already moved into it by an earlier `usemtl` switch are dropped with it when
            // nothing followed that switch (the reference reader does the same).
            if (flush_faces(shape, faces, material, name, triangulate)) shapes->push_back(shape);
            shape = shape_t();
            faces.clear();
            if (is_group) {
                std::vector<std::string> words;
                while (!at_line_end(*c.p)) {
                    words.push_back(c.word());
                    c.p += strspn(c.p, " \t\r");
                }
                name = words.size() > 1 ? words[1] : std::string();  // words[0] is the keyword itself
            } else {
                c.p += 2;
                name = c.word();
            }
            continue;
        }
        // t (subdivision tags), s, l, p, ... are ignored
    }
    const bool flushed = flush_faces(shape, faces, material, name, triangulate);
    if (flushed || !shape.mesh.indices.empty()) shapes->push_back(shape);

    attrib->vertices.swap(v);
    attrib->normals.swap(vn);
    attrib->texcoords.swap(vt);
    return true;
}

}  // namespace ptobj
