// sq_host.cpp — host side above the render boundary: .obj/.sq/camera loaders and the BIH
// build + flatten, with the reference's arithmetic (include/squigly_host.h).
//
// Array-based and O(n log n): the reference's list code (`!!` indexing, src/Obj.hs:83-85) is
// O(n^2) and cannot load the 1M-triangle configuration.  Tree shape, split planes and leaf
// order are bit-identical to src/BIH.hs because they decide traversal tie-breaks.
// Citations are relative to the reference repository root.
#include "../../include/squigly_host.h"
#include "sq_error.h"
#include "sq_host_types.h"
#include "sq_math.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using sq::f3;

// ----------------------------------------------------------------------------------------------
// Text scanning.  The reference grammar is Parsec (src/Obj.hs:96-171); this scanner accepts and
// rejects the same byte strings.  Malformed numbers, which the reference turns into lazy `read`
// failures, are reported as errors up front.
// ----------------------------------------------------------------------------------------------
namespace {

struct Scanner {
    const char* s; size_t n; size_t i = 0;
    Scanner(const char* p, size_t len) : s(p), n(len) {}
    int peek() const { return i < n ? (unsigned char)s[i] : -1; }
    static bool ws(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }
    static bool digit(int c) { return c >= '0' && c <= '9'; }
    static bool alnum(int c) { return digit(c) || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
    void skip_ws() { while (ws(peek())) ++i; }
    // Parsec `string`: returns 0 on match, 1 on failure with nothing consumed, 2 on failure after consuming
    int literal(const char* lit) {
        for (size_t k = 0; lit[k]; ++k) {
            if (peek() != (unsigned char)lit[k]) return k == 0 ? 1 : 2;
            ++i;
        }
        return 0;
    }
    // fractional (src/Obj.hs:115-121): -?digit*(.digit*)? then `read`, which needs digits on both sides of '.'
    bool number(float& out) {
        std::string tok;
        if (peek() == '-') { tok.push_back('-'); ++i; }
        size_t a = 0, b = 0; bool dot = false;
        while (digit(peek())) { tok.push_back((char)peek()); ++i; ++a; }
        if (peek() == '.') { tok.push_back('.'); ++i; dot = true; }
        while (digit(peek())) { tok.push_back((char)peek()); ++i; ++b; }
        if (a == 0 || (dot && b == 0)) { sq_set_error("no parse for number '%s' at byte %zu", tok.c_str(), i); return false; }
        out = std::strtof(tok.c_str(), nullptr);   // correctly rounded, like read/fromRational
        return true;
    }
    bool vec3(float v[3]) {                          // src/Obj.hs:166-171
        for (int k = 0; k < 3; ++k) { if (!number(v[k])) return false; skip_ws(); }
        return true;
    }
    bool word(std::string& out) {                    // src/Obj.hs:129-130
        out.clear();
        while (peek() >= 0 && !ws(peek())) { out.push_back((char)peek()); ++i; }
        if (out.empty()) { sq_set_error("expected a word at byte %zu", i); return false; }
        skip_ws();
        return true;
    }
};

struct ObjObject { std::string mtl; size_t face_begin, face_end, vert_end; };   // vert_end: vertices read up to and including this object
struct ObjFile {
    std::string mtllib;
    std::vector<f3> verts;                 // all objects' vertices, concatenated (src/Obj.hs:76)
    std::vector<int32_t> faces;            // 3 one-based indices per face
    std::vector<ObjObject> objects;
};

bool scan_obj(const char* text, size_t len, ObjFile& f) {
    Scanner sc(text, len);
    if (sc.literal("mtllib")) { sq_set_error("obj: expected 'mtllib' at byte %zu", sc.i); return false; }   // src/Obj.hs:126-127
    sc.skip_ws();
    if (!sc.word(f.mtllib)) return false;
    while (sc.peek() == 'o') {                                                                              // many parseObj, src/Obj.hs:97-103
        ++sc.i; sc.skip_ws();
        size_t name_len = 0;
        while (Scanner::alnum(sc.peek()) || sc.peek() == '.' || sc.peek() == '_') { ++sc.i; ++name_len; }  // src/Obj.hs:105-107
        if (!name_len) { sq_set_error("obj: empty object name at byte %zu", sc.i); return false; }
        sc.skip_ws();
        while (sc.peek() == 'v') {                                                                          // src/Obj.hs:109-110
            ++sc.i; sc.skip_ws();
            float v[3];
            if (!sc.vec3(v)) return false;
            f.verts.push_back(sq::mk(v[0], v[2], v[1]));                                                    // swapYZ, src/Obj.hs:112-113
        }
        if (sc.literal("usemtl")) { sq_set_error("obj: expected 'usemtl' at byte %zu", sc.i); return false; } // src/Obj.hs:123-124
        sc.skip_ws();
        ObjObject ob;
        ob.vert_end = f.verts.size();
        if (!sc.word(ob.mtl)) return false;
        if (sc.peek() == 's') {                                                                             // optional parseS, src/Obj.hs:132-133
            size_t mark = sc.i;
            if (sc.literal("s on")) {
                sc.i = mark;
                if (sc.literal("s off")) { sq_set_error("obj: bad 's' line at byte %zu", sc.i); return false; }
            }
            sc.skip_ws();
        }
        ob.face_begin = f.faces.size() / 3;
        while (sc.peek() == 'f') {                                                                          // src/Obj.hs:135-144
            ++sc.i; sc.skip_ws();
            for (int k = 0; k < 3; ++k) {
                if (!Scanner::digit(sc.peek())) { sq_set_error("obj: expected a face index at byte %zu", sc.i); return false; }
                long long v = 0;
                while (Scanner::digit(sc.peek())) { v = v * 10 + (sc.peek() - '0'); if (v > 2000000000LL) v = 2000000000LL; ++sc.i; }
                sc.skip_ws();
                f.faces.push_back((int32_t)v);
            }
        }
        ob.face_end = f.faces.size() / 3;
        f.objects.push_back(ob);
    }
    return true;
}

bool scan_sq(const char* text, size_t len, std::vector<std::string>& names, std::vector<sq_material>& mats) {  // src/Obj.hs:146-161
    Scanner sc(text, len);
    for (;;) {
        int r = sc.literal("newmtl ");
        if (r == 1) break;
        if (r == 2) { sq_set_error("sq: expected 'newmtl ' at byte %zu", sc.i); return false; }
        std::string name; sq_material m;
        if (!sc.word(name)) return false;
        sc.skip_ws();
        if (sc.literal("reflective ")) { sq_set_error("sq: expected 'reflective ' at byte %zu", sc.i); return false; }
        if (!sc.number(m.reflective)) return false;
        sc.skip_ws();
        if (!sc.vec3(m.surf)) return false;
        sc.skip_ws();
        if (sc.literal("emissive ")) { sq_set_error("sq: expected 'emissive ' at byte %zu", sc.i); return false; }
        if (!sc.number(m.emissive)) return false;
        sc.skip_ws();
        if (!sc.vec3(m.emit)) return false;
        sc.skip_ws();
        names.push_back(name); mats.push_back(m);
    }
    return true;
}

// ---- Haskell's derived Show text, for --debug (src/Obj.hs:55-57) ----
// show :: Float -> String (Numeric.showFloat): the shortest digits that identify the value; positional notation for
// 0.1 <= |x| < 10^7, otherwise d.ddde<n>; always a digit on both sides of the point.
std::string show_float(float x) {
    if (x != x) return "NaN";
    if (x - x != 0.0f) return x > 0 ? "Infinity" : "-Infinity";
    std::string out;
    uint32_t bits; std::memcpy(&bits, &x, 4);
    if (bits >> 31) { out = "-"; x = -x; }
    if (x == 0.0f) return out + "0.0";
    char buf[48]; std::string digits; int e10 = 0;
    for (int p = 1; p <= 9; ++p) {                      // shortest correctly rounded decimal that reads back as x
        std::snprintf(buf, sizeof buf, "%.*e", p - 1, (double)x);
        if (std::strtof(buf, nullptr) == x || p == 9) {
            digits.clear();
            const char* c = buf;
            for (; *c && *c != 'e'; ++c) if (*c >= '0' && *c <= '9') digits.push_back(*c);
            e10 = std::atoi(c + 1) + 1;                 // x = 0.d1d2... * 10^e10 (floatToDigits' convention)
            break;
        }
    }
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    if (e10 < 0 || e10 > 7) {                           // exponent form (formatRealFloat FFGeneric: e < 0 || e > 7)
        out += digits.substr(0, 1) + "." + (digits.size() > 1 ? digits.substr(1) : std::string("0")) + "e" + std::to_string(e10 - 1);
        return out;
    }
    std::string ip = digits.substr(0, std::min<size_t>((size_t)e10, digits.size()));
    ip.append((size_t)e10 - ip.size(), '0');
    if (ip.empty()) ip = "0";
    std::string fp = digits.size() > (size_t)e10 ? digits.substr((size_t)e10) : std::string("0");
    return out + ip + "." + fp;
}
std::string show_v3(const float v[3]) {                 // V3 {_x = .., _y = .., _z = ..}  (src/V3.hs:5)
    return "V3 {_x = " + show_float(v[0]) + ", _y = " + show_float(v[1]) + ", _z = " + show_float(v[2]) + "}";
}
// show :: String -> String, i.e. showList over GHC.Show.showLitChar.  The reference reads its files with readFile, which
// decodes them (UTF-8 in any modern locale) to code points before the parser sees them: a name is shown per CODE POINT.
//   > '\DEL'  -> \ddd (decimal), followed by \& if a digit comes next      '\DEL' -> \DEL      '\\' -> \\     '"' -> \"
//   >= ' '    -> itself              \a \b \f \n \r \t \v by letter            \SO -> \SO, followed by \& if an 'H' comes next
//   other control characters by their ASCII names (\NUL \SOH ... \US)
// A byte sequence that is not valid UTF-8 makes the reference's readFile throw; here each such byte is shown as the code
// point of the same number (Latin-1), which no test can pin.
std::string show_string(const std::string& t) {
    static const char* const kAscii[32] = { "NUL", "SOH", "STX", "ETX", "EOT", "ENQ", "ACK", "a", "b", "t", "n", "v", "f", "r", "SO", "SI",
                                            "DLE", "DC1", "DC2", "DC3", "DC4", "NAK", "SYN", "ETB", "CAN", "EM", "SUB", "ESC", "FS", "GS", "RS", "US" };
    std::vector<uint32_t> cps;
    for (size_t i = 0; i < t.size();) {                  // UTF-8 -> code points (shortest form, no surrogates, <= U+10FFFF)
        const unsigned char c = (unsigned char)t[i];
        int n = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
        uint32_t cp = n == 1 ? c : n == 2 ? (c & 0x1fu) : n == 3 ? (c & 0x0fu) : (c & 0x07u);
        bool ok = n > 0 && i + (size_t)n <= t.size();
        for (int k = 1; ok && k < n; ++k) { const unsigned char d = (unsigned char)t[i + (size_t)k]; ok = (d >> 6) == 2; cp = (cp << 6) | (d & 0x3fu); }
        static const uint32_t kMin[5] = { 0, 0, 0x80, 0x800, 0x10000 };
        if (ok && (cp < kMin[n] || cp > 0x10ffffu || (cp >= 0xd800u && cp <= 0xdfffu))) ok = false;
        if (!ok) { cps.push_back(c); ++i; } else { cps.push_back(cp); i += (size_t)n; }
    }
    std::string o = "\"";
    for (size_t i = 0; i < cps.size(); ++i) {
        const uint32_t c = cps[i];
        const uint32_t next = i + 1 < cps.size() ? cps[i + 1] : 0;
        if (c > 127) { o += "\\" + std::to_string(c); if (next >= '0' && next <= '9') o += "\\&"; }
        else if (c == 127) o += "\\DEL";
        else if (c == '"') o += "\\\"";
        else if (c == '\\') o += "\\\\";
        else if (c >= 32) o.push_back((char)c);
        else { o += std::string("\\") + kAscii[c]; if (c == 14 && next == 'H') o += "\\&"; }
    }
    return o + "\"";
}

bool read_file(const char* path, std::string& out) {
    FILE* fp = std::fopen(path, "rb");
    if (!fp) { sq_set_error("cannot open '%s'", path); return false; }
    char buf[1 << 16]; size_t got;
    out.clear();
    while ((got = std::fread(buf, 1, sizeof buf, fp)) > 0) out.append(buf, got);
    std::fclose(fp);
    return true;
}

}  // namespace

extern "C" int sq_mesh_from_text(const char* obj_text, size_t obj_len, const char* sq_text, size_t sq_len, sq_mesh** out) {
    if (!obj_text || !sq_text || !out) return sq_set_error("null argument");
    ObjFile f; std::vector<std::string> names; std::vector<sq_material> mats;
    if (!scan_obj(obj_text, obj_len, f)) return 1;
    if (!scan_sq(sq_text, sq_len, names, mats)) return 1;
    sq_mesh* m = new sq_mesh;
    m->mats = mats;
    if (!f.objects.empty()) {                           // print (head objs): Object {verts = [..], mtl = "..", faces = [..]}  (src/Obj.hs:88-94)
        const ObjObject& ob = f.objects[0];
        std::string t = "Object {verts = [";
        for (size_t i = 0; i < ob.vert_end; ++i) { const float v[3] = { f.verts[i].x, f.verts[i].y, f.verts[i].z }; t += (i ? "," : "") + show_v3(v); }
        t += "], mtl = " + show_string(ob.mtl) + ", faces = [";
        for (size_t fi = ob.face_begin; fi < ob.face_end; ++fi)
            t += std::string(fi > ob.face_begin ? "," : "") + "Face {_i1 = " + std::to_string(f.faces[3 * fi]) + ", _i2 = " + std::to_string(f.faces[3 * fi + 1]) +
                 ", _i3 = " + std::to_string(f.faces[3 * fi + 2]) + "}";
        m->show_first_object = t + "]}";
    }
    m->show_materials = "[";                            // print mats :: [(String, Material)]  (src/Color.hs:78-83)
    for (size_t i = 0; i < names.size(); ++i)
        m->show_materials += std::string(i ? "," : "") + "(" + show_string(names[i]) + ",Mat {reflective = " + show_float(mats[i].reflective) + ", surfColor = " +
                             show_v3(mats[i].surf) + ", emissive = " + show_float(mats[i].emissive) + ", emitColor = " + show_v3(mats[i].emit) + "})";
    m->show_materials += "]";
    // makeScene (src/Obj.hs:73-77): every (object, material) pair with equal names, objects outermost.
    for (const ObjObject& ob : f.objects)
        for (size_t mi = 0; mi < names.size(); ++mi) {
            if (ob.mtl != names[mi]) continue;
            for (size_t fi = ob.face_begin; fi < ob.face_end; ++fi) {                    // makeTris, src/Obj.hs:80-86
                sq_tri t;
                float* dst[3] = { t.v0, t.v1, t.v2 };
                for (int k = 0; k < 3; ++k) {
                    int32_t idx = f.faces[3 * fi + k];
                    if (idx < 1 || (size_t)idx > f.verts.size()) {
                        delete m;
                        return sq_set_error("obj: face index %d outside 1..%zu", idx, f.verts.size());
                    }
                    f3 v = f.verts[(size_t)idx - 1];
                    dst[k][0] = v.x; dst[k][1] = v.y; dst[k][2] = v.z;
                }
                t.mat = (int32_t)mi;
                m->tris.push_back(t);
            }
        }
    *out = m;
    return 0;
}

extern "C" int sq_mesh_from_obj(const char* obj_path, const char* mtl_dir, sq_mesh** out) {
    if (!obj_path || !mtl_dir || !out) return sq_set_error("null argument");
    std::string obj, sqt;
    if (!read_file(obj_path, obj)) return 1;
    ObjFile probe;
    {   // only the first line is needed to find the material file (src/Obj.hs:51-52)
        Scanner sc(obj.data(), obj.size());
        if (sc.literal("mtllib")) return sq_set_error("obj: expected 'mtllib' at byte %zu", sc.i);
        sc.skip_ws();
        if (!sc.word(probe.mtllib)) return 1;
    }
    std::string path = std::string(mtl_dir) + "/" + probe.mtllib;
    if (!read_file(path.c_str(), sqt)) return 1;
    return sq_mesh_from_text(obj.data(), obj.size(), sqt.data(), sqt.size(), out);
}

extern "C" int sq_mesh_from_arrays(const sq_tri* tris, int32_t n_tris, const sq_material* mats, int32_t n_mats, sq_mesh** out) {
    if ((!tris && n_tris) || (!mats && n_mats) || !out || n_tris < 0 || n_mats < 0) return sq_set_error("bad argument");
    for (int32_t i = 0; i < n_tris; ++i)
        if (tris[i].mat < 0 || tris[i].mat >= n_mats) return sq_set_error("triangle %d has material %d outside 0..%d", i, tris[i].mat, n_mats - 1);
    sq_mesh* m = new sq_mesh;
    m->tris.assign(tris, tris + n_tris);
    m->mats.assign(mats, mats + n_mats);
    *out = m;
    return 0;
}
extern "C" int32_t sq_mesh_num_tris(const sq_mesh* m) { return (int32_t)m->tris.size(); }
extern "C" int32_t sq_mesh_num_materials(const sq_mesh* m) { return (int32_t)m->mats.size(); }
extern "C" const sq_tri* sq_mesh_tris(const sq_mesh* m) { return m->tris.data(); }
extern "C" const sq_material* sq_mesh_materials(const sq_mesh* m) { return m->mats.data(); }
extern "C" void sq_mesh_free(sq_mesh* m) { delete m; }
extern "C" void sq_mesh_debug_show(const sq_mesh* m, const char** first_object, const char** materials) {
    if (first_object) *first_object = m ? m->show_first_object.c_str() : "";
    if (materials) *materials = m ? m->show_materials.c_str() : "";
}

// ----------------------------------------------------------------------------------------------
// Camera (src/Obj.hs:60-70, src/Geometry.hs:90-107)
// ----------------------------------------------------------------------------------------------
namespace {
// Data.Matrix product: every entry is a dot product folded from a zero accumulator, r <- a*b + r.
void mul3(const float a[9], const float b[9], float o[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float r = 0.0f;
            for (int k = 0; k < 3; ++k) r = a[3 * i + k] * b[3 * k + j] + r;
            o[3 * i + j] = r;
        }
}
}  // namespace

extern "C" void sq_rot_matrix_rads(float alp, float bet, float gam, float out9[9]) {
    float sa, ca, sb, cb, sg, cg;
    sq::fsincos(alp, sa, ca); sq::fsincos(bet, sb, cb); sq::fsincos(gam, sg, cg);
    const float rz[9] = { ca, -sa, 0, sa, ca, 0, 0, 0, 1 };
    const float ry[9] = { cb, 0, sb, 0, 1, 0, -sb, 0, cb };
    const float rx[9] = { 1, 0, 0, 0, cg, -sg, 0, sg, cg };
    float yx[9];
    mul3(ry, rx, yx);            // foldr1 (*): rz * (ry * rx)
    mul3(rz, yx, out9);
}
extern "C" int sq_camera_from_text(const char* text, size_t len, sq_camera* cam) {
    if (!text || !cam) return sq_set_error("null argument");
    Scanner sc(text, len);
    float p[3], e[3];
    if (!sc.vec3(p) || !sc.vec3(e)) return 1;
    std::memcpy(cam->pos, p, sizeof p);
    sq_rot_matrix_rads(e[0], e[1], e[2], cam->rot);
    return 0;
}
extern "C" int sq_camera_from_file(const char* path, sq_camera* cam) {
    std::string t;
    if (!path || !read_file(path, t)) return path ? 1 : sq_set_error("null argument");
    return sq_camera_from_text(t.data(), t.size(), cam);
}

// ----------------------------------------------------------------------------------------------
// BIH build (src/BIH.hs:62-99) straight into pre-order arrays.
// ----------------------------------------------------------------------------------------------
namespace {

struct Builder {
    const std::vector<sq_tri>& src;
    sq_bih& out;
    std::vector<int32_t> scratch;

    static const float* vert(const sq_tri& t, int k) { return k == 0 ? t.v0 : (k == 1 ? t.v1 : t.v2); }

    // boundingBox = getBounds . concatMap vertices (src/Geometry.hs:155-163,195-197): foldl1 min / max
    sq_bounds bounds_of(const int32_t* ids, size_t n) const {
        sq_bounds b; bool first = true;
        for (size_t i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) {
                const float* v = vert(src[(size_t)ids[i]], k);
                if (first) { for (int c = 0; c < 3; ++c) b.lo[c] = b.hi[c] = v[c]; first = false; }
                else for (int c = 0; c < 3; ++c) { b.lo[c] = sq::hmin(b.lo[c], v[c]); b.hi[c] = sq::hmax(b.hi[c], v[c]); }
            }
        if (first) for (int c = 0; c < 3; ++c) b.lo[c] = b.hi[c] = 0.0f;
        return b;
    }
    // longestAxis (src/Geometry.hs:190-193): maximumBy keeps the later of equal maxima
    static int longest_axis(const sq_bounds& b) {
        int best = 0;
        for (int c = 1; c < 3; ++c) {
            float cur = b.hi[best] - b.lo[best], cand = b.hi[c] - b.lo[c];
            if (!sq::cmp_gt(cur, cand)) best = c;
        }
        return best;
    }
    // averagePoints (vertices tri), one component (src/Geometry.hs:181-182)
    float centroid(const sq_tri& t, int ax) const { return (((0.0f + t.v0[ax]) + t.v1[ax]) + t.v2[ax]) / 3.0f; }

    int32_t emit_leaf(const int32_t* ids, size_t n, int depth) {
        sq_node nd; nd.kind = 3 | ((int32_t)n << 2); nd.lmax = 0; nd.rmin = 0; nd.link = (int32_t)out.tris.size();
        for (size_t i = 0; i < n; ++i) out.tris.push_back(src[(size_t)ids[i]]);
        out.nodes.push_back(nd);
        out.leaves++;
        if ((int32_t)n > out.longest) out.longest = (int32_t)n;
        if (depth > out.height) out.height = depth;
        return (int32_t)out.nodes.size() - 1;
    }

    // bih bbox geom (src/BIH.hs:67-80); ids[0..n) is reordered in place (stable partition)
    void build(const sq_bounds& bbox, int32_t* ids, size_t n, int depth) {
        if (n < 15) { emit_leaf(ids, n, depth); return; }
        // split (src/BIH.hs:82-99)
        const int ax = longest_axis(bbox);
        std::vector<float> cen(n);
        float sum = 0.0f, count = 0.0f;
        for (size_t i = 0; i < n; ++i) { cen[i] = centroid(src[(size_t)ids[i]], ax); sum = sum + cen[i]; count = count + 1.0f; }
        const float plane = sum / count;
        scratch.resize(n);
        size_t nl = 0, nr = 0;
        for (size_t i = 0; i < n; ++i) { if (cen[i] < plane) ids[nl++] = ids[i]; else scratch[nr++] = ids[i]; }
        for (size_t i = 0; i < nr; ++i) ids[nl + i] = scratch[i];
        float lm = bbox.lo[ax], rm = bbox.hi[ax];                     // maximumDef / minimumDef defaults
        for (size_t i = 0; i < nl; ++i) for (int k = 0; k < 3; ++k) {
            float c = vert(src[(size_t)ids[i]], k)[ax];
            lm = (i == 0 && k == 0) ? c : sq::hmax(lm, c);
        }
        for (size_t i = 0; i < nr; ++i) for (int k = 0; k < 3; ++k) {
            float c = vert(src[(size_t)ids[nl + i]], k)[ax];
            rm = (i == 0 && k == 0) ? c : sq::hmin(rm, c);
        }
        sq_node nd; nd.kind = ax; nd.lmax = 0.001f + lm; nd.rmin = (-0.001f) + rm; nd.link = -1;
        const size_t me = out.nodes.size();
        out.nodes.push_back(nd);
        if (nl == 0) {                                                // src/BIH.hs:70-72
            emit_leaf(ids, 0, depth + 1);
            out.nodes[me].link = emit_leaf(ids, nr, depth + 1);
        } else if (nr == 0) {                                         // src/BIH.hs:73-75
            emit_leaf(ids, nl, depth + 1);
            out.nodes[me].link = emit_leaf(ids, 0, depth + 1);
        } else {                                                      // src/BIH.hs:76-78
            sq_bounds lb = bounds_of(ids, nl), rb = bounds_of(ids + nl, nr);
            build(lb, ids, nl, depth + 1);
            out.nodes[me].link = (int32_t)out.nodes.size();
            build(rb, ids + nl, nr, depth + 1);
        }
    }
};

}  // namespace

extern "C" int sq_bih_build(const sq_mesh* mesh, sq_bih** outp) {
    if (!mesh || !outp) return sq_set_error("null argument");
    sq_bih* b = new sq_bih;
    b->mats = mesh->mats;
    const size_t n = mesh->tris.size();
    std::vector<int32_t> ids(n);
    for (size_t i = 0; i < n; ++i) ids[i] = (int32_t)i;
    Builder bl{ mesh->tris, *b, {} };
    b->tris.reserve(n);
    b->root = bl.bounds_of(ids.data(), n);                            // makeBIH, src/BIH.hs:62-65
    bl.build(b->root, ids.data(), n, 1);
    *outp = b;
    return 0;
}
extern "C" void sq_bih_scene(const sq_bih* b, sq_scene* out) {
    out->root = b->root;
    out->nodes = b->nodes.data(); out->n_nodes = (int32_t)b->nodes.size();
    out->tris = b->tris.data();   out->n_tris = (int32_t)b->tris.size();
    out->mats = b->mats.data();   out->n_mats = (int32_t)b->mats.size();
    out->height = b->height;
}
extern "C" int32_t sq_bih_height(const sq_bih* b) { return b->height; }
extern "C" int32_t sq_bih_num_leaves(const sq_bih* b) { return b->leaves; }
extern "C" int32_t sq_bih_longest_leaf(const sq_bih* b) { return b->longest; }
extern "C" void sq_bih_free(sq_bih* b) { delete b; }

// ---- Culling boxes: an exact reduction of the triangle tests (no counterpart in the reference) ----------------------
// A Leaf equation (src/BIH.hs:105-109) returns Nothing unless mollerTrumbore accepts one of its triangles.  The BIH only
// clips along split axes, so most leaves a ray visits are far from it: sq_cull_boxes() gives every node a box such that
//
//     a ray within the limits below whose fp32 mollerTrumbore (src/Geometry.hs:117-142, round-to-nearest, no FMA)
//     ACCEPTS some triangle of the node  ==>  the ray passes the fp32 slab test of the node's box,
//
// so a ray that fails the slab test can skip the node's triangles and the node still returns exactly what the reference
// returns (Nothing).  The box is the triangles' bounding box grown by a margin m that covers the rounding of the
// accepted test.  Derivation (u = 2^-24, eps = 0.0001f, s = o - v0, E1 = |e1|, E2 = |e2|, D >= |d|, S >= |s|, P = E1 E2 D;
// h = d x e2, a = e1.h, Nu = s.h, q = s x e1, Nv = d.q, Nt = e2.q; hats are the computed values):
//   |h^ - h|_2 <= 3.5u D E2 (two products and a difference per component), dot products are within 3.01u |x||y|, so
//   |a^ - a| <= 7u P,  |Nu^ - Nu| <= 8u S D E2,  |Nv^ - Nv| <= 8u S D E1,  |Nt^ - Nt| <= 8u S E1 E2.
//   Acceptance means |a^| >= eps, 0 <= u^ <= 1, 0 <= v^, u^ + v^ <= 1 (+u), t^ > eps.  With r = 7uP/eps <= 1/8 (P <= 29)
//   a and a^ have one sign and the exact solution (u, v, t) of  o + t d = v0 + u e1 + v e2  satisfies
//   |u - u^| <= 1.15 (r + 8u S D E2/eps) + 2.01u, the same for v with E1, and t > -1.15 * 8u S E1 E2 / eps.
//   So the point X = v0 + u e1 + v e2 lies on the ray's LINE at parameter t and within |u-u^| E1 + |v-v^| E2 of the
//   triangle; if t < 0 the origin is within |t| D of X.  Either way a point of the ray with t >= 0 lies within
//       rho = 28 (u/eps) P (S + E1 + E2) + 5u (E1 + E2)
//   of the triangle (the last term also covers e1 = fl(v1 - v0)).  The margin uses 32 and 6.
//   Slab test: with df = fl(1/d), nodf = fl(-o df) and plane value fma(l, df, nodf) (or fl(fl(l - o) df)) every plane
//   value is within 4u (|l| + |o|)/|d_k| of the exact one, so growing the box by a further 8u (|l| + |o|) makes every
//   computed interval contain the exact interval of the rho-box with room to spare: an exact hit of the rho-box passes
//   `tmax > 0 && tmin < tmax`.  Subnormal results add at most 2^-149 per operation, far below the floor added at the end.
// Limits under which this holds (rays outside them are simply not culled; leaves outside them get an infinite box):
//   every coordinate finite and <= 2^20 in magnitude (no overflow anywhere in the test), P <= 29 for every triangle of
//   the leaf, 0.25 <= |d|^2 <= 1.5624 (primary rays have |d| <= 1.2248, bounce rays |d| = 1 or the incoming length), |o|^2 <= o2max
//   = (2 max|vertex|)^2, and o, d, 1/d, o/d finite.
// Branch nodes: the box of a branch is the union (componentwise min / max) of its children's boxes, so each of its planes IS
//   a plane of some leaf box below it, slack included.  A ray that mollerTrumbore accepts for a triangle of leaf L hits L's
//   rho-box exactly; the union contains that box with L's slack or more on every side (a union plane sits further out by
//   some Delta >= 0, which adds Delta of slack against 4u Delta / |d_k| of additional plane-value error), so the computed
//   intervals of the union box contain the exact ones of L's rho-box and the ray passes the union's slab test as well.  Hence
//   "misses the box of a subtree" implies "misses the box of every leaf below", and the whole subtree returns Nothing.
// tests/test_cull.py searches for violations with adversarial grazing rays against exact (binary64 / rational) geometry: the
// leaf box and the box of every ancestor, in fp32 and in the binary16 encoding, plane values as single-rounding FMAs.
namespace {
inline float f_down(double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -INFINITY); return f; }
inline float f_up(double x) { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, INFINITY); return f; }
}
extern "C" int sq_cull_boxes(const sq_scene* sc, float* boxes, float ray_limits[3]) {
    if (!sc || !boxes || !ray_limits) return sq_set_error("null argument");
    const int32_t n = sc->n_nodes;
    const double u = 5.9604644775390625e-8, eps = (double)0.0001f, kD = 1.25, kPmax = 29.0;
    const float inf = INFINITY;
    auto disable = [&]() { for (int32_t i = 0; i < n; ++i) { float* b = boxes + 6 * (size_t)i; b[0] = b[1] = b[2] = -inf; b[3] = b[4] = b[5] = inf; }
                           ray_limits[0] = -1.0f; ray_limits[1] = 0.25f; ray_limits[2] = 1.5624f; return 0; };
    double vmax_inf = 0.0, vmax2 = 0.0;
    for (int32_t i = 0; i < sc->n_tris; ++i) {
        const float* vs[3] = { sc->tris[i].v0, sc->tris[i].v1, sc->tris[i].v2 };
        for (const float* v : vs) {
            double q = 0.0;
            for (int c = 0; c < 3; ++c) { if (!(v[c] - v[c] == 0.0f)) return disable(); vmax_inf = std::max(vmax_inf, std::fabs((double)v[c])); q += (double)v[c] * v[c]; }
            vmax2 = std::max(vmax2, q);
        }
    }
    if (sc->n_tris == 0 || vmax_inf > 1048576.0) return disable();
    const double omax = 2.0 * std::sqrt(vmax2) * (1.0 + 1e-12);
    ray_limits[0] = f_down(omax * omax * (1.0 - 1e-6)); ray_limits[1] = 0.25f; ray_limits[2] = 1.5624f;
    for (int32_t i = n - 1; i >= 0; --i) {                          // children come after their parent in pre-order
        const sq_node& nd = sc->nodes[i];
        float* b = boxes + 6 * (size_t)i;
        if ((nd.kind & 3) != 3) {
            const float* l = boxes + 6 * (size_t)(i + 1); const float* r = boxes + 6 * (size_t)nd.link;
            for (int c = 0; c < 3; ++c) { b[c] = std::min(l[c], r[c]); b[3 + c] = std::max(l[3 + c], r[3 + c]); }
            continue;
        }
        const int32_t first = nd.link, cnt = nd.kind >> 2;
        double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 }, m = 0.0; bool ok = cnt > 0;
        for (int32_t k = first; k < first + cnt && ok; ++k) {
            const sq_tri& t = sc->tris[k];
            double E1 = 0, E2 = 0, V0 = 0;
            for (int c = 0; c < 3; ++c) {
                const float e1 = t.v1[c] - t.v0[c], e2 = t.v2[c] - t.v0[c];        // the kernels' edges (src/Geometry.hs:130-131)
                E1 += (double)e1 * e1; E2 += (double)e2 * e2; V0 += (double)t.v0[c] * t.v0[c];
                for (const float* v : { t.v0, t.v1, t.v2 }) { lo[c] = std::min(lo[c], (double)v[c]); hi[c] = std::max(hi[c], (double)v[c]); }
            }
            E1 = std::sqrt(E1); E2 = std::sqrt(E2); V0 = std::sqrt(V0);
            const double P = E1 * E2 * kD;
            if (!(P <= kPmax)) { ok = false; break; }
            m = std::max(m, 32.0 * (u / eps) * P * (omax + V0 + E1 + E2) + 6.0 * u * (E1 + E2));
        }
        if (!ok) { b[0] = b[1] = b[2] = -inf; b[3] = b[4] = b[5] = inf; continue; }
        m += 8.0 * u * (vmax_inf + m + omax) + 1e-20;
        for (int c = 0; c < 3; ++c) { b[c] = f_down(lo[c] - m); b[3 + c] = f_up(hi[c] + m); }
    }
    return 0;
}

// The binary16 value nearest to x on the side asked for (up: >= x, else <= x), never a subnormal: +-2^-14 or zero instead,
// so that the device's handling of binary16 denormals cannot matter.  NaN gives the infinity of that side.
extern "C" uint32_t sq_half_outward(float x, int32_t up_) {
    const bool up = up_ != 0;
    if (!(x == x)) return up ? 0x7C00u : 0xFC00u;
    const bool neg = std::signbit(x); const double ax = std::fabs((double)x);
    const bool mag_up = up != neg;                           // round the magnitude away from zero?
    uint32_t hb;
    if (ax == 0.0) hb = 0;
    else if (ax > 65504.0) hb = (mag_up || std::isinf(ax)) ? 0x7C00u : 0x7BFFu;
    else if (ax < 6.103515625e-05) hb = mag_up ? 0x0400u : 0u;
    else {
        int e; const double mant = std::frexp(ax, &e) * 2.0; e -= 1;     // ax = mant * 2^e, mant in [1, 2)
        const double m10 = mant * 1024.0 - 1024.0; const double fl = std::floor(m10);
        uint32_t q = (uint32_t)fl + ((mag_up && fl != m10) ? 1u : 0u);
        if (q == 1024u) { q = 0; ++e; }
        hb = e > 15 ? 0x7C00u : (((uint32_t)(e + 15) << 10) | q);
    }
    return neg ? (hb | 0x8000u) : hb;
}
