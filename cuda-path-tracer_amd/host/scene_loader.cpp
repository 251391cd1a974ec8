// The scene front end of the host facade in C++: scene JSON (the reference's current schema and the
// older capitalised schema of Scenes/cornell.json, pbr_*.json, sphere.json), glTF 2.0 and Wavefront OBJ,
// filling a pth::Scene in the order the reference's loader fills its own, so that material, texture,
// image, BLAS and instance ids come out as scene.cpp:156-578 assigns them:
//   * texture 0 of each kind and material 0 are the defaults (scene.cpp:235-260);
//   * named materials follow in key order (nlohmann::json keeps objects in a std::map, scene.cpp:337-356),
//     each texture specification appending a new texture (ParseTexture, scene.h:76-179);
//   * a glTF object appends its own materials and images when it is loaded (scene.cpp:439-462) and
//     yields one instance per primitive per node in the loader's stack order (mesh/gltf_model.cpp:175-193,
//     scene.cpp:472-512); every object entry reloads its file;
//   * TYPE "instance": a NAME + INSTANCE entry defines a source that is not placed itself, a SOURCE
//     entry places a copy with its own TRS in front (scene.cpp:526-577).
// No third-party parser is linked: the JSON reader, the base64 decoder, the glTF accessor logic and
// the OBJ reader below are this project's.  Image files are the exception - this image offers no
// decoder to link (no libjpeg / libpng headers; the reference's stb copy is not ours to take) - so the
// caller passes a decode callback; scenes without image textures need none.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ptamd_host.h"
#include "pt_host.h"

namespace pth {
namespace {

// ------------------------------------------------------------------------------------ JSON
struct Json {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;  // file order

    const Json *find(const std::string &key) const {
        if (type != Object) return nullptr;
        for (const auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const std::string &key) const { return find(key) != nullptr; }
    bool is_number() const { return type == Number; }
    bool is_string() const { return type == String; }
    bool is_array() const { return type == Array; }
    bool is_object() const { return type == Object; }
};

class JsonParser {
public:
    explicit JsonParser(const std::string &text) : s(text) {}
    Json Parse() {
        Json v = Value();
        Skip();
        if (p != s.size()) Error("trailing characters");
        return v;
    }

private:
    const std::string &s;
    size_t p = 0;
    int depth = 0;  // nesting of the value being parsed: bounded, so that a hostile file cannot exhaust the stack
    [[noreturn]] void Error(const char *what) const { throw Exception("JSON: " + std::string(what) + " at byte " + std::to_string(p)); }
    void Skip() {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) ++p;
    }
    struct Nest {
        int &d;
        explicit Nest(int &depth) : d(depth) { ++d; }
        ~Nest() { --d; }
    };
    Json Value() {
        Skip();
        if (p >= s.size()) Error("unexpected end");
        const Nest nest(depth);
        if (depth > 200) Error("nesting deeper than 200 levels");
        const char c = s[p];
        Json v;
        if (c == '{') {
            v.type = Json::Object;
            ++p;
            Skip();
            if (p < s.size() && s[p] == '}') { ++p; return v; }
            for (;;) {
                Skip();
                if (p >= s.size() || s[p] != '"') Error("object key expected");
                std::string key = StringBody();
                Skip();
                if (p >= s.size() || s[p] != ':') Error("':' expected");
                ++p;
                Json child = Value();
                bool replaced = false;  // a repeated key keeps its last value, like the reference's parser
                for (auto &kv : v.obj)
                    if (kv.first == key) { kv.second = child; replaced = true; }
                if (!replaced) v.obj.emplace_back(std::move(key), std::move(child));
                Skip();
                if (p < s.size() && s[p] == ',') { ++p; continue; }
                if (p < s.size() && s[p] == '}') { ++p; break; }
                Error("',' or '}' expected");
            }
        } else if (c == '[') {
            v.type = Json::Array;
            ++p;
            Skip();
            if (p < s.size() && s[p] == ']') { ++p; return v; }
            for (;;) {
                v.arr.push_back(Value());
                Skip();
                if (p < s.size() && s[p] == ',') { ++p; continue; }
                if (p < s.size() && s[p] == ']') { ++p; break; }
                Error("',' or ']' expected");
            }
        } else if (c == '"') {
            v.type = Json::String;
            v.str = StringBody();
        } else if (s.compare(p, 4, "true") == 0) {
            v.type = Json::Bool; v.b = true; p += 4;
        } else if (s.compare(p, 5, "false") == 0) {
            v.type = Json::Bool; v.b = false; p += 5;
        } else if (s.compare(p, 4, "null") == 0) {
            p += 4;
        } else {
            const char *start = s.c_str() + p;
            char *end = nullptr;
            v.num = std::strtod(start, &end);
            if (end == start) Error("value expected");
            v.type = Json::Number;
            p += size_t(end - start);
        }
        return v;
    }
    std::string StringBody() {
        std::string out;
        ++p;  // opening quote
        while (p < s.size() && s[p] != '"') {
            char c = s[p++];
            if (c == '\\') {
                if (p >= s.size()) Error("bad escape");
                const char e = s[p++];
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (p + 4 > s.size()) Error("bad \\u escape");
                        const unsigned cp = unsigned(std::strtoul(s.substr(p, 4).c_str(), nullptr, 16));
                        p += 4;
                        if (cp < 0x80) out += char(cp);
                        else if (cp < 0x800) { out += char(0xC0 | (cp >> 6)); out += char(0x80 | (cp & 0x3F)); }
                        else { out += char(0xE0 | (cp >> 12)); out += char(0x80 | ((cp >> 6) & 0x3F)); out += char(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += e;  // \" \\ \/
                }
            } else {
                out += c;
            }
        }
        if (p >= s.size()) Error("unterminated string");
        ++p;
        return out;
    }
};

// A JSON number used as a count, an index or an integer option.  Converting a double that is not finite, or
// that lies outside the target type's range, to an integer is undefined behaviour: such values end in an error.
long long Whole(double v, const char *what) {
    if (!(v >= -9.0e15 && v <= 9.0e15)) throw Exception(std::string("'") + what + "' is not a usable whole number");  // NaN fails both comparisons
    return static_cast<long long>(v);
}
int WholeInt(double v, const char *what) {
    const long long w = Whole(v, what);
    if (w < -2147483647LL - 1 || w > 2147483647LL) throw Exception(std::string("'") + what + "' is outside the range of a 32-bit integer");
    return static_cast<int>(w);
}
size_t WholeSize(double v, const char *what) {
    const long long w = Whole(v, what);
    if (w < 0) throw Exception(std::string("'") + what + "' is negative");
    return static_cast<size_t>(w);
}

std::string ReadFile(const std::string &path, bool binary = true) {
    std::ifstream f(path, binary ? std::ios::binary : std::ios::in);
    if (!f) throw Exception("could not open file '" + path + "'");
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
bool FileExists(const std::string &path) { return bool(std::ifstream(path)); }
std::string DirName(const std::string &path) {
    const size_t k = path.find_last_of('/');
    return k == std::string::npos ? std::string(".") : path.substr(0, k);
}
std::string Join(const std::string &dir, const std::string &rel) { return (!rel.empty() && rel[0] == '/') ? rel : dir + "/" + rel; }
std::string Lower(std::string s) {
    for (char &c : s) c = char(std::tolower((unsigned char)c));
    return s;
}

std::string Base64Decode(const std::string &in, size_t from) {
    struct Table {
        int v[256];
        Table() {
            for (int &t : v) t = -1;
            const char *abc = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
            for (int i = 0; i < 64; ++i) v[(unsigned char)abc[i]] = i;
        }
    };
    static const Table t;  // initialised once, thread-safe
    const int *table = t.v;
    std::string out;
    out.reserve((in.size() - from) * 3 / 4);
    unsigned acc = 0;
    int bits = 0;
    for (size_t i = from; i < in.size(); ++i) {
        const int v = table[(unsigned char)in[i]];
        if (v < 0) continue;  // padding, line breaks
        acc = (acc << 6) | unsigned(v);
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out += char((acc >> bits) & 0xFF);
        }
    }
    return out;
}

// ------------------------------------------------------------------------------ geometry
struct MeshData {
    std::vector<float> p, n, t;  // xyz, xyz, uv per vertex
    std::vector<uint32_t> i;
    uint32_t vertex_count() const { return uint32_t(p.size() / 3); }
};

Vec3 At(const std::vector<float> &v, size_t k) { return {v[3 * k], v[3 * k + 1], v[3 * k + 2]}; }
void Push(std::vector<float> &v, Vec3 a) { v.push_back(a.x); v.push_back(a.y); v.push_back(a.z); }

// LoadWavefrontObjMesh, mesh/wavefront_obj.cpp:7-107: unindexed triangles; normals are recomputed per
// face when the file has none or FACE_NORMALS is set.  Polygons are fanned into triangles, negative
// indices count from the end, as the OBJ format defines them.
MeshData LoadObj(const std::string &path, bool face_normals, bool invert_normals) {
    std::istringstream file(ReadFile(path, false));
    std::vector<float> v, vn, vt;
    struct Corner { long vi, ti, ni; };
    std::vector<Corner> corners;  // three per triangle
    std::string line;
    while (std::getline(file, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag)) continue;
        auto numbers = [&](int count, std::vector<float> &dst) {
            for (int k = 0; k < count; ++k) {
                std::string tok;
                ls >> tok;
                dst.push_back(float(std::strtod(tok.c_str(), nullptr)));
            }
        };
        if (tag == "v") numbers(3, v);
        else if (tag == "vn") numbers(3, vn);
        else if (tag == "vt") numbers(2, vt);
        else if (tag == "f") {
            std::vector<Corner> face;
            std::string tok;
            while (ls >> tok) {
                Corner c{0, 0, 0};
                const size_t a = tok.find('/');
                c.vi = std::strtol(tok.substr(0, a).c_str(), nullptr, 10);
                if (a != std::string::npos) {
                    const size_t b = tok.find('/', a + 1);
                    const std::string ts = tok.substr(a + 1, b == std::string::npos ? std::string::npos : b - a - 1);
                    if (!ts.empty()) c.ti = std::strtol(ts.c_str(), nullptr, 10);
                    if (b != std::string::npos && b + 1 < tok.size()) c.ni = std::strtol(tok.substr(b + 1).c_str(), nullptr, 10);
                }
                face.push_back(c);
            }
            for (size_t k = 1; k + 1 < face.size(); ++k) {
                corners.push_back(face[0]);
                corners.push_back(face[k]);
                corners.push_back(face[k + 1]);
            }
        }
    }
    auto fix = [](long i, size_t n) { return size_t(i > 0 ? i - 1 : long(n) + i); };
    const bool compute = vn.empty() || face_normals;
    MeshData m;
    const size_t nv = v.size() / 3, nn = vn.size() / 3, nt = vt.size() / 2;
    for (size_t f = 0; f < corners.size(); f += 3) {
        Vec3 pos[3], nrm[3];
        float uv[3][2] = {{0, 0}, {0, 0}, {0, 0}};
        for (int c = 0; c < 3; ++c) {
            const Corner &k = corners[f + c];
            const size_t vi = fix(k.vi, nv);
            if (vi >= nv) throw Exception("OBJ '" + path + "': vertex index out of range");
            pos[c] = At(v, vi);
            if (k.ti) {
                const size_t ti = fix(k.ti, nt);
                if (ti >= nt) throw Exception("OBJ '" + path + "': texture index out of range");
                uv[c][0] = vt[2 * ti];
                uv[c][1] = vt[2 * ti + 1];
            }
            nrm[c] = Vec3{0, 0, 0};
            if (!compute && k.ni) {
                const size_t ni = fix(k.ni, nn);
                if (ni >= nn) throw Exception("OBJ '" + path + "': normal index out of range");
                nrm[c] = At(vn, ni);
            }
        }
        Vec3 face_normal{0, 0, 0};
        if (compute) face_normal = Normalize(Cross(Normalize(pos[1] - pos[0]), Normalize(pos[2] - pos[0])));
        for (int c = 0; c < 3; ++c) {
            Vec3 normal = compute ? face_normal : Normalize(nrm[c]);
            if (invert_normals) normal = normal * -1.0f;
            Push(m.p, pos[c]);
            Push(m.n, normal);
            m.t.push_back(uv[c][0]);
            m.t.push_back(uv[c][1]);
            m.i.push_back(uint32_t(m.i.size()));
        }
    }
    return m;
}

// The older scene files name analytic primitives (TYPE cube / sphere), which the reference's parser at
// this commit does not handle.  This project's definition: cube = unit cube [-0.5, 0.5]^3 with per-face
// normals and [0,1]^2 uvs per face, sphere = radius 0.5, 32 stacks x 64 slices, smooth normals.
MeshData UnitCube() {
    static const float faces[6][3][3] = {{{0, 0, 1}, {1, 0, 0}, {0, 1, 0}},  {{0, 0, -1}, {-1, 0, 0}, {0, 1, 0}}, {{1, 0, 0}, {0, 0, -1}, {0, 1, 0}},
                                         {{-1, 0, 0}, {0, 0, 1}, {0, 1, 0}}, {{0, 1, 0}, {1, 0, 0}, {0, 0, -1}},  {{0, -1, 0}, {1, 0, 0}, {0, 0, 1}}};
    static const int signs[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
    MeshData m;
    for (const auto &f : faces) {
        const uint32_t base = m.vertex_count();
        for (const auto &s : signs) {
            for (int a = 0; a < 3; ++a) m.p.push_back(0.5f * (f[0][a] + float(s[0]) * f[1][a] + float(s[1]) * f[2][a]));
            for (int a = 0; a < 3; ++a) m.n.push_back(f[0][a]);
            m.t.push_back(float((s[0] + 1) / 2.0));
            m.t.push_back(float((s[1] + 1) / 2.0));
        }
        for (uint32_t k : {0u, 1u, 2u, 0u, 2u, 3u}) m.i.push_back(base + k);
    }
    return m;
}
MeshData UvSphere(int stacks = 32, int slices = 64, double radius = 0.5) {
    const double pi = 3.141592653589793;
    MeshData m;
    for (int i = 0; i <= stacks; ++i) {
        const double theta = pi * i / stacks;
        for (int k = 0; k <= slices; ++k) {
            const double phi = 2.0 * pi * k / slices;
            const double n[3] = {std::sin(theta) * std::cos(phi), std::cos(theta), std::sin(theta) * std::sin(phi)};
            for (int a = 0; a < 3; ++a) m.p.push_back(float(radius * n[a]));
            Push(m.n, Normalize(Vec3{float(n[0]), float(n[1]), float(n[2])}));
            m.t.push_back(float(double(k) / slices));
            m.t.push_back(float(double(i) / stacks));
        }
    }
    const uint32_t row = uint32_t(slices) + 1;
    for (int i = 0; i < stacks; ++i)
        for (int k = 0; k < slices; ++k) {
            const uint32_t a = uint32_t(i) * row + uint32_t(k), b = a + 1, c = a + row, d = c + 1;
            if (i != 0) for (uint32_t x : {a, b, c}) m.i.push_back(x);
            if (i != stacks - 1) for (uint32_t x : {b, d, c}) m.i.push_back(x);
        }
    return m;
}

// ---------------------------------------------------------------------------------- loader
class Loader {
public:
    Loader(Scene &scene, pth_image_decoder decode, void *user) : S(scene), m_decode(decode), m_user(user) {}

    void LoadSceneFile(const std::string &path, int width, int height, int max_depth) {
        const Json data = JsonParser(ReadFile(path)).Parse();
        m_dir = DirName(path);
        m_legacy = !data.has("camera") && data.has("Camera");
        int depth = 4, iterations = 1000;
        const Json *cam = nullptr;
        int file_w = 800, file_h = 800;
        if (m_legacy) {
            cam = data.find("Camera");
            depth = WholeInt(NumberOr(*cam, "DEPTH", 4), "DEPTH");
            iterations = WholeInt(NumberOr(*cam, "ITERATIONS", 1000), "ITERATIONS");
            if (const Json *res = cam->find("RES"); res && res->is_array() && res->arr.size() == 2) {
                file_w = WholeInt(res->arr[0].num, "RES");
                file_h = WholeInt(res->arr[1].num, "RES");
            }
        } else {
            if (const Json *in = data.find("integrator")) {  // ParseIntegrator, scene.cpp:156-173
                depth = WholeInt(NumberOr(*in, "DEPTH", 4), "DEPTH");
                iterations = WholeInt(NumberOr(*in, "ITERATIONS", 1000), "ITERATIONS");
            }
            cam = data.find("camera");
            if (!cam) throw Exception("scene '" + path + "' has no camera");
            const Json *film = cam->find("film");
            if (!film || !film->has("WIDTH") || !film->has("HEIGHT")) throw Exception("scene '" + path + "': camera.film needs WIDTH and HEIGHT");
            file_w = WholeInt(film->find("WIDTH")->num, "WIDTH");
            file_h = WholeInt(film->find("HEIGHT")->num, "HEIGHT");
        }
        S.SetIntegrator(max_depth > 0 ? max_depth : depth, iterations);
        Camera &c = S.GetCamera();  // ParseCamera, scene.cpp:175-233
        c.GetFilm().width = width > 0 ? width : file_w;
        c.GetFilm().height = height > 0 ? height : file_h;
        if (c.GetFilm().width < 1 || c.GetFilm().height < 1 || c.GetFilm().width > 65536 || c.GetFilm().height > 65536)
            throw Exception("scene '" + path + "': film size must lie in 1..65536 each way");
        c.SetDefaults(Vec3From(*cam, "EYE", {0, 0, 0}), Vec3From(*cam, "LOOKAT", {0, 0, -1}), Vec3From(*cam, "UP", {0, 1, 0}), float(NumberOr(*cam, "FOVY", 45.0)));
        c.ResetDefaults();

        // ParseMaterials, scene.cpp:337-356: in key order
        std::map<std::string, const Json *> materials;
        if (const Json *m = data.find(m_legacy ? "Materials" : "materials"); m && m->is_object())
            for (const auto &kv : m->obj) materials[kv.first] = &kv.second;
        for (const auto &kv : materials) {
            m_material_ids[kv.first] = m_next_material;
            ParseMaterial(*kv.second);
        }
        const Json *objects = data.find(m_legacy ? "Objects" : "objects");
        if (!objects || !objects->is_array()) return;
        // pass 1, scene.cpp:526-547
        for (const Json &obj : objects->arr) {
            const std::string type = Lower(StringOr(obj, "TYPE", ""));
            for (const Placed &pl : ParseObject(obj)) Place(pl);
            if (type == "instance" && obj.has("NAME") && obj.has("INSTANCE")) m_named[obj.find("NAME")->str] = ParseObject(*obj.find("INSTANCE"));
        }
        // pass 2, scene.cpp:549-577
        for (const Json &obj : objects->arr) {
            if (Lower(StringOr(obj, "TYPE", "")) != "instance" || !obj.has("SOURCE")) continue;
            const Mat4 xf = ObjectTransform(obj);
            const auto it = m_named.find(obj.find("SOURCE")->str);
            if (it == m_named.end()) continue;
            for (Placed pl : it->second) {
                pl.transform = Mul(xf, pl.transform);
                Place(pl);
            }
        }
    }

private:
    struct Placed {
        TriangleMesh mesh;
        int material = 0;
        Mat4 transform;
        int blas = 0;
    };
    Scene &S;
    pth_image_decoder m_decode;
    void *m_user;
    std::string m_dir;
    bool m_legacy = false;
    int m_next_material = 1;  // material 0 is the default
    std::map<std::string, int> m_material_ids;
    std::map<std::string, std::vector<Placed>> m_named;

    void Place(const Placed &p) { S.AddInstance(p.mesh, p.material, p.transform, p.blas); }

    static double NumberOr(const Json &o, const char *key, double fallback) {
        const Json *v = o.find(key);
        return v && v->is_number() ? v->num : fallback;
    }
    static std::string StringOr(const Json &o, const char *key, const char *fallback) {
        const Json *v = o.find(key);
        return v && v->is_string() ? v->str : std::string(fallback);
    }
    static bool BoolOr(const Json &o, const char *key, bool fallback) {
        const Json *v = o.find(key);
        if (!v) return fallback;
        if (v->type == Json::Bool) return v->b;
        if (v->type == Json::Number) return v->num != 0.0;
        return fallback;
    }
    static Vec3 Vec3From(const Json &o, const char *key, Vec3 fallback) {  // Utils::GetOrDefault, utils/utils.h:84-96
        const Json *v = o.find(key);
        if (!v || !v->is_array() || v->arr.size() != 3) return fallback;
        return {float(v->arr[0].num), float(v->arr[1].num), float(v->arr[2].num)};
    }
    static Mat4 ObjectTransform(const Json &obj) {
        return GetTransformMatrix(Vec3From(obj, "TRANS", {0, 0, 0}), Vec3From(obj, "ROTAT", {0, 0, 0}), Vec3From(obj, "SCALE", {1, 1, 1}));
    }

    // ---- images.  `want_channels`: 0 as stored, 3 RGB, 4 RGBA.  `channel` < 0: the first three channels.
    struct Pixels {
        std::vector<uint8_t> px;
        int w = 0, h = 0, c = 0;
    };
    Pixels Decode(const std::string &bytes, int want_channels) const {
        if (!m_decode) throw Exception("the scene names an image file and no image decoder was given to the loader");
        Pixels out;
        if (m_decode(m_user, reinterpret_cast<const uint8_t *>(bytes.data()), bytes.size(), want_channels, &out.w, &out.h, &out.c, nullptr, 0) != 0 ||
            out.w <= 0 || out.h <= 0 || out.c <= 0)
            throw Exception("the image decoder rejected a file");
        out.px.resize(size_t(out.w) * out.h * out.c);
        if (m_decode(m_user, reinterpret_cast<const uint8_t *>(bytes.data()), bytes.size(), want_channels, &out.w, &out.h, &out.c, out.px.data(), out.px.size()) != 0)
            throw Exception("the image decoder failed");
        return out;
    }
    // Adds the image as fp32 texels: `div255` v / 255.0f (mesh/gltf_model.h:96-152), else v * (1.0f / 255.0f) (image.cpp:44-49, :88-97).
    int AddImage(const Pixels &im, int channel, bool div255, bool srgb, int *w, int *h) {
        const int out_c = channel < 0 ? 3 : 1;
        std::vector<float> tex(size_t(im.w) * im.h * out_c);
        const float inv = 1.0f / 255.0f;
        for (size_t k = 0; k < size_t(im.w) * im.h; ++k)
            for (int c = 0; c < out_c; ++c) {
                int src = channel < 0 ? c : (channel < im.c ? channel : 0);
                if (src >= im.c) src = im.c - 1;
                const float v = float(im.px[k * im.c + src]);
                tex[k * out_c + c] = div255 ? v / 255.0f : v * inv;
            }
        *w = im.w;
        *h = im.h;
        return S.AddImage(tex.data(), im.w, im.h, out_c, srgb ? 1 : 0);
    }
    int AddImageTexture(int image_id, int w, int h, int channels) {
        if (channels == 1) {
            pt_texture1 t;
            std::memset(&t, 0, sizeof(t));
            t.type = PT_TEX_IMAGE;
            t.u.image.width = w; t.u.image.height = h; t.u.image.image_id = image_id;
            t.u.image.scale[0] = t.u.image.scale[1] = 1.0f;
            return S.AddTexture1(t);
        }
        pt_texture3 t;
        std::memset(&t, 0, sizeof(t));
        t.type = PT_TEX_IMAGE;
        t.u.image.width = w; t.u.image.height = h; t.u.image.image_id = image_id;
        t.u.image.scale[0] = t.u.image.scale[1] = 1.0f;
        return S.AddTexture3(t);
    }

    // ParseTexture<glm::vec3>, scene.h:76-179 (a bare array is the older files' flat colour)
    int ParseTexture3(const Json &tex, bool is_albedo) {
        if (tex.is_object() && tex.has("PATH") && tex.find("PATH")->is_string()) {
            const std::string raw = tex.find("PATH")->str;
            const std::string p = FileExists(raw) ? raw : Join(m_dir, raw);
            if (FileExists(p)) {
                int w, h;
                const int image = AddImage(Decode(ReadFile(p), 3), -1, false, is_albedo, &w, &h);
                return AddImageTexture(image, w, h, 3);
            }
        }
        const Json *rgb = nullptr;
        if (tex.is_object() && tex.has("RGB") && tex.find("RGB")->is_array() && tex.find("RGB")->arr.size() == 3) rgb = tex.find("RGB");
        else if (tex.is_array() && tex.arr.size() == 3) rgb = &tex;
        if (!rgb) return 0;
        pt_texture3 t;
        std::memset(&t, 0, sizeof(t));
        t.type = PT_TEX_CONSTANT;
        for (int k = 0; k < 3; ++k) t.u.constant.value[k] = float(rgb->arr[k].num);
        return S.AddTexture3(t);
    }
    // ParseTexture<float>
    int ParseTexture1(const Json &tex, int channel) {
        if (tex.is_object() && tex.has("PATH") && tex.find("PATH")->is_string()) {
            const std::string raw = tex.find("PATH")->str;
            const std::string p = FileExists(raw) ? raw : Join(m_dir, raw);
            if (FileExists(p)) {
                int w, h;
                const int image = AddImage(Decode(ReadFile(p), 0), channel, false, false, &w, &h);
                return AddImageTexture(image, w, h, 1);
            }
        }
        double value = 0.0;
        bool have = false;
        if (tex.is_object() && tex.has("VALUE") && tex.find("VALUE")->is_number()) { value = tex.find("VALUE")->num; have = true; }
        else if (tex.is_number()) { value = tex.num; have = true; }
        if (!have) return 0;
        pt_texture1 t;
        std::memset(&t, 0, sizeof(t));
        t.type = PT_TEX_CONSTANT;
        t.u.constant.value = float(value);
        return S.AddTexture1(t);
    }

    // Scene::ParseMaterial, scene.cpp:262-335, plus the older files' spellings
    void ParseMaterial(const Json &m) {
        const std::string t = Lower(StringOr(m, "TYPE", ""));
        pt_material mat;
        std::memset(&mat, 0, sizeof(mat));
        auto albedo_source = [&]() -> const Json * {
            if (const Json *a = m.find("ALBEDO")) return a;
            return m_legacy ? m.find("RGB") : nullptr;
        };
        if (t == "emitting") {
            mat.type = PT_MAT_EMISSIVE;
            float emit = 1.0f;
            if (const Json *r = m.find("RADIANCE")) {
                mat.tex0 = ParseTexture3(*r, false);
                emit = float(r->is_object() ? NumberOr(*r, "EMITTANCE", 1.0) : 1.0);
            } else if (m_legacy && (m.has("COLOR") || m.has("RGB"))) {
                mat.tex0 = ParseTexture3(*(m.has("COLOR") ? m.find("COLOR") : m.find("RGB")), false);
                emit = float(NumberOr(m, "EMITTANCE", 1.0));
            }
            mat.w1.f = emit;
        } else if (t == "diffuse" || t == "lambertian") {
            mat.type = PT_MAT_LAMBERTIAN;
            if (const Json *src = albedo_source()) mat.tex0 = ParseTexture3(*src, true);
        } else if (t == "mirror" || (m_legacy && t == "specular")) {
            mat.type = PT_MAT_MIRROR;
            if (const Json *src = albedo_source()) mat.tex0 = ParseTexture3(*src, true);
        } else if (t == "dielectric" || (m_legacy && t == "roughdielectric")) {
            mat.type = PT_MAT_DIELECTRIC;
            const Json *ior = m.has("INDEX_OF_REFRACTION") ? m.find("INDEX_OF_REFRACTION") : m.find("IOR");
            const Json *src = albedo_source();
            if (src && ior) mat.tex0 = ParseTexture3(*src, true);
            mat.w1.f = float(ior && ior->is_number() ? ior->num : 1.0);
        } else if (t == "metallic_roughness" || (m_legacy && t == "metallicroughness")) {
            mat.type = PT_MAT_METALLIC_ROUGHNESS;
            if (const Json *src = m.find("ALBEDO")) mat.tex0 = ParseTexture3(*src, true);
            int rough = m.has("ROUGHNESS") ? ParseTexture1(*m.find("ROUGHNESS"), 0) : 0;
            int metal = m.has("METALLIC") ? ParseTexture1(*m.find("METALLIC"), 0) : 0;
            if (metal == 0 && rough == 0 && m.has("ORM")) {  // scene.cpp:323-326
                rough = ParseTexture1(*m.find("ORM"), 1);
                metal = ParseTexture1(*m.find("ORM"), 2);
            }
            mat.w1.i = metal;
            mat.w2 = rough;
        } else {
            mat.type = PT_MAT_LAMBERTIAN;  // the reference pushes an uninitialised Material here; defined as the default
        }
        S.AddMaterial(mat);
        ++m_next_material;
    }

    // Scene::ParseObject, scene.cpp:358-518
    std::vector<Placed> ParseObject(const Json &obj) {
        const std::string t = Lower(StringOr(obj, "TYPE", ""));
        int override_material = -1;
        if (const Json *mn = obj.find("MATERIAL"); mn && mn->is_string()) {
            const auto it = m_material_ids.find(mn->str);
            if (it != m_material_ids.end()) override_material = it->second;
        }
        const Mat4 xf = ObjectTransform(obj);
        const bool face_normals = BoolOr(obj, "FACE_NORMALS", false), invert = BoolOr(obj, "INVERT_NORMALS", false);
        std::vector<Placed> out;
        auto single = [&](const MeshData &md) {
            Placed p;
            p.mesh = S.AddMesh(md.p.data(), md.n.data(), md.t.data(), md.vertex_count(), md.i.data(), uint32_t(md.i.size()));
            p.material = override_material >= 0 ? override_material : 0;
            p.transform = xf;
            p.blas = int(S.NewBlasIndex());
            out.push_back(p);
        };
        if (t == "obj") {
            if (!obj.has("PATH")) throw Exception("every obj mesh must have a path to load from");
            single(LoadObj(Join(m_dir, obj.find("PATH")->str), face_normals, invert));
        } else if (t == "gltf") {
            if (!obj.has("PATH")) throw Exception("every gltf mesh must have a path to load from");
            LoadGltf(Join(m_dir, obj.find("PATH")->str), face_normals, invert, xf, override_material, out);
        } else if (m_legacy && (t == "cube" || t == "sphere")) {
            single(t == "cube" ? UnitCube() : UvSphere());
        }
        return out;
    }

    // ---- glTF: GLTFModel::LoadGLTF + the glTF branch of Scene::ParseObject (mesh/gltf_model.cpp:21-339, scene.cpp:425-515)
    struct Gltf {
        Json j;
        std::string dir;
        std::vector<std::string> buffers;
    };
    // a member that must be there (and be of the right kind): malformed files end in an Exception, never in a null dereference
    static const Json &Need(const Json &o, const char *key, Json::Type type) {
        const Json *v = o.find(key);
        if (!v || v->type != type) throw Exception(std::string("glTF: missing or malformed \"") + key + "\"");
        return *v;
    }
    static size_t ComponentSize(int type) {
        switch (type) {
            case 5120: case 5121: return 1;
            case 5122: case 5123: return 2;
            case 5125: case 5126: return 4;
        }
        throw Exception("glTF: unknown accessor component type");
    }
    static int ComponentCount(const std::string &type) {
        if (type == "SCALAR") return 1;
        if (type == "VEC2") return 2;
        if (type == "VEC3") return 3;
        if (type == "VEC4") return 4;
        if (type == "MAT4") return 16;
        throw Exception("glTF: unknown accessor type " + type);
    }
    // element-wise read of an accessor, honouring byteStride (IterateAccessor, mesh/gltf_model.h:189-261); values are converted, not normalised
    static void ReadAccessor(const Gltf &g, int index, std::vector<double> &out, int &ncomp, size_t &count) {
        const std::vector<Json> &accessors = Need(g.j, "accessors", Json::Array).arr, &views = Need(g.j, "bufferViews", Json::Array).arr;
        if (index < 0 || size_t(index) >= accessors.size()) throw Exception("glTF: accessor index out of range");
        const Json &acc = accessors[size_t(index)];
        const size_t view_index = WholeSize(NumberOr(acc, "bufferView", 0), "bufferView");
        if (view_index >= views.size()) throw Exception("glTF: bufferView index out of range");
        const Json &view = views[view_index];
        const int ctype = WholeInt(NumberOr(acc, "componentType", 5126), "componentType");
        const size_t size = ComponentSize(ctype);
        ncomp = ComponentCount(StringOr(acc, "type", "SCALAR"));
        count = WholeSize(NumberOr(acc, "count", 0), "count");
        size_t stride = WholeSize(NumberOr(view, "byteStride", 0), "byteStride");
        if (stride == 0) stride = size * size_t(ncomp);
        const size_t start = WholeSize(NumberOr(view, "byteOffset", 0), "byteOffset") + WholeSize(NumberOr(acc, "byteOffset", 0), "byteOffset");
        const size_t buffer_index = WholeSize(NumberOr(view, "buffer", 0), "buffer");
        if (buffer_index >= g.buffers.size()) throw Exception("glTF: buffer index out of range");
        const std::string &buf = g.buffers[buffer_index];
        if (count > buf.size()) throw Exception("glTF: accessor count exceeds its buffer");  // every element takes at least a byte
        out.resize(count * size_t(ncomp));
        for (size_t e = 0; e < count; ++e)
            for (int c = 0; c < ncomp; ++c) {
                const size_t at = start + e * stride + size_t(c) * size;
                if (at + size > buf.size()) throw Exception("glTF: accessor reads past its buffer");
                const unsigned char *p = reinterpret_cast<const unsigned char *>(buf.data()) + at;
                double v = 0.0;
                switch (ctype) {
                    case 5120: v = double(*reinterpret_cast<const int8_t *>(p)); break;
                    case 5121: v = double(*p); break;
                    case 5122: { int16_t x; std::memcpy(&x, p, 2); v = double(x); break; }
                    case 5123: { uint16_t x; std::memcpy(&x, p, 2); v = double(x); break; }
                    case 5125: { uint32_t x; std::memcpy(&x, p, 4); v = double(x); break; }
                    case 5126: { float x; std::memcpy(&x, p, 4); v = double(x); break; }
                }
                out[e * size_t(ncomp) + size_t(c)] = v;
            }
    }
    std::string ImageBytes(const Gltf &g, int index) const {
        const std::vector<Json> &images = Need(g.j, "images", Json::Array).arr;
        if (index < 0 || size_t(index) >= images.size()) throw Exception("glTF: image index out of range");
        const Json &im = images[size_t(index)];
        if (const Json *uri = im.find("uri"); uri && uri->is_string()) {
            if (uri->str.compare(0, 5, "data:") == 0) return Base64Decode(uri->str, uri->str.find(',') + 1);
            return ReadFile(Join(g.dir, uri->str));
        }
        const std::vector<Json> &views = Need(g.j, "bufferViews", Json::Array).arr;
        const size_t view_index = WholeSize(NumberOr(im, "bufferView", 0), "bufferView");
        if (view_index >= views.size()) throw Exception("glTF: bufferView index out of range");
        const Json &view = views[view_index];
        const size_t off = WholeSize(NumberOr(view, "byteOffset", 0), "byteOffset"), len = WholeSize(NumberOr(view, "byteLength", 0), "byteLength");
        const size_t buffer_index = WholeSize(NumberOr(view, "buffer", 0), "buffer");
        if (buffer_index >= g.buffers.size() || off > g.buffers[buffer_index].size()) throw Exception("glTF: image bufferView out of range");
        return g.buffers[buffer_index].substr(off, len);
    }

    void LoadGltf(const std::string &path, bool face_normals, bool invert_normals, const Mat4 &instance_transform, int override_material,
                  std::vector<Placed> &out) {
        Gltf g;
        g.j = JsonParser(ReadFile(path)).Parse();
        g.dir = DirName(path);
        if (const Json *bufs = g.j.find("buffers"))
            for (const Json &b : bufs->arr) {
                const std::string uri = StringOr(b, "uri", "");
                g.buffers.push_back(uri.compare(0, 5, "data:") == 0 ? Base64Decode(uri, uri.find(',') + 1) : ReadFile(Join(g.dir, uri)));
            }
        // ---- meshes -> one aggregate mesh; primitives remember their index ranges
        struct Prim { uint32_t first_index, index_count, vertex_count; int material; };
        std::vector<std::vector<Prim>> mesh_prims;
        MeshData agg;
        if (const Json *meshes = g.j.find("meshes"))
            for (const Json &mesh : meshes->arr) {
                std::vector<Prim> prims;
                for (const Json &prim : Need(mesh, "primitives", Json::Array).arr) {
                    if (WholeInt(NumberOr(prim, "mode", 4), "mode") != 4) continue;
                    const Json &attrs = Need(prim, "attributes", Json::Object);
                    std::vector<double> pos, nrm, uv, idx;
                    int nc;
                    size_t vc, cnt;
                    ReadAccessor(g, WholeInt(Need(attrs, "POSITION", Json::Number).num, "POSITION"), pos, nc, vc);
                    if (nc != 3) throw Exception("glTF: POSITION must be VEC3");
                    const bool use_face = face_normals || !attrs.has("NORMAL");
                    if (!use_face) {
                        ReadAccessor(g, WholeInt(Need(attrs, "NORMAL", Json::Number).num, "NORMAL"), nrm, nc, cnt);
                        if (nc != 3 || cnt != vc) throw Exception("glTF: NORMAL must be VEC3 with one entry per vertex");
                    }
                    const bool has_uv = attrs.has("TEXCOORD_0");
                    if (has_uv) {
                        ReadAccessor(g, WholeInt(Need(attrs, "TEXCOORD_0", Json::Number).num, "TEXCOORD_0"), uv, nc, cnt);
                        if (nc != 2 || cnt != vc) throw Exception("glTF: TEXCOORD_0 must be VEC2 with one entry per vertex");
                    }
                    std::vector<uint32_t> indices;
                    if (prim.has("indices")) {
                        ReadAccessor(g, WholeInt(Need(prim, "indices", Json::Number).num, "indices"), idx, nc, cnt);
                        indices.resize(idx.size());
                        for (size_t k = 0; k < idx.size(); ++k) {
                            if (!(idx[k] >= 0.0 && idx[k] < double(vc))) throw Exception("glTF: vertex index out of range");
                            indices[k] = uint32_t(idx[k]);
                        }
                    } else {
                        indices.resize(vc);
                        for (size_t k = 0; k < vc; ++k) indices[k] = uint32_t(k);
                    }
                    const uint32_t nverts = agg.vertex_count(), first_index = uint32_t(agg.i.size());
                    uint32_t out_vc;
                    auto position = [&](size_t v) { return Vec3{float(pos[3 * v]), float(pos[3 * v + 1]), float(pos[3 * v + 2])}; };
                    if (!use_face) {
                        for (uint32_t k : indices) agg.i.push_back(k + nverts);
                        for (size_t v = 0; v < vc; ++v) {
                            Push(agg.p, position(v));
                            Push(agg.n, Normalize(Vec3{float(nrm[3 * v]), float(nrm[3 * v + 1]), float(nrm[3 * v + 2])}));
                            agg.t.push_back(has_uv ? float(uv[2 * v]) : 0.0f);
                            agg.t.push_back(has_uv ? float(uv[2 * v + 1]) : 0.0f);
                        }
                        out_vc = uint32_t(vc);
                    } else {  // de-indexed, one normal per face
                        const size_t ntri = indices.size() / 3;
                        for (size_t tri = 0; tri < ntri; ++tri) {
                            const Vec3 p0 = position(indices[3 * tri]), p1 = position(indices[3 * tri + 1]), p2 = position(indices[3 * tri + 2]);
                            const Vec3 fn = Normalize(Cross(p1 - p0, p2 - p0));
                            for (int c = 0; c < 3; ++c) {
                                const size_t v = indices[3 * tri + size_t(c)];
                                Push(agg.p, position(v));
                                Push(agg.n, fn);
                                agg.t.push_back(has_uv ? float(uv[2 * v]) : 0.0f);
                                agg.t.push_back(has_uv ? float(uv[2 * v + 1]) : 0.0f);
                                agg.i.push_back(nverts + uint32_t(3 * tri) + uint32_t(c));
                            }
                        }
                        out_vc = uint32_t(indices.size());
                    }
                    const int material = std::max(WholeInt(NumberOr(prim, "material", 0), "material"), 0);
                    prims.push_back(Prim{first_index, uint32_t(indices.size()), out_vc, material});
                }
                mesh_prims.push_back(std::move(prims));
            }
        if (agg.i.empty() || agg.p.empty()) return;  // "Skipping glTF object since it has zero vertices"
        if (invert_normals)
            for (float &x : agg.n) x = -x;
        const TriangleMesh whole = S.AddMesh(agg.p.data(), agg.n.data(), agg.t.data(), agg.vertex_count(), agg.i.data(), uint32_t(agg.i.size()));

        // ---- materials, appended to the scene's list; images are decoded once per texture source and object load
        const int base_material = m_next_material;
        std::map<int, int> albedo_map, metallic_map, rough_map;
        if (const Json *mats = g.j.find("materials"))
            for (const Json &mat : mats->arr) {
                pt_material pm;
                std::memset(&pm, 0, sizeof(pm));
                pm.type = PT_MAT_METALLIC_ROUGHNESS;  // absent textures fall back to texture 0
                const Json *pbr = mat.find("pbrMetallicRoughness");
                auto source_of = [&](const Json &ref) {
                    const std::vector<Json> &textures = Need(g.j, "textures", Json::Array).arr;
                    const size_t index = WholeSize(NumberOr(ref, "index", 0), "index");
                    if (index >= textures.size()) throw Exception("glTF: texture index out of range");
                    return WholeInt(NumberOr(textures[index], "source", 0), "source");
                };
                if (pbr && pbr->has("baseColorTexture")) {
                    const int src = source_of(*pbr->find("baseColorTexture"));
                    if (!albedo_map.count(src)) {
                        int w, h;
                        const int image = AddImage(Decode(ImageBytes(g, src), 4), -1, true, true, &w, &h);
                        albedo_map[src] = AddImageTexture(image, w, h, 3);
                    }
                    pm.tex0 = albedo_map[src];
                }
                if (pbr && pbr->has("metallicRoughnessTexture")) {
                    const int src = source_of(*pbr->find("metallicRoughnessTexture"));
                    if (!metallic_map.count(src)) {
                        const Pixels px = Decode(ImageBytes(g, src), 4);
                        int w, h;
                        int image = AddImage(px, 2, true, false, &w, &h);  // B = metallic
                        metallic_map[src] = AddImageTexture(image, w, h, 1);
                        image = AddImage(px, 1, true, false, &w, &h);      // G = roughness
                        rough_map[src] = AddImageTexture(image, w, h, 1);
                    }
                    pm.w1.i = metallic_map[src];
                    pm.w2 = rough_map[src];
                }
                S.AddMaterial(pm);
                ++m_next_material;
            }

        // ---- node hierarchy: world = parent * (T * R * S * matrix), visited like ForEachNode (roots pushed in order, popped from the back)
        struct Node { Mat4 world; int mesh; std::vector<int> children; };
        std::vector<Node> nodes;
        const std::vector<Json> &jnodes = Need(g.j, "nodes", Json::Array).arr;
        auto numbers = [](const Json &v, size_t count, float *out) {
            if (v.type != Json::Array || v.arr.size() < count) throw Exception("glTF: node transform with too few numbers");
            for (size_t k = 0; k < count; ++k) out[k] = float(v.arr[k].num);
        };
        // A node that is met again below itself is a cycle: refused at once (the recursion never gets deeper than the
        // file's node count, and never deeper than kMaxNodeDepth, so a damaged file ends in an Exception, not in a
        // stack overflow).  A node shared by several parents is loaded once per parent, bounded by the total below.
        constexpr int kMaxNodeDepth = 512;
        std::vector<char> on_path(jnodes.size(), 0);
        int node_depth = 0;
        std::function<int(int, const Mat4 &)> load_node = [&](int ni, const Mat4 &parent) {
            if (ni < 0 || size_t(ni) >= jnodes.size()) throw Exception("glTF: node index out of range");
            if (on_path[size_t(ni)]) throw Exception("glTF: node hierarchy has a cycle");
            if (node_depth >= kMaxNodeDepth) throw Exception("glTF: node hierarchy deeper than 512 levels");
            if (nodes.size() > jnodes.size() * 64 + 4096) throw Exception("glTF: node hierarchy is not a tree");
            struct OnPath {
                char &flag;
                int &depth;
                OnPath(char &f, int &d) : flag(f), depth(d) { flag = 1; ++depth; }
                ~OnPath() { flag = 0; --depth; }
            } guard(on_path[size_t(ni)], node_depth);
            const Json &n = jnodes[size_t(ni)];
            const Mat4 I = Mat4::Identity();
            Vec3 t{0, 0, 0}, s{1, 1, 1};
            float q[4] = {0, 0, 0, 1};
            Mat4 matrix = I;
            float f3[3];
            if (const Json *v = n.find("translation")) { numbers(*v, 3, f3); t = {f3[0], f3[1], f3[2]}; }
            if (const Json *v = n.find("scale")) { numbers(*v, 3, f3); s = {f3[0], f3[1], f3[2]}; }
            if (const Json *v = n.find("rotation")) numbers(*v, 4, q);
            if (const Json *v = n.find("matrix")) numbers(*v, 16, matrix.m);
            const Mat4 local = Mul(Mul(Mul(Translate(I, t), FromQuat(q[0], q[1], q[2], q[3])), Scale(I, s)), matrix);
            Node node;
            node.world = Mul(parent, local);
            node.mesh = WholeInt(NumberOr(n, "mesh", -1), "mesh");
            const int self = int(nodes.size());
            nodes.push_back(node);
            if (const Json *ch = n.find("children"))
                for (const Json &c : ch->arr) {
                    const int child = load_node(WholeInt(c.num, "node index"), nodes[size_t(self)].world);
                    nodes[size_t(self)].children.push_back(child);
                }
            return self;
        };
        std::vector<int> stack;
        const int scene_index = WholeInt(NumberOr(g.j, "scene", 0), "scene");
        const std::vector<Json> &scenes = Need(g.j, "scenes", Json::Array).arr;
        if (scene_index < 0 || size_t(scene_index) >= scenes.size()) throw Exception("glTF: scene index out of range");
        for (const Json &r : Need(scenes[size_t(scene_index)], "nodes", Json::Array).arr) stack.push_back(load_node(WholeInt(r.num, "node index"), Mat4::Identity()));
        std::vector<int> order;
        while (!stack.empty()) {
            const int cur = stack.back();
            stack.pop_back();
            order.push_back(cur);
            for (int c : nodes[size_t(cur)].children) stack.push_back(c);
        }
        std::map<std::pair<int, int>, int> blas_indices;
        for (int ni : order) {
            const Node &node = nodes[size_t(ni)];
            if (node.mesh < 0) continue;
            const Mat4 world = Mul(instance_transform, node.world);
            const auto &prims = mesh_prims.at(size_t(node.mesh));
            for (size_t pi = 0; pi < prims.size(); ++pi) {
                const auto key = std::make_pair(node.mesh, int(pi));
                if (!blas_indices.count(key)) blas_indices[key] = int(S.NewBlasIndex());
                Placed p;
                p.mesh.first_index = whole.first_index + prims[pi].first_index;
                p.mesh.index_count = prims[pi].index_count;
                p.mesh.vertex_count = prims[pi].vertex_count;
                p.material = override_material >= 0 ? override_material : base_material + prims[pi].material;
                p.transform = world;
                p.blas = blas_indices[key];
                out.push_back(p);
            }
        }
    }
};

}  // namespace

// Scene::LoadScene (scene.cpp:649-687) up to the acceleration structures: the caller finishes with
// Finalize() or Finalize(ctx).
void LoadSceneFile(Scene &scene, const std::string &path, pth_image_decoder decode, void *user, int width, int height, int max_depth) {
    Loader(scene, decode, user).LoadSceneFile(path, width, height, max_depth);
}

}  // namespace pth
