// gltf_loader.cc — glTF 2.0 (.glb / .gltf) to app::Model, following ModelLoader of the reference
// (src/model_loader.cc:17-214), which does this through fastgltf.  Reproduced behaviour:
//   * accepted extensions: KHR_lights_punctual and KHR_materials_specular (parsed, unused), KHR_materials_ior,
//     _volume, _transmission, _emissive_strength (:23-29); any other REQUIRED extension is an error;
//   * a primitive without indices gets 0..n-1 (the GenerateMeshIndices option, :33-36);
//   * a texture names an image through `source`, image 0 when absent (:74-94);
//   * attenuation = (1 - attenuationColor) / max(1e-5, attenuationDistance) — the Blender convention the reference
//     prefers over the extension's -log form (:95-104); f0 from the ior (:114);
//   * one default material appended last, used by primitives without one (:126, :147);
//   * one Mesh per primitive; TEXCOORD_0 only, uv = 0 when absent; tangents from TANGENT, else generated (:134-194);
//   * the default scene's node tree flattened: one Object per primitive of a node's mesh with
//     NormalMatrix = transpose(inverse(Model)); cameras with their node matrix (:195-210).
// Meshes with uv but no TANGENT get MikkTSpace tangents (mikk_tangents.cc, a restatement of the published algorithm —
// the library is not in the reference's tree); the no-uv fallback is the reference's (compute_tangents.cc:64-70).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>

#include "app_model.h"
#include "image_decode.h"

namespace app {
namespace {

[[noreturn]] void fail(const std::string& what) { throw std::runtime_error("Failed to load glTF: " + what); }

// ------------------------------------------------------------------------------------------------ JSON
struct Json {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json* find(const char* key) const {
        if (type != Object) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Json& at(size_t i) const {
        if (type != Array || i >= arr.size()) fail("index out of range in the JSON");
        return arr[i];
    }
    size_t size() const { return type == Array ? arr.size() : 0; }
    double number(const char* key, double def) const {
        const Json* j = find(key);
        return (j && j->type == Number) ? j->num : def;
    }
    long index(const char* key) const {  // non-negative integer member, -1 when absent
        const Json* j = find(key);
        return (j && j->type == Number && j->num >= 0) ? static_cast<long>(j->num) : -1;
    }
    std::string text(const char* key, const char* def = "") const {
        const Json* j = find(key);
        return (j && j->type == String) ? j->str : def;
    }
};

class JsonParser {
  public:
    JsonParser(const char* b, const char* e) : p_(b), e_(e) {}
    Json parse() {
        if (e_ - p_ >= 3 && !std::memcmp(p_, "\xef\xbb\xbf", 3)) p_ += 3;
        Json j = value(0);
        ws();
        return j;
    }

  private:
    const char* p_;
    const char* e_;
    void ws() { while (p_ < e_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_; }
    char peek() { ws(); if (p_ >= e_) fail("unexpected end of JSON"); return *p_; }
    void expect(char c) { if (peek() != c) fail(std::string("JSON: expected '") + c + "'"); ++p_; }
    Json value(int depth) {
        if (depth > 200) fail("JSON nested too deeply");
        Json j;
        const char c = peek();
        if (c == '{') {
            ++p_;
            j.type = Json::Object;
            if (peek() == '}') { ++p_; return j; }
            for (;;) {
                if (peek() != '"') fail("JSON: expected a member name");
                std::string key = string();
                expect(':');
                j.obj.emplace_back(std::move(key), value(depth + 1));
                if (peek() == ',') { ++p_; continue; }
                expect('}');
                return j;
            }
        }
        if (c == '[') {
            ++p_;
            j.type = Json::Array;
            if (peek() == ']') { ++p_; return j; }
            for (;;) {
                j.arr.push_back(value(depth + 1));
                if (peek() == ',') { ++p_; continue; }
                expect(']');
                return j;
            }
        }
        if (c == '"') { j.type = Json::String; j.str = string(); return j; }
        if (c == 't' || c == 'f' || c == 'n') {
            auto lit = [&](const char* s) { const size_t n = std::strlen(s); if (static_cast<size_t>(e_ - p_) < n || std::memcmp(p_, s, n)) fail("JSON: bad literal"); p_ += n; };
            if (c == 't') { lit("true"); j.type = Json::Bool; j.b = true; }
            else if (c == 'f') { lit("false"); j.type = Json::Bool; }
            else lit("null");
            return j;
        }
        const char* start = p_;
        while (p_ < e_ && (std::strchr("+-.eE", *p_) || (*p_ >= '0' && *p_ <= '9'))) ++p_;
        if (p_ == start) fail("JSON: unexpected character");
        j.type = Json::Number;
        j.num = std::strtod(std::string(start, p_).c_str(), nullptr);
        return j;
    }
    std::string string() {
        ++p_;  // opening quote
        std::string s;
        while (p_ < e_ && *p_ != '"') {
            if (*p_ != '\\') { s.push_back(*p_++); continue; }
            if (++p_ >= e_) break;
            const char c = *p_++;
            switch (c) {
                case 'n': s.push_back('\n'); break;
                case 't': s.push_back('\t'); break;
                case 'r': s.push_back('\r'); break;
                case 'b': s.push_back('\b'); break;
                case 'f': s.push_back('\f'); break;
                case 'u': {
                    if (e_ - p_ < 4) fail("JSON: bad \\u escape");
                    unsigned cp = static_cast<unsigned>(std::strtoul(std::string(p_, p_ + 4).c_str(), nullptr, 16));
                    p_ += 4;
                    if (cp < 0x80) s.push_back(static_cast<char>(cp));
                    else if (cp < 0x800) { s.push_back(static_cast<char>(0xc0 | (cp >> 6))); s.push_back(static_cast<char>(0x80 | (cp & 63))); }
                    else { s.push_back(static_cast<char>(0xe0 | (cp >> 12))); s.push_back(static_cast<char>(0x80 | ((cp >> 6) & 63))); s.push_back(static_cast<char>(0x80 | (cp & 63))); }
                    break;
                }
                default: s.push_back(c);
            }
        }
        if (p_ >= e_) fail("JSON: unterminated string");
        ++p_;
        return s;
    }
};

std::vector<uint8_t> read_file(const std::filesystem::path& p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) fail("cannot open " + p.string());
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

std::vector<uint8_t> base64(const std::string& s, size_t from) {
    std::vector<uint8_t> out;
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = from; i < s.size(); ++i) {
        const char c = s[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else continue;  // padding, line breaks
        acc = (acc << 6) | static_cast<uint32_t>(v);
        bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back(static_cast<uint8_t>(acc >> bits)); }
    }
    return out;
}

std::string percent_decode(const std::string& s) {
    std::string o;
    for (size_t i = 0; i < s.size(); ++i)
        if (s[i] == '%' && i + 2 < s.size() + 0 && std::isxdigit(static_cast<unsigned char>(s[i + 1])) && std::isxdigit(static_cast<unsigned char>(s[i + 2]))) {
            o.push_back(static_cast<char>(std::strtoul(s.substr(i + 1, 2).c_str(), nullptr, 16)));
            i += 2;
        } else o.push_back(s[i]);
    return o;
}

std::vector<uint8_t> load_uri(const std::string& uri, const std::filesystem::path& dir) {
    if (uri.rfind("data:", 0) == 0) {
        const size_t comma = uri.find(',');
        if (comma == std::string::npos || uri.find(";base64") == std::string::npos) fail("unsupported data URI");
        return base64(uri, comma + 1);
    }
    return read_file(dir / percent_decode(uri));
}

// ------------------------------------------------------------------------------------------------ matrices (column-major)
struct M4 {
    float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float& at(int c, int r) { return m[c * 4 + r]; }
    float at(int c, int r) const { return m[c * 4 + r]; }
};
M4 mul(const M4& a, const M4& b) {
    M4 r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float t = a.at(0, i) * b.at(j, 0);
            t = t + a.at(1, i) * b.at(j, 1);
            t = t + a.at(2, i) * b.at(j, 2);
            t = t + a.at(3, i) * b.at(j, 3);
            r.at(j, i) = t;
        }
    return r;
}
M4 inverse(const M4& m) {  // cofactor expansion in the order glm::inverse(mat4) uses
    auto e = [&](int c, int r) { return m.at(c, r); };
    const float c00 = e(2, 2) * e(3, 3) - e(3, 2) * e(2, 3), c02 = e(1, 2) * e(3, 3) - e(3, 2) * e(1, 3), c03 = e(1, 2) * e(2, 3) - e(2, 2) * e(1, 3);
    const float c04 = e(2, 1) * e(3, 3) - e(3, 1) * e(2, 3), c06 = e(1, 1) * e(3, 3) - e(3, 1) * e(1, 3), c07 = e(1, 1) * e(2, 3) - e(2, 1) * e(1, 3);
    const float c08 = e(2, 1) * e(3, 2) - e(3, 1) * e(2, 2), c10 = e(1, 1) * e(3, 2) - e(3, 1) * e(1, 2), c11 = e(1, 1) * e(2, 2) - e(2, 1) * e(1, 2);
    const float c12 = e(2, 0) * e(3, 3) - e(3, 0) * e(2, 3), c14 = e(1, 0) * e(3, 3) - e(3, 0) * e(1, 3), c15 = e(1, 0) * e(2, 3) - e(2, 0) * e(1, 3);
    const float c16 = e(2, 0) * e(3, 2) - e(3, 0) * e(2, 2), c18 = e(1, 0) * e(3, 2) - e(3, 0) * e(1, 2), c19 = e(1, 0) * e(2, 2) - e(2, 0) * e(1, 2);
    const float c20 = e(2, 0) * e(3, 1) - e(3, 0) * e(2, 1), c22 = e(1, 0) * e(3, 1) - e(3, 0) * e(1, 1), c23 = e(1, 0) * e(2, 1) - e(2, 0) * e(1, 1);
    const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    const float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const float v0[4] = {e(1, 0), e(0, 0), e(0, 0), e(0, 0)}, v1[4] = {e(1, 1), e(0, 1), e(0, 1), e(0, 1)};
    const float v2[4] = {e(1, 2), e(0, 2), e(0, 2), e(0, 2)}, v3[4] = {e(1, 3), e(0, 3), e(0, 3), e(0, 3)};
    const float sa[4] = {+1, -1, +1, -1}, sb[4] = {-1, +1, -1, +1};
    M4 inv;
    for (int i = 0; i < 4; ++i) {
        inv.at(0, i) = ((v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i]) * sa[i];
        inv.at(1, i) = ((v0[i] * f0[i] - v2[i] * f3[i]) + v3[i] * f4[i]) * sb[i];
        inv.at(2, i) = ((v0[i] * f1[i] - v1[i] * f3[i]) + v3[i] * f5[i]) * sa[i];
        inv.at(3, i) = ((v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i]) * sb[i];
    }
    const float det = (e(0, 0) * inv.at(0, 0) + e(0, 1) * inv.at(1, 0)) + (e(0, 2) * inv.at(2, 0) + e(0, 3) * inv.at(3, 0));
    const float ood = 1.0f / det;
    for (float& v : inv.m) v = v * ood;
    return inv;
}
M4 transpose(const M4& m) {
    M4 r;
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < 4; ++k) r.at(c, k) = m.at(k, c);
    return r;
}
fmat4x4 to_fmat(const M4& m) {
    fmat4x4 r;
    std::memcpy(&r, m.m, 64);
    return r;
}

// ------------------------------------------------------------------------------------------------ the asset
struct Asset {
    Json root;
    std::filesystem::path dir;
    std::vector<std::vector<uint8_t>> buffers;

    const Json& list(const char* name) const {
        static const Json empty = [] { Json j; j.type = Json::Array; return j; }();
        const Json* j = root.find(name);
        return (j && j->type == Json::Array) ? *j : empty;
    }
    // A size, offset or count from the document: a JSON number that must be a non-negative integer small enough to be exact
    // in a double (anything else would be undefined behaviour in the conversion, or land near 2^64).
    static size_t size_field(const Json& obj, const char* name) {
        const double v = obj.number(name, 0);
        if (!(v >= 0.0) || v > 9007199254740992.0 || v != std::floor(v)) fail(std::string("bad ") + name + " (not a non-negative integer)");
        return static_cast<size_t>(v);
    }
    struct View {
        const uint8_t* data;
        size_t length, stride;
    };
    View view(long index) const {
        const Json& v = list("bufferViews").at(static_cast<size_t>(index));
        const long b = v.index("buffer");
        if (b < 0 || static_cast<size_t>(b) >= buffers.size()) fail("bufferView without a valid buffer");
        const size_t off = size_field(v, "byteOffset"), len = size_field(v, "byteLength"), stride = size_field(v, "byteStride");
        const size_t total = buffers[b].size();
        if (off > total || len > total - off) fail("bufferView exceeds its buffer");
        if (stride > 252) fail("bufferView with a byteStride above 252");  // the format's limit
        return View{buffers[b].data() + off, len, stride};
    }
    struct Accessor {
        const uint8_t* data = nullptr;
        size_t count = 0, stride = 0;
        int component = 0, width = 0;  // glTF component type, components per element
        bool normalized = false;
    };
    Accessor accessor(long index, const char* what) const {
        const Json& a = list("accessors").at(static_cast<size_t>(index));
        if (a.find("sparse")) fail("sparse accessors are not supported");
        const long bv = a.index("bufferView");
        if (bv < 0) fail(std::string("Malformed GLTF: No ") + what + ".");
        Accessor r;
        r.component = static_cast<int>(a.number("componentType", 0));
        const std::string type = a.text("type");
        r.width = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : type == "MAT4" ? 16 : 0;
        int csize;
        switch (r.component) {
            case 5120: case 5121: csize = 1; break;
            case 5122: case 5123: csize = 2; break;
            case 5125: case 5126: csize = 4; break;
            default: fail("accessor with an unknown component type");
        }
        if (!r.width) fail("accessor with an unsupported element type");
        r.count = size_field(a, "count");
        const Json* n = a.find("normalized");
        r.normalized = n && n->type == Json::Bool && n->b;
        const View v = view(bv);
        const size_t off = size_field(a, "byteOffset");
        const size_t elem = static_cast<size_t>(csize) * r.width;
        r.stride = v.stride ? v.stride : elem;
        if (r.stride < elem) fail("accessor elements overlap: byteStride below the element size");
        // range check without additions or products that could wrap: (count - 1) * stride <= length - off - elem
        if (r.count) {
            if (off > v.length || elem > v.length - off) fail("accessor exceeds its bufferView");
            if ((r.count - 1) > (v.length - off - elem) / r.stride) fail("accessor exceeds its bufferView");
        }
        r.data = v.data + off;
        return r;
    }
};

float component_as_float(const uint8_t* p, int component, bool normalized) {
    switch (component) {
        case 5126: { float f; std::memcpy(&f, p, 4); return f; }
        case 5125: { uint32_t v; std::memcpy(&v, p, 4); return static_cast<float>(v); }
        case 5123: { uint16_t v; std::memcpy(&v, p, 2); return normalized ? static_cast<float>(v) / 65535.0f : static_cast<float>(v); }
        case 5122: { int16_t v; std::memcpy(&v, p, 2); return normalized ? std::max(static_cast<float>(v) / 32767.0f, -1.0f) : static_cast<float>(v); }
        case 5121: { return normalized ? static_cast<float>(*p) / 255.0f : static_cast<float>(*p); }
        default: { const int8_t v = static_cast<int8_t>(*p); return normalized ? std::max(static_cast<float>(v) / 127.0f, -1.0f) : static_cast<float>(v); }
    }
}
void element_as_floats(const Asset::Accessor& a, size_t i, float* out, int n) {
    const int csize = (a.component == 5120 || a.component == 5121) ? 1 : (a.component == 5122 || a.component == 5123) ? 2 : 4;
    for (int k = 0; k < n; ++k) out[k] = (k < a.width) ? component_as_float(a.data + i * a.stride + static_cast<size_t>(k) * csize, a.component, a.normalized) : 0.0f;
}

// ------------------------------------------------------------------------------------------------ tangents
fvec3 sub3(fvec3 a, fvec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
fvec3 cross3(fvec3 a, fvec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
float dot3(fvec3 a, fvec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

void compute_tangents_no_uv(Mesh& mesh) {  // compute_tangents.cc:64-70
    for (vertex& v : mesh.vertices) {
        const fvec3 diff = sub3(v.normal, fvec3{0.0f, 1.0f, 0.0f});
        const fvec3 up = (dot3(diff, diff) > 1e-4f) ? fvec3{0.0f, 1.0f, 0.0f} : fvec3{0.0f, -1.0f, 0.0f};
        const fvec3 t = cross3(v.normal, up);
        v.tangent = fvec4{t.x, t.y, t.z, 1.0f};
    }
}
int texture_image(const Asset& a, const Json* tex_info) {  // model_loader.cc:74-94
    if (!tex_info) return -1;
    const long t = tex_info->index("index");
    if (t < 0) return -1;
    const long src = a.list("textures").at(static_cast<size_t>(t)).index("source");
    return src < 0 ? 0 : static_cast<int>(src);
}

}  // namespace

Model load_gltf_model(const std::filesystem::path& path) {
    Asset asset;
    asset.dir = path.parent_path();
    const std::vector<uint8_t> file = read_file(path);
    std::vector<uint8_t> glb_bin;
    bool has_glb_bin = false;
    if (file.size() >= 12 && !std::memcmp(file.data(), "glTF", 4)) {
        uint32_t version, total;
        std::memcpy(&version, file.data() + 4, 4);
        std::memcpy(&total, file.data() + 8, 4);
        if (version != 2 || total > file.size()) fail("not a version 2 GLB container");
        size_t pos = 12;
        bool have_json = false;
        while (pos + 8 <= total) {
            uint32_t len, type;
            std::memcpy(&len, file.data() + pos, 4);
            std::memcpy(&type, file.data() + pos + 4, 4);
            if (pos + 8ull + len > total) fail("truncated GLB chunk");
            const char* body = reinterpret_cast<const char*>(file.data() + pos + 8);
            if (type == 0x4e4f534a && !have_json) { asset.root = JsonParser(body, body + len).parse(); have_json = true; }
            else if (type == 0x004e4942 && !has_glb_bin) { glb_bin.assign(body, body + len); has_glb_bin = true; }
            pos += 8ull + len;
        }
        if (!have_json) fail("GLB without a JSON chunk");
    } else {
        const char* b = reinterpret_cast<const char*>(file.data());
        asset.root = JsonParser(b, b + file.size()).parse();
    }
    if (asset.root.type != Json::Object) fail("the document is not a JSON object");

    static const char* supported[] = {"KHR_lights_punctual", "KHR_materials_specular", "KHR_materials_ior", "KHR_materials_volume",
                                      "KHR_materials_transmission", "KHR_materials_emissive_strength"};
    if (const Json* req = asset.root.find("extensionsRequired"))
        for (const Json& e : req->arr) {
            bool ok = false;
            for (const char* s : supported) ok = ok || e.str == s;
            if (!ok) fail("required extension " + e.str + " is not supported");
        }

    // buffers
    const Json& jbuffers = asset.list("buffers");
    for (size_t i = 0; i < jbuffers.size(); ++i) {
        const Json& b = jbuffers.at(i);
        const std::string uri = b.text("uri");
        if (uri.empty()) {
            if (i != 0 || !has_glb_bin) fail("buffer without data");
            asset.buffers.push_back(glb_bin);
        } else {
            asset.buffers.push_back(load_uri(uri, asset.dir));
        }
        if (asset.buffers.back().size() < static_cast<size_t>(b.number("byteLength", 0))) fail("buffer shorter than its byteLength");
    }

    Model model;
    // images (CPUTexture<sdr_pixel>(image, asset), cpu_framebuffer.cc:36-83)
    for (const Json& im : asset.list("images").arr) {
        std::vector<uint8_t> bytes;
        const std::string uri = im.text("uri");
        const Asset::View* view = nullptr;
        Asset::View v{};
        if (!uri.empty()) bytes = load_uri(uri, asset.dir);
        else if (im.index("bufferView") >= 0) { v = asset.view(im.index("bufferView")); view = &v; }
        else fail("image without a source");
        const DecodedImage d = view ? decode_image(view->data, view->length) : decode_image(bytes.data(), bytes.size());
        std::vector<sdr_pixel> px(static_cast<size_t>(d.width) * d.height);
        std::memcpy(px.data(), d.rgba.data(), d.rgba.size());
        model.images_.emplace_back(d.width, d.height, std::move(px));
    }

    // materials (model_loader.cc:66-126)
    for (const Json& m : asset.list("materials").arr) {
        Material mat;
        const Json* pbr = m.find("pbrMetallicRoughness");
        if (pbr) {
            if (const Json* f = pbr->find("baseColorFactor"); f && f->size() == 4)
                mat.baseColorFactor = {static_cast<float>(f->arr[0].num), static_cast<float>(f->arr[1].num), static_cast<float>(f->arr[2].num), static_cast<float>(f->arr[3].num)};
            mat.metallicFactor = static_cast<float>(pbr->number("metallicFactor", 1.0));
            mat.roughnessFactor = static_cast<float>(pbr->number("roughnessFactor", 1.0));
            mat.baseColorTextureIndex = texture_image(asset, pbr->find("baseColorTexture"));
            mat.metallicRoughnessTextureIndex = texture_image(asset, pbr->find("metallicRoughnessTexture"));
        }
        mat.normalTextureIndex = texture_image(asset, m.find("normalTexture"));
        mat.emissiveTextureIndex = texture_image(asset, m.find("emissiveTexture"));
        if (const Json* f = m.find("emissiveFactor"); f && f->size() == 3)
            mat.emissiveFactor = {static_cast<float>(f->arr[0].num), static_cast<float>(f->arr[1].num), static_cast<float>(f->arr[2].num)};
        const Json* ext = m.find("extensions");
        const Json* e_ior = ext ? ext->find("KHR_materials_ior") : nullptr;
        const Json* e_tr = ext ? ext->find("KHR_materials_transmission") : nullptr;
        const Json* e_vol = ext ? ext->find("KHR_materials_volume") : nullptr;
        const Json* e_em = ext ? ext->find("KHR_materials_emissive_strength") : nullptr;
        mat.ior = e_ior ? static_cast<float>(e_ior->number("ior", 1.5)) : 1.5f;
        {   // f0_dielectric (brdf.cc:70-72)
            const float q = (mat.ior - 1.0f) / (mat.ior + 1.0f);
            mat.dielectric_f0 = q * q;
        }
        if (e_tr) {
            mat.transmisionFactor = static_cast<float>(e_tr->number("transmissionFactor", 0.0));
            mat.transmissionTextureIndex = texture_image(asset, e_tr->find("transmissionTexture"));
        }
        if (e_vol) {
            fvec3 color{1.0f, 1.0f, 1.0f};
            if (const Json* f = e_vol->find("attenuationColor"); f && f->size() == 3)
                color = {static_cast<float>(f->arr[0].num), static_cast<float>(f->arr[1].num), static_cast<float>(f->arr[2].num)};
            const float dist = static_cast<float>(e_vol->number("attenuationDistance", std::numeric_limits<double>::infinity()));
            const float den = std::max(1e-5f, dist);
            mat.attenuationFactor = {(1.0f - color.x) / den, (1.0f - color.y) / den, (1.0f - color.z) / den};
            mat.hasVolume = true;
        }
        mat.emissiveStrength = e_em ? static_cast<float>(e_em->number("emissiveStrength", 1.0)) : 1.0f;
        if (const Json* d = m.find("doubleSided")) mat.doubleSided = d->type == Json::Bool && d->b;
        const std::string mode = m.text("alphaMode", "OPAQUE");
        mat.alphaBlending = mode == "BLEND";
        mat.alpha_cutoff = mode == "MASK" ? static_cast<float>(m.number("alphaCutoff", 0.5)) : -1.0f;
        model.materials_.push_back(mat);
    }
    model.materials_.emplace_back();  // default material at the last index

    // meshes: one per primitive (model_loader.cc:128-194)
    std::vector<std::pair<uint32_t, uint32_t>> mesh_ids;  // glTF mesh -> (first primitive mesh, count)
    uint32_t count = 0;
    for (const Json& gm : asset.list("meshes").arr) {
        const Json* prims = gm.find("primitives");
        const size_t n_prims = prims ? prims->size() : 0;
        for (size_t pi = 0; pi < n_prims; ++pi) {
            const Json& prim = prims->arr[pi];
            const long mode = prim.index("mode");
            if (mode >= 0 && mode != 4) fail("only triangle-list primitives are supported");
            const Json* attrs = prim.find("attributes");
            if (!attrs || attrs->index("POSITION") < 0) fail("a primitive without POSITION");
            if (attrs->index("NORMAL") < 0) fail("a primitive without NORMAL");
            const Asset::Accessor pos = asset.accessor(attrs->index("POSITION"), "positions");
            const Asset::Accessor nrm = asset.accessor(attrs->index("NORMAL"), "normals");
            const bool has_uv = attrs->index("TEXCOORD_0") >= 0;
            Asset::Accessor uv;
            if (has_uv) uv = asset.accessor(attrs->index("TEXCOORD_0"), "uvs (but were promised)");
            if (nrm.count < pos.count || (has_uv && uv.count < pos.count)) fail("vertex attributes of different lengths");
            Mesh mesh;
            const long mi = prim.index("material");
            mesh.materialIndex = mi >= 0 ? static_cast<uint32_t>(mi) : static_cast<uint32_t>(model.materials_.size() - 1);
            if (mesh.materialIndex >= model.materials_.size()) fail("a primitive names a material that does not exist");
            mesh.vertices.resize(pos.count);
            for (size_t i = 0; i < pos.count; ++i) {
                vertex& v = mesh.vertices[i];
                element_as_floats(pos, i, &v.position.x, 3);
                element_as_floats(nrm, i, &v.normal.x, 3);
                if (has_uv) element_as_floats(uv, i, &v.uv.x, 2);
            }
            if (prim.index("indices") >= 0) {
                const Asset::Accessor ix = asset.accessor(prim.index("indices"), "indices");
                mesh.indices.resize(ix.count);
                for (size_t i = 0; i < ix.count; ++i) {
                    const uint8_t* p = ix.data + i * ix.stride;
                    uint32_t v;
                    if (ix.component == 5125) std::memcpy(&v, p, 4);
                    else if (ix.component == 5123) { uint16_t s; std::memcpy(&s, p, 2); v = s; }
                    else if (ix.component == 5121) v = *p;
                    else fail("index accessor with a bad component type");
                    if (v >= pos.count) fail("vertex index out of range");
                    mesh.indices[i] = v;
                }
            } else {  // GenerateMeshIndices
                mesh.indices.resize(pos.count);
                for (size_t i = 0; i < pos.count; ++i) mesh.indices[i] = static_cast<uint32_t>(i);
            }
            mesh.indices.resize(mesh.indices.size() - mesh.indices.size() % 3);
            if (attrs->index("TANGENT") >= 0) {
                const Asset::Accessor tg = asset.accessor(attrs->index("TANGENT"), "tangents");
                for (size_t i = 0; i < std::min(tg.count, pos.count); ++i) element_as_floats(tg, i, &mesh.vertices[i].tangent.x, 4);
            } else if (has_uv) {
                compute_tangents_mikk(mesh);  // compute_tangents.cc:58-62
            } else {
                compute_tangents_no_uv(mesh);
            }
            model.meshes_.push_back(std::move(mesh));
        }
        mesh_ids.push_back({count, static_cast<uint32_t>(n_prims)});
        count += static_cast<uint32_t>(n_prims);
    }

    // default scene, flattened (model_loader.cc:195-210)
    const Json& nodes = asset.list("nodes");
    const Json& scenes = asset.list("scenes");
    const long scene_index = std::max<long>(asset.root.index("scene"), 0);
    struct Item {
        size_t node;
        M4 parent;
        int depth;
    };
    std::vector<Item> todo;
    if (scenes.size()) {
        const Json* roots = scenes.at(static_cast<size_t>(scene_index)).find("nodes");
        if (roots)
            for (size_t i = roots->size(); i-- > 0;) todo.push_back({static_cast<size_t>(roots->arr[i].num), M4{}, 0});
    }
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (it.depth > 256) fail("node hierarchy too deep (a cycle?)");
        const Json& node = nodes.at(it.node);
        M4 world;
        if (const Json* mj = node.find("matrix"); mj && mj->size() == 16) {
            M4 local;
            for (int k = 0; k < 16; ++k) local.m[k] = static_cast<float>(mj->arr[k].num);
            world = mul(it.parent, local);
        } else {  // scale(rotate(translate(parent, t), r), s)
            float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
            if (const Json* j = node.find("translation"); j && j->size() == 3) for (int k = 0; k < 3; ++k) t[k] = static_cast<float>(j->arr[k].num);
            if (const Json* j = node.find("rotation"); j && j->size() == 4) for (int k = 0; k < 4; ++k) q[k] = static_cast<float>(j->arr[k].num);
            if (const Json* j = node.find("scale"); j && j->size() == 3) for (int k = 0; k < 3; ++k) s[k] = static_cast<float>(j->arr[k].num);
            M4 T;
            T.at(3, 0) = t[0]; T.at(3, 1) = t[1]; T.at(3, 2) = t[2];
            M4 R;
            const float x = q[0], y = q[1], z = q[2], w = q[3];
            R.at(0, 0) = 1 - 2 * (y * y + z * z); R.at(0, 1) = 2 * (x * y + w * z);     R.at(0, 2) = 2 * (x * z - w * y);
            R.at(1, 0) = 2 * (x * y - w * z);     R.at(1, 1) = 1 - 2 * (x * x + z * z); R.at(1, 2) = 2 * (y * z + w * x);
            R.at(2, 0) = 2 * (x * z + w * y);     R.at(2, 1) = 2 * (y * z - w * x);     R.at(2, 2) = 1 - 2 * (x * x + y * y);
            M4 S;
            S.at(0, 0) = s[0]; S.at(1, 1) = s[1]; S.at(2, 2) = s[2];
            world = mul(mul(mul(it.parent, T), R), S);
        }
        if (const long mesh = node.index("mesh"); mesh >= 0) {
            if (static_cast<size_t>(mesh) >= mesh_ids.size()) fail("a node names a mesh that does not exist");
            const M4 normal = transpose(inverse(world));
            for (uint32_t i = 0; i < mesh_ids[mesh].second; ++i) {
                Object o;
                o.ModelMatrix = to_fmat(world);
                o.NormalMatrix = to_fmat(normal);
                o.meshIndex = mesh_ids[mesh].first + i;
                model.objects_.push_back(o);
            }
        }
        if (const long cam = node.index("camera"); cam >= 0) {
            const Json& c = asset.list("cameras").at(static_cast<size_t>(cam));
            if (const Json* p = c.find("perspective")) {  // orthographic cameras are never snapped to (viewer.cc:69-71)
                Camera out;
                out.ModelMatrix = to_fmat(world);
                out.camera_params.yfov = static_cast<float>(p->number("yfov", 1.0));
                out.camera_params.znear = static_cast<float>(p->number("znear", 0.1));
                if (p->find("zfar")) out.camera_params.zfar = static_cast<float>(p->number("zfar", 1000.0));
                if (p->find("aspectRatio")) out.camera_params.aspectRatio = static_cast<float>(p->number("aspectRatio", 1.0));
                model.cameras_.push_back(out);
            }
        }
        if (const Json* ch = node.find("children"))
            for (size_t i = ch->size(); i-- > 0;) todo.push_back({static_cast<size_t>(ch->arr[i].num), world, it.depth + 1});
    }
    return model;
}

}  // namespace app
