// app_renderer.cc — Renderer / Viewer / loaders of the C++ host above the C ABI of libptgpu.
#include "app_renderer.h"

#include <thread>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>

namespace app {

// ---------------------------------------------------------------------------------------------
// textures and framebuffer (src/cpu_framebuffer.cc:85-148)
// ---------------------------------------------------------------------------------------------
namespace {

[[noreturn]] void load_failure(const std::filesystem::path& p, const char* why) {
    throw std::runtime_error("Unable to load image: " + p.string() + " (" + why + ")");
}

// Radiance RGBE reader with the semantics of stbi_loadf(path, ..., 4): alpha = 1, rgb = mantissa * 2^(e - 136)
void read_radiance(const std::filesystem::path& path, std::vector<hdr_pixel>& out, int& w, int& h) {
    std::ifstream f(path, std::ios::binary);
    if (!f) load_failure(path, "cannot open");
    std::vector<unsigned char> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t pos = 0;
    auto line = [&]() {
        std::string s;
        while (pos < data.size() && data[pos] != '\n') s.push_back(static_cast<char>(data[pos++]));
        ++pos;
        return s;
    };
    const std::string magic = line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE") load_failure(path, "not a Radiance picture");
    bool fmt_ok = false;
    for (;;) {
        const std::string s = line();
        if (s.empty()) break;
        if (s == "FORMAT=32-bit_rle_rgbe") fmt_ok = true;
        if (pos >= data.size()) break;
    }
    if (!fmt_ok) load_failure(path, "unsupported format");
    const std::string dims = line();
    if (std::sscanf(dims.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) load_failure(path, "unsupported layout");
    std::vector<unsigned char> rgbe(static_cast<size_t>(w) * h * 4);
    auto need = [&](size_t n) {
        if (pos + n > data.size()) load_failure(path, "truncated");
    };
    if (w < 8 || w >= 32768) {
        need(rgbe.size());
        std::memcpy(rgbe.data(), data.data() + pos, rgbe.size());
    } else {
        for (int y = 0; y < h; ++y) {
            need(4);
            if (!(data[pos] == 2 && data[pos + 1] == 2 && !(data[pos + 2] & 0x80))) {  // flat data from here on
                const size_t rest = static_cast<size_t>(h - y) * w * 4;
                need(rest);
                std::memcpy(rgbe.data() + static_cast<size_t>(y) * w * 4, data.data() + pos, rest);
                break;
            }
            if (((static_cast<int>(data[pos + 2]) << 8) | data[pos + 3]) != w) load_failure(path, "corrupt scanline");
            pos += 4;
            unsigned char* row = rgbe.data() + static_cast<size_t>(y) * w * 4;
            for (int ch = 0; ch < 4; ++ch) {
                int x = 0;
                while (x < w) {
                    need(2);
                    int count = data[pos++];
                    if (count > 128) {
                        count -= 128;
                        if (x + count > w) load_failure(path, "corrupt run");
                        const unsigned char v = data[pos++];
                        for (int k = 0; k < count; ++k) row[(x + k) * 4 + ch] = v;
                    } else {
                        if (count == 0 || x + count > w) load_failure(path, "corrupt run");
                        need(static_cast<size_t>(count));
                        for (int k = 0; k < count; ++k) row[(x + k) * 4 + ch] = data[pos++];
                    }
                    x += count;
                }
            }
        }
    }
    out.resize(static_cast<size_t>(w) * h);
    for (size_t i = 0; i < out.size(); ++i) {
        const unsigned char* p = rgbe.data() + i * 4;
        if (p[3] == 0) {
            out[i] = hdr_pixel{0.0f, 0.0f, 0.0f, 1.0f};
        } else {
            const float s = std::ldexp(1.0f, static_cast<int>(p[3]) - 136);
            out[i] = hdr_pixel{p[0] * s, p[1] * s, p[2] * s, 1.0f};
        }
    }
}

float linear_to_srgb(float c) {  // src/cpu_framebuffer.h:14-20
    if (c <= 0.0031308f) return 12.92f * c;
    return 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f;
}

uint32_t crc32_of(const unsigned char* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return ~crc;
}
void put_be32(std::vector<unsigned char>& v, uint32_t x) {
    for (int s = 24; s >= 0; s -= 8) v.push_back(static_cast<unsigned char>(x >> s));
}
void png_chunk(std::ofstream& f, const char tag[4], const std::vector<unsigned char>& payload) {
    std::vector<unsigned char> head;
    put_be32(head, static_cast<uint32_t>(payload.size()));
    f.write(reinterpret_cast<const char*>(head.data()), 4);
    std::vector<unsigned char> body(tag, tag + 4);
    body.insert(body.end(), payload.begin(), payload.end());
    f.write(reinterpret_cast<const char*>(body.data()), static_cast<std::streamsize>(body.size()));
    std::vector<unsigned char> tail;
    put_be32(tail, crc32_of(body.data(), body.size()));
    f.write(reinterpret_cast<const char*>(tail.data()), 4);
}
// 8-bit RGBA PNG; the zlib stream uses stored (uncompressed) blocks, which every decoder accepts
void write_png(const std::filesystem::path& path, const std::vector<unsigned char>& rgba8, int w, int h) {
    std::vector<unsigned char> raw;
    raw.reserve(static_cast<size_t>(h) * (1 + static_cast<size_t>(w) * 4));
    for (int y = 0; y < h; ++y) {
        raw.push_back(0);  // filter type 0
        raw.insert(raw.end(), rgba8.begin() + static_cast<size_t>(y) * w * 4, rgba8.begin() + static_cast<size_t>(y + 1) * w * 4);
    }
    std::vector<unsigned char> z{0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (unsigned char c : raw) {
        a = (a + c) % 65521u;
        b = (b + a) % 65521u;
    }
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back(static_cast<unsigned char>(n & 0xff));
        z.push_back(static_cast<unsigned char>(n >> 8));
        z.push_back(static_cast<unsigned char>(~n & 0xff));
        z.push_back(static_cast<unsigned char>((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        if (raw.empty()) break;
    }
    put_be32(z, (b << 16) | a);
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write " + path.string());
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    f.write(reinterpret_cast<const char*>(sig), 8);
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, static_cast<uint32_t>(w));
    put_be32(ihdr, static_cast<uint32_t>(h));
    const unsigned char rest[5] = {8, 6, 0, 0, 0};
    ihdr.insert(ihdr.end(), rest, rest + 5);
    png_chunk(f, "IHDR", ihdr);
    png_chunk(f, "IDAT", z);
    png_chunk(f, "IEND", {});
}
void write_radiance(const std::filesystem::path& path, const hdr_pixel* px, int w, int h) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write " + path.string());
    f << "#?RADIANCE\n# Written by pathtracer_amd\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y " << h << " +X " << w << "\n";
    std::vector<unsigned char> row(static_cast<size_t>(w) * 4);
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            const hdr_pixel& p = px[static_cast<size_t>(y) * w + x];
            const float m = std::max(p.x, std::max(p.y, p.z));
            unsigned char* o = row.data() + x * 4;
            if (!(m >= 1e-32f)) {
                o[0] = o[1] = o[2] = o[3] = 0;
            } else {
                int e;
                const float norm = std::frexp(m, &e) * 256.0f / m;
                auto q = [&](float v) { return static_cast<unsigned char>(std::clamp(v * norm, 0.0f, 255.0f)); };
                o[0] = q(p.x); o[1] = q(p.y); o[2] = q(p.z);
                o[3] = static_cast<unsigned char>(e + 128);
            }
        }
        f.write(reinterpret_cast<const char*>(row.data()), static_cast<std::streamsize>(row.size()));
    }
}

}  // namespace

template <>
CPUTexture<hdr_pixel>::CPUTexture(const std::filesystem::path& path) {
    read_radiance(path, data_, width_, height_);
}
template <>
CPUTexture<hdr_pixel> CPUTexture<hdr_pixel>::create_white_texture() { return CPUTexture(hdr_pixel{1.0f, 1.0f, 1.0f, 1.0f}); }
template <>
CPUTexture<hdr_pixel> CPUTexture<hdr_pixel>::create_black_texture() { return CPUTexture(hdr_pixel{0.0f, 0.0f, 0.0f, 1.0f}); }

CPUFrameBuffer::CPUFrameBuffer() : CPUFrameBuffer(1, 1) {}
CPUFrameBuffer::CPUFrameBuffer(int width, int height)
    : CPUTexture<hdr_pixel>(width, height, std::vector<hdr_pixel>(static_cast<size_t>(width) * height, hdr_pixel{0, 0, 0, 1})) {}
void CPUFrameBuffer::clear(const hdr_pixel clearColor) { std::fill(data_.begin(), data_.end(), clearColor); }
hdr_pixel& CPUFrameBuffer::at(int x, int y) { return data_[static_cast<size_t>(y) * width_ + x]; }
const hdr_pixel& CPUFrameBuffer::at(int x, int y) const { return data_[static_cast<size_t>(y) * width_ + x]; }
void CPUFrameBuffer::save_to_file(const std::filesystem::path& filePath) const {
    if (filePath.extension() == ".png") {
        std::vector<unsigned char> out(data_.size() * 4);
        for (size_t i = 0; i < data_.size(); ++i)
            for (int ch = 0; ch < 4; ++ch) {  // float_pixel_to_srgb8, src/cpu_framebuffer.h:52-59 (alpha goes through the curve too)
                const float c = std::clamp(data_[i][ch], 0.0f, 1.0f);
                out[i * 4 + ch] = static_cast<unsigned char>(linear_to_srgb(c) * 255 + 0.5f);
            }
        write_png(filePath, out, width_, height_);
    } else {
        write_radiance(filePath, data_.data(), width_, height_);
    }
}

// ---------------------------------------------------------------------------------------------
// scene container reader
// ---------------------------------------------------------------------------------------------
Model load_gltf_model(const std::filesystem::path& path);  // gltf_loader.cc

bool ModelLoader::loadFromFile(std::filesystem::path path) {
    if (!std::filesystem::exists(path)) {
        std::fprintf(stdout, "Failed to find \"%s\"\n", path.string().c_str());
        return false;
    }
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    char magic[8] = {};
    f.read(magic, 8);
    if (std::memcmp(magic, "PTSC0001", 8) != 0) {  // glTF 2.0: binary container or JSON text
        try {
            model_ = load_gltf_model(path);
        } catch (const std::exception& e) {
            std::fprintf(stderr, "%s\n", e.what());
            return false;
        }
        return true;
    }
    try {
        return load_scene_container(f, std::filesystem::file_size(path));
    } catch (const std::exception& e) {  // length_error / bad_alloc from a header that lies about its counts
        std::fprintf(stderr, "%s\n", e.what());
        return false;
    }
}

bool ModelLoader::load_scene_container(std::ifstream& f, uint64_t file_bytes) {
    uint64_t n[8];
    if (!f.read(reinterpret_cast<char*>(n), sizeof(n))) return false;
    const uint64_t nv = n[0], ni = n[1], nm = n[2], no = n[3], nmat = n[4], nimg = n[5], ntex = n[6], ncam = n[7];
    {   // every array must fit the file: checked count by count, so no product or sum can wrap
        const uint64_t elem[8] = {sizeof(ptgpu_vertex), 4, sizeof(ptgpu_mesh), sizeof(ptgpu_object), sizeof(ptgpu_material),
                                  sizeof(ptgpu_image), 4, 48};
        uint64_t left = file_bytes >= 72 ? file_bytes - 72 : 0;
        for (int k = 0; k < 8; ++k) {
            if (n[k] > left / elem[k]) return false;
            left -= n[k] * elem[k];
        }
    }
    std::vector<ptgpu_vertex> vertices(nv);
    std::vector<uint32_t> indices(ni);
    std::vector<ptgpu_mesh> meshes(nm);
    std::vector<ptgpu_object> objects(no);
    std::vector<ptgpu_material> materials(nmat);
    std::vector<ptgpu_image> images(nimg);
    std::vector<sdr_pixel> texels(ntex);
    auto rd = [&](void* p, size_t bytes) { return bytes == 0 || static_cast<bool>(f.read(static_cast<char*>(p), static_cast<std::streamsize>(bytes))); };
    static_assert(sizeof(ptgpu_mesh) == 20 && sizeof(ptgpu_object) == 132 && sizeof(ptgpu_material) == 100 && sizeof(ptgpu_image) == 16);
    if (!rd(vertices.data(), nv * sizeof(ptgpu_vertex)) || !rd(indices.data(), ni * 4) || !rd(meshes.data(), nm * sizeof(ptgpu_mesh)) ||
        !rd(objects.data(), no * sizeof(ptgpu_object)) || !rd(materials.data(), nmat * sizeof(ptgpu_material)) ||
        !rd(images.data(), nimg * sizeof(ptgpu_image)) || !rd(texels.data(), ntex * 4))
        return false;
    Model m;
    m.meshes_.resize(nm);
    for (uint64_t i = 0; i < nm; ++i) {
        const ptgpu_mesh& s = meshes[i];
        if (static_cast<uint64_t>(s.first_vertex) + s.n_vertices > nv || static_cast<uint64_t>(s.first_index) + s.n_indices > ni) return false;
        Mesh& d = m.meshes_[i];
        d.vertices.resize(s.n_vertices);
        static_assert(sizeof(vertex) == sizeof(ptgpu_vertex));
        std::memcpy(static_cast<void*>(d.vertices.data()), vertices.data() + s.first_vertex, s.n_vertices * sizeof(vertex));
        d.indices.assign(indices.begin() + s.first_index, indices.begin() + s.first_index + s.n_indices);
        d.materialIndex = s.material_index;
    }
    m.objects_.resize(no);
    for (uint64_t i = 0; i < no; ++i) {
        std::memcpy(&m.objects_[i].ModelMatrix, objects[i].model_matrix, 64);
        std::memcpy(&m.objects_[i].NormalMatrix, objects[i].normal_matrix, 64);
        m.objects_[i].meshIndex = objects[i].mesh_index;
    }
    m.materials_.resize(nmat);
    for (uint64_t i = 0; i < nmat; ++i) {
        const ptgpu_material& s = materials[i];
        Material& d = m.materials_[i];
        d.baseColorFactor = {s.base_color_factor[0], s.base_color_factor[1], s.base_color_factor[2], s.base_color_factor[3]};
        d.emissiveFactor = {s.emissive_factor[0], s.emissive_factor[1], s.emissive_factor[2]};
        d.attenuationFactor = {s.attenuation_factor[0], s.attenuation_factor[1], s.attenuation_factor[2]};
        d.metallicFactor = s.metallic_factor;
        d.roughnessFactor = s.roughness_factor;
        d.baseColorTextureIndex = s.base_color_texture;
        d.metallicRoughnessTextureIndex = s.metallic_roughness_texture;
        d.normalTextureIndex = s.normal_texture;
        d.ior = s.ior;
        d.dielectric_f0 = s.dielectric_f0;
        d.transmisionFactor = s.transmission_factor;
        d.transmissionTextureIndex = s.transmission_texture;
        d.emissiveTextureIndex = s.emissive_texture;
        d.emissiveStrength = s.emissive_strength;
        d.doubleSided = s.double_sided != 0;
        d.hasVolume = s.has_volume != 0;
        d.alphaBlending = s.alpha_blending != 0;
        d.alpha_cutoff = s.alpha_cutoff;
    }
    for (uint64_t i = 0; i < nimg; ++i) {
        const ptgpu_image& s = images[i];
        const uint64_t n_px = static_cast<uint64_t>(s.width > 0 ? s.width : 0) * static_cast<uint64_t>(s.height > 0 ? s.height : 0);
        if (s.width <= 0 || s.height <= 0 || s.texel_offset > ntex || n_px > ntex - s.texel_offset) return false;
        m.images_.emplace_back(s.width, s.height,
                               std::vector<sdr_pixel>(texels.begin() + s.texel_offset, texels.begin() + s.texel_offset + static_cast<uint64_t>(s.width) * s.height));
    }
    for (uint64_t i = 0; i < ncam; ++i) {
        float v[12];
        if (!rd(v, sizeof(v))) return false;
        Camera c;
        // the viewer reads position = column 3, direction = -column 2, up = column 1 (src/viewer.cc:61-63)
        c.ModelMatrix[1] = {v[6], v[7], v[8], 0.0f};
        c.ModelMatrix[2] = {-v[3], -v[4], -v[5], 0.0f};
        c.ModelMatrix[3] = {v[0], v[1], v[2], 1.0f};
        c.camera_params.yfov = v[9];
        c.camera_params.znear = v[10];
        c.camera_params.zfar = v[11];
        m.cameras_.push_back(c);
    }
    model_ = std::move(m);
    return true;
}

bool save_scene_container(const Model& m, const std::filesystem::path& path) {
    std::vector<ptgpu_vertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<ptgpu_mesh> meshes;
    for (const Mesh& s : m.meshes_) {
        meshes.push_back({static_cast<uint32_t>(vertices.size()), static_cast<uint32_t>(s.vertices.size()), static_cast<uint32_t>(indices.size()),
                          static_cast<uint32_t>(s.indices.size()), s.materialIndex});
        const size_t v0 = vertices.size();
        vertices.resize(v0 + s.vertices.size());
        std::memcpy(static_cast<void*>(vertices.data() + v0), s.vertices.data(), s.vertices.size() * sizeof(vertex));
        indices.insert(indices.end(), s.indices.begin(), s.indices.end());
    }
    std::vector<ptgpu_object> objects(m.objects_.size());
    for (size_t i = 0; i < objects.size(); ++i) {
        std::memcpy(objects[i].model_matrix, &m.objects_[i].ModelMatrix, 64);
        std::memcpy(objects[i].normal_matrix, &m.objects_[i].NormalMatrix, 64);
        objects[i].mesh_index = m.objects_[i].meshIndex;
    }
    std::vector<ptgpu_material> materials(m.materials_.size());
    for (size_t i = 0; i < materials.size(); ++i) {
        const Material& s = m.materials_[i];
        ptgpu_material& d = materials[i];
        std::memcpy(d.base_color_factor, &s.baseColorFactor, 16);
        std::memcpy(d.emissive_factor, &s.emissiveFactor, 12);
        std::memcpy(d.attenuation_factor, &s.attenuationFactor, 12);
        d.metallic_factor = s.metallicFactor; d.roughness_factor = s.roughnessFactor;
        d.base_color_texture = s.baseColorTextureIndex; d.metallic_roughness_texture = s.metallicRoughnessTextureIndex;
        d.normal_texture = s.normalTextureIndex; d.ior = s.ior; d.dielectric_f0 = s.dielectric_f0;
        d.transmission_factor = s.transmisionFactor; d.transmission_texture = s.transmissionTextureIndex;
        d.emissive_texture = s.emissiveTextureIndex; d.emissive_strength = s.emissiveStrength;
        d.double_sided = s.doubleSided; d.has_volume = s.hasVolume; d.alpha_blending = s.alphaBlending; d.alpha_cutoff = s.alpha_cutoff;
    }
    std::vector<ptgpu_image> images;
    std::vector<sdr_pixel> texels;
    for (const CPUTexture<sdr_pixel>& t : m.images_) {
        images.push_back({t.width(), t.height(), texels.size()});
        texels.insert(texels.end(), t.data(), t.data() + static_cast<size_t>(t.width()) * t.height());
    }
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    const uint64_t n[8] = {vertices.size(), indices.size(), meshes.size(), objects.size(), materials.size(), images.size(), texels.size(), m.cameras_.size()};
    f.write("PTSC0001", 8);
    f.write(reinterpret_cast<const char*>(n), sizeof(n));
    auto wr = [&](const void* p, size_t bytes) { if (bytes) f.write(static_cast<const char*>(p), static_cast<std::streamsize>(bytes)); };
    wr(vertices.data(), vertices.size() * sizeof(ptgpu_vertex));
    wr(indices.data(), indices.size() * 4);
    wr(meshes.data(), meshes.size() * sizeof(ptgpu_mesh));
    wr(objects.data(), objects.size() * sizeof(ptgpu_object));
    wr(materials.data(), materials.size() * sizeof(ptgpu_material));
    wr(images.data(), images.size() * sizeof(ptgpu_image));
    wr(texels.data(), texels.size() * 4);
    for (const Camera& c : m.cameras_) {  // position, direction, up as the viewer reads them (src/viewer.cc:61-63)
        const float v[12] = {c.ModelMatrix[3].x, c.ModelMatrix[3].y, c.ModelMatrix[3].z, -c.ModelMatrix[2].x, -c.ModelMatrix[2].y, -c.ModelMatrix[2].z,
                             c.ModelMatrix[1].x, c.ModelMatrix[1].y, c.ModelMatrix[1].z, c.camera_params.yfov, c.camera_params.znear,
                             c.camera_params.zfar.value_or(1000.0f)};
        wr(v, sizeof(v));
    }
    return static_cast<bool>(f);
}

// ---------------------------------------------------------------------------------------------
// Renderer
// ---------------------------------------------------------------------------------------------
void Renderer::fail(const char* what, int code) const {
    const char* msg = ptgpu_last_error(ctx_);
    throw std::runtime_error(std::string(what) + " failed (" + std::to_string(code) + "): " + (msg ? msg : ""));
}

Renderer::Renderer(int device) {
    const int rc = ptgpu_create(device, &ctx_);
    if (rc != PTGPU_OK) {
        ctx_ = nullptr;
        fail("ptgpu_create", rc);
    }
}
Renderer::Renderer(const std::vector<int>& devices, bool always_reduce) {
    if (devices.empty()) throw std::runtime_error("Renderer: no devices");
    for (int d : devices) {
        ptgpu_ctx* c = nullptr;
        const int rc = ptgpu_create(d, &c);
        if (rc != PTGPU_OK) {
            const std::string msg = ptgpu_last_error(nullptr);
            for (ptgpu_ctx* made : ctxs_) ptgpu_destroy(made);
            ctxs_.clear();
            throw std::runtime_error("ptgpu_create failed (" + std::to_string(rc) + ") on device " + std::to_string(d) + ": " + msg);
        }
        ctxs_.push_back(c);
    }
    ctx_ = ctxs_[0];
    if (devices.size() > 1 || always_reduce) {
        try {
            group_ = std::make_unique<DeviceGroup>(devices);
        } catch (...) {
            for (ptgpu_ctx* made : ctxs_) ptgpu_destroy(made);
            ctxs_.clear();
            ctx_ = nullptr;
            throw;
        }
    }
}
Renderer::~Renderer() {
    group_.reset();
    if (ctxs_.empty()) {
        if (ctx_) ptgpu_destroy(ctx_);
    } else {
        for (ptgpu_ctx* c : ctxs_) ptgpu_destroy(c);
    }
}

void Renderer::push_settings() {
    ptgpu_settings s{};
    s.samples_per_pixel = renderSettings_.samplesPerPixel;
    s.max_ray_bounces = renderSettings_.maxRayBounces;
    s.max_new_rays_per_bounce = renderSettings_.maxNewRaysPerBounce;
    s.max_triangles_per_bvh_leaf = renderSettings_.maxTrianglesPerBVHLeaf;
    s.envmap_rotation = renderSettings_.envmapRotation;
    s.program_mode = static_cast<int32_t>(renderSettings_.programMode);
    s.accel_struct_type = static_cast<int32_t>(renderSettings_.accelStructType);
    s.seed = seed_;
    s.sample_begin = 0;
    s.sample_step = 1;
    s.flags = (single_path_ ? PTGPU_FLAG_SINGLE_PATH : 0u) | (wide_nodes_ ? PTGPU_FLAG_WIDE_NODES : 0u);
    s.tile_begin = 0;
    s.tile_step = 1;
    const size_t n = ctxs_.empty() ? 1 : ctxs_.size();
    for (size_t k = 0; k < n; ++k) {
        s.sample_begin = group_ ? static_cast<uint32_t>(k) : 0u;   // device k of n: samples k, k + n, ...
        s.sample_step = group_ ? static_cast<uint32_t>(n) : 1u;
        const int rc = ptgpu_set_render_settings(ctxs_.empty() ? ctx_ : ctxs_[k], &s);
        if (rc != PTGPU_OK) fail("ptgpu_set_render_settings", rc);
    }
}
void Renderer::set_render_settings(const RenderSettings& settings) {  // src/renderer.cc:67-75
    renderSettings_ = settings;
    push_settings();
}
RenderSettings Renderer::get_render_settings() const { return renderSettings_; }

void Renderer::update_camera_transform_state(fvec3 position, fvec3 direction, fvec3 up, Perspective p) {  // src/renderer.cc:25-39
    const int rc = ptgpu_update_camera_transform_state(ctx_, &position.x, &direction.x, &up.x, p.yfov, p.aspectRatio.value_or(-1.0f),  // absent -> the reference defaults (src/renderer.cc:33-35)
                                                       p.znear, p.zfar.value_or(-1.0f));
    if (rc != PTGPU_OK) fail("ptgpu_update_camera_transform_state", rc);
    for (size_t k = 1; k < ctxs_.size(); ++k) {  // the other devices get the very same matrices
        float m[16], o[3];
        if (ptgpu_get_camera_raw(ctx_, m, o) != PTGPU_OK || ptgpu_set_camera_raw(ctxs_[k], m, o) != PTGPU_OK)
            throw std::runtime_error("camera state could not be copied to device context " + std::to_string(k));
    }
}

void Renderer::load_scene(const Model& model, const CPUTexture<hdr_pixel>& envmap) {  // src/renderer.cc:41-65
    std::vector<ptgpu_vertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<ptgpu_mesh> meshes(model.meshes_.size());
    for (size_t i = 0; i < model.meshes_.size(); ++i) {
        const Mesh& m = model.meshes_[i];
        meshes[i] = {static_cast<uint32_t>(vertices.size()), static_cast<uint32_t>(m.vertices.size()), static_cast<uint32_t>(indices.size()),
                     static_cast<uint32_t>(m.indices.size()), m.materialIndex};
        const size_t v0 = vertices.size();
        vertices.resize(v0 + m.vertices.size());
        std::memcpy(vertices.data() + v0, m.vertices.data(), m.vertices.size() * sizeof(vertex));
        indices.insert(indices.end(), m.indices.begin(), m.indices.end());
    }
    std::vector<ptgpu_object> objects(model.objects_.size());
    for (size_t i = 0; i < objects.size(); ++i) {
        std::memcpy(objects[i].model_matrix, &model.objects_[i].ModelMatrix, 64);
        std::memcpy(objects[i].normal_matrix, &model.objects_[i].NormalMatrix, 64);
        objects[i].mesh_index = model.objects_[i].meshIndex;
    }
    std::vector<ptgpu_material> materials(model.materials_.size());
    for (size_t i = 0; i < materials.size(); ++i) {
        const Material& s = model.materials_[i];
        ptgpu_material& d = materials[i];
        std::memcpy(d.base_color_factor, &s.baseColorFactor, 16);
        std::memcpy(d.emissive_factor, &s.emissiveFactor, 12);
        std::memcpy(d.attenuation_factor, &s.attenuationFactor, 12);
        d.metallic_factor = s.metallicFactor;
        d.roughness_factor = s.roughnessFactor;
        d.base_color_texture = s.baseColorTextureIndex;
        d.metallic_roughness_texture = s.metallicRoughnessTextureIndex;
        d.normal_texture = s.normalTextureIndex;
        d.ior = s.ior;
        d.dielectric_f0 = s.dielectric_f0;
        d.transmission_factor = s.transmisionFactor;
        d.transmission_texture = s.transmissionTextureIndex;
        d.emissive_texture = s.emissiveTextureIndex;
        d.emissive_strength = s.emissiveStrength;
        d.double_sided = s.doubleSided;
        d.has_volume = s.hasVolume;
        d.alpha_blending = s.alphaBlending;
        d.alpha_cutoff = s.alpha_cutoff;
    }
    std::vector<ptgpu_image> images(model.images_.size());
    std::vector<sdr_pixel> texels;
    for (size_t i = 0; i < images.size(); ++i) {
        const CPUTexture<sdr_pixel>& t = model.images_[i];
        images[i] = {t.width(), t.height(), texels.size()};
        texels.insert(texels.end(), t.data(), t.data() + static_cast<size_t>(t.width()) * t.height());
    }
    ptgpu_model m{};
    m.vertices = vertices.data(); m.n_vertices = vertices.size();
    m.indices = indices.data(); m.n_indices = indices.size();
    m.meshes = meshes.data(); m.n_meshes = static_cast<uint32_t>(meshes.size());
    m.objects = objects.data(); m.n_objects = static_cast<uint32_t>(objects.size());
    m.materials = materials.data(); m.n_materials = static_cast<uint32_t>(materials.size());
    m.images = images.data(); m.n_images = static_cast<uint32_t>(images.size());
    m.texels = reinterpret_cast<const uint8_t*>(texels.data()); m.n_texels = texels.size();
    ptgpu_envmap e{envmap.width(), envmap.height(), &envmap.data()->x};
    if (ctxs_.size() <= 1) {
        const int rc = ptgpu_load_scene(ctx_, &m, &e);
        if (rc != PTGPU_OK) fail("ptgpu_load_scene", rc);
        return;
    }
    // Every device holds the whole scene (it is read-only and small next to 288 GB), but the acceleration structure is built
    // ONCE, as the reference builds it once per load_scene (src/renderer.cc:41-64): the first context builds and uploads,
    // the others copy its device buffers (over xGMI), one host thread per device.
    const int rc0 = ptgpu_load_scene(ctxs_[0], &m, &e);
    if (rc0 != PTGPU_OK) fail("ptgpu_load_scene", rc0);
    std::vector<int> rcs(ctxs_.size(), PTGPU_OK);
    std::vector<std::thread> workers;
    for (size_t k = 1; k < ctxs_.size(); ++k) workers.emplace_back([&, k] { rcs[k] = ptgpu_load_scene_from(ctxs_[k], ctxs_[0]); });
    for (std::thread& t : workers) t.join();
    for (size_t k = 1; k < ctxs_.size(); ++k)
        if (rcs[k] != PTGPU_OK)
            throw std::runtime_error("ptgpu_load_scene_from failed (" + std::to_string(rcs[k]) + ") on device context " + std::to_string(k) +
                                     ": " + ptgpu_last_error(ctxs_[k]));
    push_settings();  // (a copied scene brings the first context's scene settings along; each device keeps its own sample subset)
}

void Renderer::render_frame(CPUFrameBuffer& framebuffer) {  // src/renderer.cc:77-123
    if (group_) return render_frame_on_group(framebuffer);
    const int rc = ptgpu_render_frame(ctx_, framebuffer.width(), framebuffer.height(), framebuffer.raw(), &stats_);
    if (rc != PTGPU_OK) fail("render_frame", rc);
}

// The frame of src/renderer.cc:77-123 on several GPUs: the (pixel, sample) pairs are independent, so device k renders the
// samples k, k + n, ... of every pixel into its own sum buffer; one RCCL reduce(sum) brings the partial sums to the first
// device, which divides by the sample count.  The sub-pixel table and the sampler keys are indexed by the global sample
// number, so the union over the devices is the sample set of a one-device frame.
void Renderer::render_frame_on_group(CPUFrameBuffer& framebuffer) {
    const int w = framebuffer.width(), h = framebuffer.height();
    const size_t n_floats = static_cast<size_t>(w) * h * 4;
    const size_t n = group_->size();
    std::vector<float*> bufs(n);
    for (size_t k = 0; k < n; ++k) bufs[k] = group_->sums(k, n_floats);
    std::vector<int> rcs(n, PTGPU_OK);
    std::vector<ptgpu_stats> stats(n);
    std::vector<std::thread> workers;
    for (size_t k = 0; k < n; ++k)
        workers.emplace_back([&, k] { rcs[k] = ptgpu_render_frame_sums(ctxs_[k], w, h, bufs[k], group_->stream(k), &stats[k]); });
    for (std::thread& t : workers) t.join();
    for (size_t k = 0; k < n; ++k)
        if (rcs[k] != PTGPU_OK)
            throw std::runtime_error("render_frame failed (" + std::to_string(rcs[k]) + ") on device context " + std::to_string(k) + ": " +
                                     ptgpu_last_error(ctxs_[k]));
    group_->reduce_to_first(n_floats);
    const uint32_t spp = renderSettings_.samplesPerPixel ? renderSettings_.samplesPerPixel : 1u;
    const int rc = ptgpu_resolve_sums(ctx_, w, h, bufs[0], spp, group_->stream(0));
    if (rc != PTGPU_OK) fail("ptgpu_resolve_sums", rc);
    group_->download_first(framebuffer.raw(), n_floats);
    stats_ = stats[0];
    for (size_t k = 1; k < n; ++k) {  // counts add up, times are those of the slowest device
        stats_.rays += stats[k].rays;
        stats_.camera_samples += stats[k].camera_samples;
        stats_.extend_launches += stats[k].extend_launches;
        stats_.shade_launches += stats[k].shade_launches;
        stats_.extend_ms = std::max(stats_.extend_ms, stats[k].extend_ms);
        stats_.shade_ms = std::max(stats_.shade_ms, stats[k].shade_ms);
        stats_.generate_ms = std::max(stats_.generate_ms, stats[k].generate_ms);
        stats_.total_ms = std::max(stats_.total_ms, stats[k].total_ms);
        stats_.traverse_ms = std::max(stats_.traverse_ms, stats[k].traverse_ms);
    }
}

BBox Renderer::get_scene_bound() const {
    BBox b;
    const int rc = ptgpu_get_scene_bound(ctx_, &b.min.x, &b.max.x);
    if (rc != PTGPU_OK) fail("ptgpu_get_scene_bound", rc);
    return b;
}

std::vector<ptgpu_hit> Renderer::render_hits(int width, int height, uint32_t sample) {
    std::vector<ptgpu_hit> out(static_cast<size_t>(width) * height);
    const int rc = ptgpu_render_hits(ctx_, width, height, sample, out.data());
    if (rc != PTGPU_OK) fail("ptgpu_render_hits", rc);
    return out;
}

// ---------------------------------------------------------------------------------------------
// Viewer (src/viewer.cc)
// ---------------------------------------------------------------------------------------------
Viewer::Viewer(Model&& model, CPUTexture<hdr_pixel>&& environmentTexture, const RenderSettings& settings, int device, bool wide_nodes)
    : model_(std::move(model)), environmentTexture_(std::move(environmentTexture)), renderer_(device) {
    start(settings, wide_nodes);
}
Viewer::Viewer(Model&& model, CPUTexture<hdr_pixel>&& environmentTexture, const RenderSettings& settings, const std::vector<int>& devices,
               bool always_reduce, bool wide_nodes)
    : model_(std::move(model)), environmentTexture_(std::move(environmentTexture)), renderer_(devices, always_reduce) {
    start(settings, wide_nodes);
}
void Viewer::start(const RenderSettings& settings, bool wide_nodes) {
    if (!model_.cameras_.empty()) activeCameraIndex_ = 0;
    framebuffer_ = CPUFrameBuffer(windowDimensions_.x, windowDimensions_.y);
    renderer_.set_wide_nodes(wide_nodes);
    renderer_.set_render_settings(settings);
    renderer_.load_scene(model_, environmentTexture_);
    if (!snap_to_camera()) {
        set_up_default_camera_transforms();
        renderer_.update_camera_transform_state(position_, direction_, up_, cam_params_);
    }
}
void Viewer::resize_window(const ivec2& newDimensions) {
    windowDimensions_ = newDimensions;
    framebuffer_ = CPUFrameBuffer(windowDimensions_.x, windowDimensions_.y);
}
void Viewer::set_active_camera(std::optional<uint32_t> cameraIndex) {
    if (cameraIndex.has_value()) {
        if (*cameraIndex >= model_.cameras_.size()) throw std::out_of_range("Camera index out of range in Viewer::set_active_camera");
        activeCameraIndex_ = cameraIndex;
        snap_to_camera();
    }
}
bool Viewer::snap_to_camera() {  // src/viewer.cc:54-79
    bool success = false;
    if (activeCameraIndex_.has_value()) {
        const Camera& camera = model_.cameras_[*activeCameraIndex_];
        position_ = xyz(camera.ModelMatrix[3]);
        const fvec3 z = xyz(camera.ModelMatrix[2]);
        direction_ = fvec3{-z.x, -z.y, -z.z};
        up_ = xyz(camera.ModelMatrix[1]);
        cam_params_ = camera.camera_params;
        success = true;
    }
    cam_params_.aspectRatio = static_cast<float>(windowDimensions_.x) / windowDimensions_.y;
    renderer_.update_camera_transform_state(position_, direction_, up_, cam_params_);
    return success;
}
void Viewer::set_render_settings(const RenderSettings& settings) {  // src/viewer.cc:81-88
    const RenderSettings current = renderer_.get_render_settings();
    renderer_.set_render_settings(settings);
    if (current != settings) renderer_.load_scene(model_, environmentTexture_);
}
void Viewer::set_up_default_camera_transforms() {  // src/viewer.cc:91-105
    direction_ = fvec3{0.0f, 0.0f, -1.0f};
    up_ = fvec3{0.0f, 1.0f, 0.0f};
    cam_params_ = Perspective{static_cast<float>(windowDimensions_.x) / windowDimensions_.y, 60.0f * 0.01745329251994329576923690768489f, 1000.0f, 0.1f};
    const BBox bounds = renderer_.get_scene_bound();
    const fvec3 dims{bounds.max.x - bounds.min.x, bounds.max.y - bounds.min.y, bounds.max.z - bounds.min.z};
    const float max_dim = std::max(dims.x, std::max(dims.y, dims.z));
    position_ = fvec3{bounds.min.x + dims.x * 0.5f, bounds.min.y + dims.y * 0.5f, bounds.min.z + dims.z * 0.5f + 1.0f * max_dim * 1.5f};
}

}  // namespace app
