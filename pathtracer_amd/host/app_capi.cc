// app_capi.cc — plain-C entry points of libptapp.so for callers without a C++ toolchain (the Python tests):
// scene conversion through ModelLoader, image decoding, tangent generation.  None touches the GPU.
#include <cstdio>
#include <cstring>
#include <exception>

#include "app_model.h"
#include "image_decode.h"

extern "C" {

// Loads `in_path` with app::ModelLoader (glTF 2.0 or scene container) and writes it as a scene container.
// 0 on success; on failure a message is copied to err (if given) and -1 returned.
int ptapp_convert_scene(const char* in_path, const char* out_path, char* err, size_t err_len) {
    auto report = [&](const char* msg) {
        if (err && err_len) std::snprintf(err, err_len, "%s", msg);
        return -1;
    };
    try {
        app::ModelLoader loader;
        if (!loader.loadFromFile(in_path)) return report("ModelLoader::loadFromFile failed");
        if (!app::save_scene_container(loader.constructModel(), out_path)) return report("cannot write the scene container");
    } catch (const std::exception& e) {
        return report(e.what());
    }
    return 0;
}

// Decodes PNG / JPEG bytes.  Writes width, height and the file's channel count; when rgba is non-null and
// capacity >= width * height * 4 the pixels are copied.  0 on success, -1 on failure (message in err).
int ptapp_decode_image(const unsigned char* bytes, size_t n, int* width, int* height, int* channels, unsigned char* rgba,
                       size_t capacity, char* err, size_t err_len) {
    try {
        const app::DecodedImage d = app::decode_image(bytes, n);
        if (width) *width = d.width;
        if (height) *height = d.height;
        if (channels) *channels = d.channels;
        if (rgba && capacity >= d.rgba.size()) std::memcpy(rgba, d.rgba.data(), d.rgba.size());
    } catch (const std::exception& e) {
        if (err && err_len) std::snprintf(err, err_len, "%s", e.what());
        return -1;
    }
    return 0;
}

// MikkTSpace tangents (app::compute_tangents_mikk) of an indexed triangle mesh given as plain arrays: positions and
// normals 3 floats, uvs 2 floats per vertex; writes 4 floats per vertex (xyz, sign).  0 on success.
int ptapp_compute_tangents(const float* positions, const float* normals, const float* uvs, size_t n_vertices, const unsigned* indices,
                           size_t n_indices, float* tangents_out) {
    if (!positions || !normals || !uvs || !indices || !tangents_out || n_indices % 3 != 0) return -1;
    try {
        app::Mesh mesh;
        mesh.vertices.resize(n_vertices);
        for (size_t i = 0; i < n_vertices; ++i) {
            mesh.vertices[i].position = app::fvec3{positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]};
            mesh.vertices[i].normal = app::fvec3{normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
            mesh.vertices[i].uv = app::fvec2{uvs[2 * i], uvs[2 * i + 1]};
        }
        for (size_t k = 0; k < n_indices; ++k)
            if (indices[k] >= n_vertices) return -1;
        mesh.indices.assign(indices, indices + n_indices);
        app::compute_tangents_mikk(mesh);
        for (size_t i = 0; i < n_vertices; ++i) std::memcpy(tangents_out + 4 * i, &mesh.vertices[i].tangent.x, 16);
    } catch (const std::exception&) {
        return -1;
    }
    return 0;
}

}  // extern "C"
