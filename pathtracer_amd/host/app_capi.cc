// app_capi.cc — two plain-C entry points of libptapp.so for callers without a C++ toolchain (the Python tests):
// scene conversion through ModelLoader and image decoding.  Neither touches the GPU.
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

// Decodes PNG / baseline JPEG bytes.  Writes width, height and the file's channel count; when rgba is non-null and
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

}  // extern "C"
