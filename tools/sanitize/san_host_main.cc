// san_host — feeds files to the C++ host's readers (glTF / GLB / scene container through ModelLoader, PNG / JPEG through
// decode_image) in a build instrumented with AddressSanitizer and UndefinedBehaviorSanitizer.  The files are whatever
// the caller passes, malformed ones included: a reader may refuse a file, it may not touch memory it does not own.
// Prints one line per file; a sanitizer report ends the process with a non-zero status.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
int ptapp_convert_scene(const char* in_path, const char* out_path, char* err, size_t err_len);
int ptapp_decode_image(const unsigned char* bytes, size_t n, int* width, int* height, int* channels, unsigned char* rgba, size_t capacity,
                       char* err, size_t err_len);
}

int main(int argc, char** argv) {
    int accepted = 0, refused = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string p = argv[i];
        char err[256] = "";
        int rc;
        auto ends = [&](const char* s) { return p.size() >= std::strlen(s) && p.compare(p.size() - std::strlen(s), std::string::npos, s) == 0; };
        if (ends(".glb") || ends(".gltf") || ends(".ptsc")) {
            rc = ptapp_convert_scene(p.c_str(), (p + ".out").c_str(), err, sizeof(err));
        } else {
            std::vector<unsigned char> data;
            if (FILE* f = std::fopen(p.c_str(), "rb")) {
                unsigned char buf[65536];
                size_t n;
                while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) data.insert(data.end(), buf, buf + n);
                std::fclose(f);
            }
            int w = 0, h = 0, ch = 0;
            rc = ptapp_decode_image(data.data(), data.size(), &w, &h, &ch, nullptr, 0, err, sizeof(err));
            if (rc == 0 && w > 0 && h > 0 && static_cast<long long>(w) * h < (1 << 24)) {
                std::vector<unsigned char> px(static_cast<size_t>(w) * h * 4);
                rc = ptapp_decode_image(data.data(), data.size(), &w, &h, &ch, px.data(), px.size(), err, sizeof(err));
            }
        }
        (rc == 0 ? accepted : refused)++;
    }
    std::printf("accepted %d refused %d\n", accepted, refused);
    return 0;
}
