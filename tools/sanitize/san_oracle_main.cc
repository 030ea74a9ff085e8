// san_oracle — drives the CPU parity oracle (oracle/pt_oracle.cc) through every program mode on a scene container,
// in a build instrumented with AddressSanitizer and UndefinedBehaviorSanitizer.  Test infrastructure: a sanitizer
// report makes the process exit non-zero; a clean run prints one checksum line per mode.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

#include "../../oracle/pt_oracle.h"

namespace {
struct Scene {
    std::vector<orc_vertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<orc_mesh> meshes;
    std::vector<orc_object> objects;
    std::vector<orc_material> materials;
    std::vector<orc_image> images;
    std::vector<uint8_t> texels;
    float camera[12] = {0, 0, 5, 0, 0, -1, 0, 1, 0, 0.8f, 0.1f, 100.0f};
};
bool load(const char* path, Scene& s) {
    std::ifstream f(path, std::ios::binary);
    char magic[8];
    uint64_t n[8];
    if (!f.read(magic, 8) || std::memcmp(magic, "PTSC0001", 8) != 0 || !f.read(reinterpret_cast<char*>(n), sizeof(n))) return false;
    s.vertices.resize(n[0]); s.indices.resize(n[1]); s.meshes.resize(n[2]); s.objects.resize(n[3]); s.materials.resize(n[4]);
    s.images.resize(n[5]); s.texels.resize(n[6] * 4);
    auto rd = [&](void* p, size_t bytes) { return bytes == 0 || static_cast<bool>(f.read(static_cast<char*>(p), static_cast<std::streamsize>(bytes))); };
    if (!rd(s.vertices.data(), n[0] * sizeof(orc_vertex)) || !rd(s.indices.data(), n[1] * 4) || !rd(s.meshes.data(), n[2] * sizeof(orc_mesh)) ||
        !rd(s.objects.data(), n[3] * 132) || !rd(s.materials.data(), n[4] * sizeof(orc_material)) || !rd(s.images.data(), n[5] * sizeof(orc_image)) ||
        !rd(s.texels.data(), n[6] * 4))
        return false;
    if (n[7]) return rd(s.camera, sizeof(s.camera));
    return true;
}
}  // namespace

int main(int argc, char** argv) {
    static_assert(sizeof(orc_object) == 132 && sizeof(orc_material) == 100 && sizeof(orc_image) == 16, "container record sizes");
    if (argc < 2) return 2;
    Scene sc;
    if (!load(argv[1], sc)) {
        std::fprintf(stderr, "cannot read %s\n", argv[1]);
        return 2;
    }
    const orc_model model{sc.vertices.data(), sc.vertices.size(), sc.indices.data(), sc.indices.size(), sc.meshes.data(),
                          static_cast<uint32_t>(sc.meshes.size()), sc.objects.data(), static_cast<uint32_t>(sc.objects.size()),
                          sc.materials.data(), static_cast<uint32_t>(sc.materials.size()), sc.images.data(),
                          static_cast<uint32_t>(sc.images.size()), sc.texels.data(), sc.texels.size() / 4};
    // a small synthetic environment: a vertical gradient with one bright texel
    const int ew = 16, eh = 8;
    std::vector<float> env(static_cast<size_t>(ew) * eh * 4, 1.0f);
    for (int y = 0; y < eh; ++y)
        for (int x = 0; x < ew; ++x) {
            float* p = &env[(static_cast<size_t>(y) * ew + x) * 4];
            p[0] = 0.2f + 0.1f * y; p[1] = 0.3f; p[2] = 1.0f - 0.1f * y;
        }
    env[4 * (3 * ew + 5)] = 40.0f;
    const orc_envmap envmap{ew, eh, env.data()};
    const int w = 48, h = 32;
    struct Mode { const char* name; int program, as; uint32_t spp, bounces, newrays, leaf, flags; int rng, math; };
    const Mode modes[] = {
        {"raycast", 0, 1, 1, 4, 0, 8, 0, 0, 1},  {"raycast-naive", 0, 0, 1, 4, 0, 8, 0, 0, 1}, {"ao", 1, 1, 4, 1, 4, 8, 0, 0, 1},
        {"pbr", 2, 1, 3, 4, 0, 8, 0, 0, 1},      {"pbr-libm-minstd", 2, 1, 2, 3, 0, 2, 0, 1, 0}, {"pbr-single", 2, 1, 4, 8, 0, 1, ORC_FLAG_SINGLE_PATH, 0, 1},
    };
    for (const Mode& m : modes) {
        orc_settings st{};
        st.samples_per_pixel = m.spp; st.max_ray_bounces = m.bounces; st.max_new_rays_per_bounce = m.newrays;
        st.max_triangles_per_bvh_leaf = m.leaf; st.envmap_rotation = 0.3f; st.program_mode = m.program; st.accel_struct_type = m.as;
        st.seed = 99; st.sample_begin = 0; st.sample_step = 1; st.flags = m.flags; st.tile_step = 1;
        orc_scene* s = orc_scene_create(&model, &envmap, &st);
        if (!s) return 3;
        orc_set_rng_mode(s, m.rng);
        orc_set_math_mode(s, m.math);
        orc_update_camera_transform_state(s, sc.camera, sc.camera + 3, sc.camera + 6, sc.camera[9], static_cast<float>(w) / h, sc.camera[10], sc.camera[11]);
        std::vector<float> img(static_cast<size_t>(w) * h * 4);
        orc_stats stats{};
        if (orc_render_frame(s, w, h, img.data(), 2, &stats) != 0) return 4;
        std::vector<orc_hit> hits(static_cast<size_t>(w) * h);
        if (orc_render_hits(s, w, h, 0, hits.data(), 2) != 0) return 5;
        orc_bvh_info info{};
        for (uint32_t k = 0; k < model.n_meshes; ++k) (void)orc_get_bvh_info(s, k, &info);
        double sum = 0.0;
        for (float v : img) sum += std::isfinite(v) ? v : 0.0;
        std::printf("%s rays %llu sum %.6f\n", m.name, static_cast<unsigned long long>(stats.rays), sum);
        orc_scene_destroy(s);
    }
    return 0;
}
