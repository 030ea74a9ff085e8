// san_bvh — runs the library's host tree builder (pathtracer_amd/csrc/pt_bvh.cc) in a build whose host code is
// instrumented with AddressSanitizer and UndefinedBehaviorSanitizer: seeded triangle soups of several sizes and
// shapes (compact, far from the origin, degenerate, all centroids equal), every leaf size, the reference's and the tuned splits, the single-leaf form and the
// instance table.  No GPU call is made.  Prints a checksum per case.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "pt_bvh.h"

int main() {
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    struct Case { const char* name; uint32_t n_tris; float offset, extent, tri_size; };
    const Case cases[] = {{"compact", 4000, 0.0f, 1.0f, 0.05f}, {"far", 3000, 5000.0f, 1.0f, 0.01f}, {"tiny", 5, 0.0f, 1.0f, 0.5f},
                          {"one", 1, 0.0f, 1.0f, 0.5f},         {"flat", 2500, 0.0f, 0.0f, 0.0f},    {"empty", 0, 0.0f, 1.0f, 0.1f}};
    for (const Case& c : cases) {
        std::vector<ptgpu_vertex> verts(3 * static_cast<size_t>(c.n_tris));
        std::vector<uint32_t> idx(verts.size());
        for (size_t t = 0; t < c.n_tris; ++t) {
            const float cx = c.offset + c.extent * u(rng), cy = c.offset + c.extent * u(rng), cz = c.offset + c.extent * u(rng);
            for (int k = 0; k < 3; ++k) {
                ptgpu_vertex& v = verts[3 * t + k];
                std::memset(&v, 0, sizeof(v));
                v.position[0] = cx + c.tri_size * u(rng);
                v.position[1] = cy + c.tri_size * u(rng);
                v.position[2] = cz + c.tri_size * u(rng);
                v.normal[2] = 1.0f;
                idx[3 * t + k] = static_cast<uint32_t>(3 * t + k);
            }
        }
        for (uint32_t leaf : {1u, 2u, 8u, 64u})
            for (int form = 0; form < 3; ++form) {  // the reference's splits, one leaf, the tuned splits
                pth::MeshTree tree;
                pth::build_mesh_tree(verts.data(), idx.data(), static_cast<uint32_t>(idx.size()), leaf, form == 1, form == 2, tree);
                pth::relocate(tree, 7u, 11u);
                std::printf("%s leaf %u form %d: nodes %zu tris %zu depth %u checksum %016llx\n", c.name, leaf, form, tree.nodes.size(),
                            tree.tris.size(), tree.info.max_depth, static_cast<unsigned long long>(tree.info.checksum));
            }
        if (c.n_tris) {  // an instance table over three placements of the mesh
            ptgpu_mesh mesh{0u, static_cast<uint32_t>(verts.size()), 0u, static_cast<uint32_t>(idx.size()), 0u};
            ptgpu_object objs[3];
            std::memset(objs, 0, sizeof(objs));
            for (int o = 0; o < 3; ++o) {
                for (int d = 0; d < 4; ++d) objs[o].model_matrix[5 * d] = objs[o].normal_matrix[5 * d] = 1.0f;
                objs[o].model_matrix[12] = 3.0f * static_cast<float>(o);
            }
            ptgpu_material mat{};
            ptgpu_model model{};
            model.vertices = verts.data(); model.n_vertices = verts.size(); model.indices = idx.data(); model.n_indices = idx.size();
            model.meshes = &mesh; model.n_meshes = 1; model.objects = objs; model.n_objects = 3; model.materials = &mat; model.n_materials = 1;
            pth::InstanceTable table;
            pth::build_instances(model, table);
            std::printf("%s instances %zu bound %.3f..%.3f\n", c.name, table.instances.size(), table.scene_min[0], table.scene_max[0]);
        }
    }
    return 0;
}
