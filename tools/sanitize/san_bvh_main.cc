// san_bvh — runs the library's host tree builder (pathtracer_amd/csrc/pt_bvh.cc) in a build whose host code is
// instrumented with AddressSanitizer and UndefinedBehaviorSanitizer: seeded triangle soups of several sizes and
// shapes (compact, far from the origin, degenerate, all centroids equal), every leaf size, the reference's and the tuned splits, the single-leaf form and the
// instance table, and checks every tree (each triangle in exactly one leaf, leaf sizes, boxes nested).  No GPU call is
// made.  Prints a checksum per case.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "pt_bvh.h"

// Every triangle in exactly one leaf, no leaf above the limit (unless the whole mesh is one leaf), every child box inside
// nothing smaller than the triangles below it.  Returns an error text or nullptr.
static const char* check_tree(const pth::MeshTree& t, uint32_t n_tris, uint32_t max_leaf, bool single) {
    if (n_tris == 0) return nullptr;
    std::vector<int> seen(n_tris, 0);
    struct Item { uint32_t ref; float lo[3], hi[3]; bool boxed; };
    std::vector<Item> todo;
    Item root{};
    root.ref = t.root;
    root.boxed = false;
    todo.push_back(root);
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (it.ref & PT_LEAF_BIT) {
            const uint32_t first = it.ref & ~PT_LEAF_BIT;
            if (first >= t.tris.size()) return "leaf reference out of range";
            const uint32_t count = t.tris[first].leaf_count;
            if (count == 0 || first + count > t.tris.size()) return "bad leaf count";
            if (!single && n_tris > max_leaf && count > max_leaf) return "leaf above the limit";
            for (uint32_t k = 0; k < count; ++k) {
                const ptd::DevTri& tr = t.tris[first + k];
                if (tr.index % 3 || tr.index / 3 >= n_tris) return "bad triangle index";
                ++seen[tr.index / 3];
                if (it.boxed)
                    for (const float* p : {tr.p1, tr.p2, tr.p3})
                        for (int a = 0; a < 3; ++a)
                            if (p[a] < it.lo[a] || p[a] > it.hi[a]) return "triangle outside its parent's box";
            }
            continue;
        }
        if (it.ref >= t.nodes.size()) return "node reference out of range";
        const ptd::DevNode& nd = t.nodes[it.ref];
        Item l{}, r{};
        l.ref = nd.left; r.ref = nd.right;
        l.boxed = r.boxed = true;
        std::memcpy(l.lo, nd.l_lo, 12); std::memcpy(l.hi, nd.l_hi, 12);
        std::memcpy(r.lo, nd.r_lo, 12); std::memcpy(r.hi, nd.r_hi, 12);
        if (it.boxed)
            for (int a = 0; a < 3; ++a)
                if (l.lo[a] < it.lo[a] || r.lo[a] < it.lo[a] || l.hi[a] > it.hi[a] || r.hi[a] > it.hi[a]) return "child box outside its parent's";
        todo.push_back(r);
        todo.push_back(l);
    }
    for (int v : seen)
        if (v != 1) return "a triangle is missing or in two leaves";
    return nullptr;
}

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
                if (const char* bad = check_tree(tree, c.n_tris, leaf, form == 1)) {
                    std::printf("%s leaf %u form %d: %s\n", c.name, leaf, form, bad);
                    return 1;
                }
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
