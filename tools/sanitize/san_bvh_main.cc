// san_bvh — runs the library's host side of load_scene (pathtracer_amd/csrc/pt_bvh.cc and pt_scene_build.cc) in a build whose
// host code is instrumented with AddressSanitizer and UndefinedBehaviorSanitizer.  First the per-mesh tree builder: seeded
// triangle soups of several sizes and shapes (compact, far from the origin, degenerate, all centroids equal), every leaf size,
// the reference's and the tuned splits, the single-leaf form and the instance table, every tree checked (each triangle in
// exactly one leaf, leaf sizes, boxes nested).  Then the whole scene build with four-wide nodes — the optimal and the greedy
// collapse, the 8-bit quantisation, the numbering of the most visited nodes, the tree over the instance bounds — checked for:
// every triangle in one leaf of the wide tree, every child's decoded box containing its triangles after rounding, unused
// slots inverted and pointing at the dummy leaf, the instance tree holding every instance once; an object scaled to nothing
// left out of the instances; and models with not-a-number or overflowing coordinates refused instead of walked.  No GPU call is made.  Prints a line per case.
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <vector>

#include "pt_bvh.h"
#include "pt_kernels.h"
#include "pt_scene_build.h"

namespace {

uint32_t rng_state = 12345u;
float uniform() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return static_cast<float>(rng_state >> 8) * (1.0f / 16777216.0f);
}

void check(bool ok, const char* what) {
    if (!ok) {
        std::fprintf(stderr, "invariant violated: %s\n", what);
        std::exit(1);
    }
}

struct Soup {
    std::vector<ptgpu_vertex> vertices;
    std::vector<uint32_t> indices;
};
Soup make_soup(uint32_t n_tris, float offset) {
    Soup s;
    s.vertices.resize(static_cast<size_t>(n_tris) * 3);
    s.indices.resize(s.vertices.size());
    for (uint32_t i = 0; i < n_tris; ++i) {
        const float cx = uniform() * 10.0f + offset, cy = uniform() * 10.0f, cz = uniform() * 10.0f;
        const bool degenerate = (i % 17) == 0, duplicate = (i % 23) == 0 && i > 0;
        for (int k = 0; k < 3; ++k) {
            ptgpu_vertex& v = s.vertices[3 * i + k];
            std::memset(&v, 0, sizeof(v));
            if (duplicate) {
                v = s.vertices[3 * (i - 1) + k];
            } else {
                v.position[0] = cx + (degenerate ? 0.0f : uniform() * 0.3f);
                v.position[1] = cy + (degenerate ? 0.0f : uniform() * 0.3f);
                v.position[2] = cz + (degenerate ? 0.0f : uniform() * 0.3f);
            }
            s.indices[3 * i + k] = 3 * i + k;
        }
    }
    return s;
}

struct TestModel {
    std::vector<Soup> soups;
    std::vector<ptgpu_vertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<ptgpu_mesh> meshes;
    std::vector<ptgpu_object> objects;
    ptgpu_material material;
    ptgpu_model model;
};
void identity(float* m) {
    std::memset(m, 0, 64);
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}
// `n_meshes` soups, `n_objects` instances of them at scattered positions
void make_model(TestModel& t, const std::vector<uint32_t>& mesh_sizes, uint32_t n_objects, float offset) {
    t = TestModel{};
    for (uint32_t n : mesh_sizes) {
        const Soup s = make_soup(n, offset);
        ptgpu_mesh m{static_cast<uint32_t>(t.vertices.size()), static_cast<uint32_t>(s.vertices.size()), static_cast<uint32_t>(t.indices.size()),
                     static_cast<uint32_t>(s.indices.size()), 0u};
        t.vertices.insert(t.vertices.end(), s.vertices.begin(), s.vertices.end());
        t.indices.insert(t.indices.end(), s.indices.begin(), s.indices.end());
        t.meshes.push_back(m);
    }
    for (uint32_t i = 0; i < n_objects; ++i) {
        ptgpu_object o{};
        identity(o.model_matrix);
        identity(o.normal_matrix);
        const float sc = 0.5f + uniform();
        o.model_matrix[0] = o.model_matrix[5] = o.model_matrix[10] = sc;
        o.normal_matrix[0] = o.normal_matrix[5] = o.normal_matrix[10] = 1.0f / sc;
        o.model_matrix[12] = uniform() * 200.0f;
        o.model_matrix[13] = uniform() * 20.0f;
        o.model_matrix[14] = uniform() * 200.0f;
        o.mesh_index = (i + 4u) % static_cast<uint32_t>(mesh_sizes.size());  // (the first object gets the fifth mesh)
        t.objects.push_back(o);
    }
    std::memset(&t.material, 0, sizeof(t.material));
    t.material.base_color_texture = t.material.metallic_roughness_texture = t.material.normal_texture = -1;
    t.material.transmission_texture = t.material.emissive_texture = -1;
    t.model = ptgpu_model{};
    t.model.vertices = t.vertices.data();
    t.model.n_vertices = t.vertices.size();
    t.model.indices = t.indices.data();
    t.model.n_indices = t.indices.size();
    t.model.meshes = t.meshes.data();
    t.model.n_meshes = static_cast<uint32_t>(t.meshes.size());
    t.model.objects = t.objects.data();
    t.model.n_objects = static_cast<uint32_t>(t.objects.size());
    t.model.materials = &t.material;
    t.model.n_materials = 1;
}

// The four-wide trees of a built scene: every triangle of every instanced mesh in exactly one leaf; a child's decoded box
// (plane = origin + scale * q in binary32, as the kernel's exact path decodes it) contains every triangle below it;
// unused slots are inverted and name the dummy leaf; the first n_top4 nodes are reachable like the others.
uint64_t check_wide(const pth::BuiltScene& b) {
    const uint32_t n_tris = static_cast<uint32_t>(b.tris.size()) - 2u;
    uint32_t dummy = PT_LEAF_BIT | n_tris;
    if (!b.leaves64.empty()) {  // the dummy leaf is the record whose corners are all zero and that names triangle 0 once more
        dummy = PT_NO_NODE;
        for (const ptd::Node4& nd : b.nodes4)
            for (int k = 0; k < 4; ++k)
                if ((nd.ref[k] & PT_LEAF_BIT) && ((nd.qlo[0] >> (8 * k)) & 0xffu) == 255u && ((nd.qhi[0] >> (8 * k)) & 0xffu) == 0u) dummy = nd.ref[k];
    }
    std::vector<int> seen(n_tris, 0);
    std::vector<char> root_done(b.nodes4.size(), 0);
    uint64_t visited = 0;
    struct Item {
        uint32_t ref;
        float lo[3], hi[3];
    };
    for (const ptd::DevXform& xf : b.xforms) {
        if (xf.root4 & PT_LEAF_BIT) continue;
        check(xf.root4 < b.nodes4.size(), "wide root in range");
        if (root_done[xf.root4]) continue;  // another instance of the same mesh
        root_done[xf.root4] = 1;
        const float big = std::numeric_limits<float>::max();
        std::vector<Item> todo{Item{xf.root4, {-big, -big, -big}, {big, big, big}}};
        while (!todo.empty()) {
            const Item it = todo.back();
            todo.pop_back();
            if ((it.ref & PT_LEAF_BIT) && !b.leaves64.empty()) {
                // leaf records: every triangle of the leaf, rebuilt from the record's corners and selectors, equals a
                // 48-byte record of the same leaf (same p1, same edges, same index), in the same order
                uint32_t rec = it.ref & ~PT_LEAF_BIT;
                check(rec < b.leaves64.size(), "leaf record in range");
                for (;; ++rec) {
                    check(rec < b.leaves64.size(), "a continued leaf ends inside the record array");
                    const ptd::Leaf64& r = b.leaves64[rec];
                    const int n_in = (r.info & 1u) ? 2 : 1;
                    for (int which = 0; which < n_in; ++which) {
                        const uint32_t sel[3] = {which ? (r.info >> 8) & 3u : 0u, which ? (r.info >> 10) & 3u : 1u, which ? (r.info >> 12) & 3u : 2u};
                        const uint32_t index = which ? r.index1 : r.index0;
                        // find the 48-byte record with this index among the real triangles (indices are unique per mesh; the
                        // leaf's box check below does the rest)
                        bool found = false;
                        for (uint32_t t = 0; t < n_tris && !found; ++t) {
                            const ptd::TriQ& q = b.tris[t];
                            if (q.index != index) continue;
                            bool equal = true;
                            for (int a = 0; a < 3; ++a) {
                                const float p1 = r.v[sel[0]][a], e1 = r.v[sel[1]][a] - p1, e2 = r.v[sel[2]][a] - p1;
                                equal = equal && std::memcmp(&p1, &q.p1[a], 4) == 0 && std::memcmp(&e1, &q.e1[a], 4) == 0 && std::memcmp(&e2, &q.e2[a], 4) == 0;
                            }
                            if (!equal) continue;
                            found = true;
                            ++seen[t];
                        }
                        check(found, "a leaf record's triangle is one of the scene's triangles, bit for bit");
                        for (int k = 0; k < 3; ++k)
                            for (int a = 0; a < 3; ++a) check(r.v[sel[k]][a] >= it.lo[a] && r.v[sel[k]][a] <= it.hi[a], "corner inside its leaf's decoded box");
                    }
                    if (!(r.info & 2u)) break;
                }
                continue;
            }
            if (it.ref & PT_LEAF_BIT) {
                const uint32_t first = it.ref & ~PT_LEAF_BIT;
                check(first < n_tris, "wide leaf starts at a real triangle");
                const uint32_t count = b.tris[first].leaf_count;
                check(count >= 1 && count <= 2 && first + count <= n_tris, "wide leaf of one or two triangles");
                for (uint32_t k = 0; k < count; ++k) {
                    const ptd::TriQ& t = b.tris[first + k];
                    ++seen[first + k];
                    for (int a = 0; a < 3; ++a) {
                        const float p[3] = {t.p1[a], t.p1[a] + t.e1[a], t.p1[a] + t.e2[a]};
                        for (float v : p) {
                            // (p2 and p3 are recomputed from the stored edges here: allow that rounding)
                            const float slack = 4.0f * std::numeric_limits<float>::epsilon() * (std::fabs(v) + std::fabs(t.p1[a]));
                            check(v >= it.lo[a] - slack && v <= it.hi[a] + slack, "triangle inside its leaf's decoded box");
                        }
                    }
                }
                continue;
            }
            check(it.ref < b.nodes4.size(), "wide node reference in range");
            ++visited;
            const ptd::Node4& nd = b.nodes4[it.ref];
            int n_real = 0;
            for (int k = 0; k < 4; ++k) {
                const bool unused = nd.ref[k] == dummy;
                Item ch;
                ch.ref = nd.ref[k];
                for (int a = 0; a < 3; ++a) {
                    const uint32_t ql = (nd.qlo[a] >> (8 * k)) & 0xffu, qh = (nd.qhi[a] >> (8 * k)) & 0xffu;
                    if (unused) check(ql == 255u && qh == 0u, "unused slot is an inverted box");
                    else check(ql <= qh, "a child's low plane is not above its high plane");
                    ch.lo[a] = nd.origin[a] + nd.scale[a] * static_cast<float>(ql);
                    ch.hi[a] = nd.origin[a] + nd.scale[a] * static_cast<float>(qh);
                }
                if (unused) continue;
                check(n_real == k, "real children come before unused slots");
                ++n_real;
                todo.push_back(ch);
            }
            check(n_real >= 2, "a wide node has at least two children");
        }
    }
    for (const ptd::DevXform& xf : b.xforms)  // meshes that are a single leaf
        if (xf.root4 & PT_LEAF_BIT) {
            const uint32_t first = xf.root4 & ~PT_LEAF_BIT;
            for (uint32_t k = 0; k < b.tris[first].leaf_count; ++k) seen[first + k] = 1;
        }
    for (uint32_t i = 0; i < n_tris; ++i) check(seen[i] <= 1, "no triangle in two wide leaves");
    return visited;
}

void check_instance_tree(const pth::BuiltScene& b) {
    if (b.tlas.empty()) return;
    std::vector<int> seen(b.instances.size(), 0);
    const uint32_t n = static_cast<uint32_t>(b.tlas.size());
    check(n == 2u * b.instances.size() - 1u, "a binary tree over the instances");
    for (uint32_t i = 0; i < n; ++i) {  // following every "enter" link visits all nodes in order
        const ptd::TlasNode& nd = b.tlas[i];
        check(nd.skip > i && nd.skip <= n, "skip link points forward, at most one past the end");
        if (nd.inst != PT_NO_NODE) {
            check(nd.inst < b.instances.size(), "instance in range");
            ++seen[nd.inst];
            check(nd.skip == i + 1u, "a leaf's skip link is the next node");
            for (int a = 0; a < 3; ++a)
                check(nd.lo[a] == b.instances[nd.inst].dop[2 * a] && nd.hi[a] == b.instances[nd.inst].dop[2 * a + 1], "leaf box = first three slabs");
        }
    }
    for (int s : seen) check(s == 1, "every instance in exactly one leaf of the instance tree");
}

}  // namespace

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
    // ---- (2) the whole scene build, four-wide nodes ----
    uint64_t wide_nodes = 0;
    for (int leaf_records : {-1, 1})
    for (bool greedy : {false, true})
        for (bool reference_splits : {false, true})
            for (float offset : {0.0f, 5000.0f})
                for (uint32_t n_objects : {1u, 5u, 130u}) {  // 130 > kTlasLinearMax: with the tree over the instance bounds
                    if (leaf_records > 0 && (greedy || n_objects == 130u)) continue;
                    TestModel tm;
                    if (leaf_records > 0) make_model(tm, {1u, 2u, 3u, 5u, 300u, 1500u}, n_objects, offset);
                    else make_model(tm, {1u, 2u, 3u, 5u, 300u, 20000u}, n_objects, offset);
                    std::string err;
                    check(pth::validate_model(tm.model, &err), "a finite model validates");
                    pth::BuildOptions bo;
                    bo.want_wide = true;
                    bo.greedy_collapse = greedy;
                    bo.reference_splits = reference_splits;
                    bo.build_threads = 2;
                    bo.leaf_records = leaf_records;
                    pth::BuiltScene built;
                    int code = 0;
                    if (!pth::build_scene(tm.model, bo, built, &err, &code)) {
                        std::fprintf(stderr, "build_scene failed: %s\n", err.c_str());
                        return 1;
                    }
                    check(!built.nodes4.empty() && built.tris.size() >= 2, "four-wide nodes were made");
                    check((leaf_records > 0) == !built.leaves64.empty(), "leaf records exactly when asked for");
                    check(built.n_top4 <= static_cast<uint32_t>(ptd::kTopNodes4) && built.n_top4 <= built.nodes4.size(), "top-node count within the LDS table");
                    check(built.n_top <= static_cast<uint32_t>(ptd::kTopNodes), "binary top-node count within the LDS table");
                    check((n_objects > ptd::kTlasLinearMax) == !built.tlas.empty(), "instance tree exactly when there are many instances");
                    wide_nodes += check_wide(built);
                    check_instance_tree(built);
                    check(built.max_depth >= 3u * built.max_depth4 + 1u, "stack bound covers three pushes per wide level");
                }
    // ---- bad coordinates are refused, and the collapse does not walk outside its tables even if they are not ----
    {
        TestModel tm;
        make_model(tm, {300u}, 1, 0.0f);
        std::string err;
        tm.vertices[17].position[1] = std::numeric_limits<float>::quiet_NaN();
        check(!pth::validate_model(tm.model, &err) && err.find("not finite") != std::string::npos, "a NaN position is refused");
        tm.vertices[17].position[1] = 1.0e20f;
        check(!pth::validate_model(tm.model, &err), "a position beyond 1e18 is refused");
        tm.vertices[17].position[1] = std::numeric_limits<float>::infinity();
        check(!pth::validate_model(tm.model, &err), "an infinite position is refused");
        tm.vertices[17].position[1] = 0.5f;
        tm.objects[0].model_matrix[13] = std::numeric_limits<float>::quiet_NaN();
        check(!pth::validate_model(tm.model, &err), "a NaN matrix entry is refused");
        tm.objects[0].model_matrix[13] = 0.0f;
        check(pth::validate_model(tm.model, &err), "the repaired model validates");
        // an object whose model matrix cannot be inverted arrives with a normal matrix that is not finite: the model is
        // accepted, the object is left out of the instances and the rest of the scene is built as usual
        for (float bad : {std::numeric_limits<float>::quiet_NaN(), std::numeric_limits<float>::infinity()})
            for (uint32_t victim : {0u, 3u, 6u}) {
                TestModel hm;
                make_model(hm, {40u, 90u, 20u, 64u, 150u}, 7, 0.0f);
                hm.objects[victim].model_matrix[0] = 0.0f;
                for (int k : {0, 5, 10}) hm.objects[victim].normal_matrix[k] = bad;
                check(pth::validate_model(hm.model, &err), "an object scaled to nothing is accepted");
                pth::BuildOptions bo;
                bo.want_wide = true;
                pth::BuiltScene built;
                int code = 0;
                check(pth::build_scene(hm.model, bo, built, &err, &code), "the scene around it builds");
                check(built.instances.size() == 6u, "it is left out of the instances");
                for (const auto& in : built.instances) check(in.object_index != victim && in.object_index < 7u, "the others keep their object numbers");
                wide_nodes += check_wide(built);
                check_instance_tree(built);
            }
        // the builders themselves on what validation would have refused: they must terminate inside their arrays and either
        // produce a structurally sound result or say why not
        for (float bad : {std::numeric_limits<float>::quiet_NaN(), 1.0e20f, -1.0e30f, std::numeric_limits<float>::infinity()}) {
            for (uint32_t victim : {3u, 450u, 899u}) {
                TestModel bm;
                make_model(bm, {300u}, 1, 0.0f);
                bm.vertices[victim].position[victim % 3] = bad;
                for (bool greedy : {false, true}) {
                    pth::BuildOptions bo;
                    bo.want_wide = true;
                    bo.greedy_collapse = greedy;
                    pth::BuiltScene built;
                    int code = 0;
                    const bool ok = pth::build_scene(bm.model, bo, built, &err, &code);
                    if (ok) {
                        // structure only: the boxes are meaningless with such coordinates
                        for (const ptd::Node4& nd : built.nodes4)
                            for (int k = 0; k < 4; ++k)
                                check((nd.ref[k] & PT_LEAF_BIT) ? (nd.ref[k] & ~PT_LEAF_BIT) < built.tris.size() : nd.ref[k] < built.nodes4.size(),
                                      "references stay inside the arrays even for unusable coordinates");
                    } else {
                        check(!err.empty(), "a refused build says why");
                    }
                }
            }
        }
    }
    std::printf("scene builds: wide nodes walked %llu, builds %llu\n", static_cast<unsigned long long>(wide_nodes),
                static_cast<unsigned long long>(pth::host_build_count()));
    return 0;
}
