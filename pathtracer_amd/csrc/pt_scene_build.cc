// pt_scene_build.cc — host side of Renderer::load_scene (see pt_scene_build.h).
#include "pt_scene_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <queue>

#include "pt_kernels.h"

namespace pth {
namespace {

constexpr uint32_t kWideLeaf = 2;  // largest leaf of a tree built for four-wide nodes
// Coordinates beyond this make products of two box extents overflow binary32 (areas, costs): such a model is refused.
constexpr float kMaxCoordinate = 1.0e18f;

std::atomic<uint64_t> g_builds{0};

bool fail(std::string* err, int* code, int c, const char* fmt, ...) __attribute__((format(printf, 4, 5)));
bool fail(std::string* err, int* code, int c, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (err) *err = buf;
    if (code) *code = c;
    return false;
}

float srgb_to_linear_host(float ch) {  // cpu_framebuffer.h:21-27
    if (ch <= 0.04045f) return ch / 12.92f;
    return std::pow((ch + 0.055f) / 1.055f, 2.4f);
}

inline bool usable(float v) { return std::fabs(v) <= kMaxCoordinate; }  // false for NaN and infinities too

struct Child {
    uint32_t ref;
    float lo[3], hi[3];
};
inline float half_area(const Child& ch) {
    const float dx = ch.hi[0] - ch.lo[0], dy = ch.hi[1] - ch.lo[1], dz = ch.hi[2] - ch.lo[2];
    return dx * dy + dx * dz + dy * dz;
}
inline void children_of(const std::vector<ptd::DevNode>& nodes, uint32_t node, Child out[2]) {
    const ptd::DevNode& nd = nodes[node];
    out[0].ref = nd.left;
    out[1].ref = nd.right;
    std::memcpy(out[0].lo, nd.l_lo, 12);
    std::memcpy(out[0].hi, nd.l_hi, 12);
    std::memcpy(out[1].lo, nd.r_lo, 12);
    std::memcpy(out[1].hi, nd.r_hi, 12);
}

struct TopEntry {
    float priority;
    uint32_t node, mesh;
    bool operator<(const TopEntry& o) const { return priority < o.priority; }
};

}  // namespace

uint64_t host_build_count() { return g_builds.load(); }

bool validate_model(const ptgpu_model& model, std::string* err) {
    int code = 0;
    if ((model.n_vertices && !model.vertices) || (model.n_indices && !model.indices) || (model.n_meshes && !model.meshes) ||
        (model.n_objects && !model.objects) || (model.n_materials && !model.materials) || (model.n_images && !model.images) ||
        (model.n_texels && !model.texels))
        return fail(err, &code, PTGPU_E_INVALID, "model: a non-empty array has a null pointer");
    for (uint32_t i = 0; i < model.n_meshes; ++i) {
        const ptgpu_mesh& m = model.meshes[i];
        if (m.n_indices % 3 != 0) return fail(err, &code, PTGPU_E_INVALID, "mesh %u: index count %u is not a multiple of 3", i, m.n_indices);
        if (static_cast<uint64_t>(m.first_vertex) + m.n_vertices > model.n_vertices ||
            static_cast<uint64_t>(m.first_index) + m.n_indices > model.n_indices)
            return fail(err, &code, PTGPU_E_INVALID, "mesh %u: vertex/index range outside the model arrays", i);
        if (m.material_index >= model.n_materials) return fail(err, &code, PTGPU_E_INVALID, "mesh %u: material index out of range", i);
        for (uint32_t k = 0; k < m.n_indices; ++k)
            if (model.indices[m.first_index + k] >= m.n_vertices) return fail(err, &code, PTGPU_E_INVALID, "mesh %u: index %u out of range", i, k);
        // Not-a-number or huge positions would poison every box above them: comparisons with NaN are all false, areas
        // overflow to infinity, and the builders' decisions (split choice, four-wide collapse) stop being consistent.
        for (uint32_t v = 0; v < m.n_vertices; ++v) {
            const float* p = model.vertices[m.first_vertex + v].position;
            if (!usable(p[0]) || !usable(p[1]) || !usable(p[2]))
                return fail(err, &code, PTGPU_E_INVALID, "mesh %u: vertex %u has a position that is not finite or beyond 1e18", i, v);
        }
    }
    for (uint32_t i = 0; i < model.n_objects; ++i) {
        const ptgpu_object& o = model.objects[i];
        if (o.mesh_index >= model.n_meshes) return fail(err, &code, PTGPU_E_INVALID, "object %u: mesh index out of range", i);
        // (a normal matrix that is not finite is the mark of a model matrix without an inverse — an object scaled to nothing
        // — and is let through: build_instances leaves such an object out, which is how the reference's picture shows it)
        for (int k = 0; k < 16; ++k)
            if (!usable(o.model_matrix[k]))
                return fail(err, &code, PTGPU_E_INVALID, "object %u: an entry of the model matrix is not finite or beyond 1e18", i);
    }
    if (model.n_objects >= (1u << 30)) return fail(err, &code, PTGPU_E_INVALID, "too many objects");
    for (uint32_t i = 0; i < model.n_materials; ++i) {
        const ptgpu_material& m = model.materials[i];
        const int32_t tex[5] = {m.base_color_texture, m.metallic_roughness_texture, m.normal_texture, m.transmission_texture,
                                m.emissive_texture};
        for (int32_t t : tex)
            if (t >= static_cast<int32_t>(model.n_images)) return fail(err, &code, PTGPU_E_INVALID, "material %u: texture index out of range", i);
    }
    for (uint32_t i = 0; i < model.n_images; ++i) {
        const ptgpu_image& im = model.images[i];
        if (im.width <= 0 || im.height <= 0 || im.texel_offset + static_cast<uint64_t>(im.width) * im.height > model.n_texels)
            return fail(err, &code, PTGPU_E_INVALID, "image %u: bad size or texel range", i);
    }
    if (model.n_texels >= (1ull << 32)) return fail(err, &code, PTGPU_E_INVALID, "texel pool too large");
    return true;
}

// The most visited nodes go first in the node array: the traversal kernel keeps nodes[0 .. n_top) in LDS, which
// takes their fetches off the vector L1's address unit (the busiest unit of that kernel).  "Most visited" is
// estimated the way the surface area heuristic does: a node is visited in proportion to the area of its box,
// summed over the instances of its mesh (scaled by the instance's size).  Only the numbering changes.
uint32_t number_top_nodes(std::vector<ptd::DevNode>& nodes, std::vector<uint32_t>& mesh_root, const std::vector<float>& mesh_weight,
                          std::vector<ptd::DevXform>& xforms, uint32_t capacity) {
    if (nodes.empty()) return 0;
    auto area = [](const float* lo, const float* hi) {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dx * dz + dy * dz;
    };
    std::priority_queue<TopEntry> queue;
    for (uint32_t i = 0; i < mesh_root.size(); ++i)
        if (!(mesh_root[i] & PT_LEAF_BIT) && mesh_weight[i] > 0.0f) queue.push({std::numeric_limits<float>::max(), mesh_root[i], i});
    std::vector<uint32_t> order;  // old indices of the chosen nodes
    while (!queue.empty() && order.size() < capacity) {
        const TopEntry e = queue.top();
        queue.pop();
        order.push_back(e.node);
        const ptd::DevNode& nd = nodes[e.node];
        if (!(nd.left & PT_LEAF_BIT)) queue.push({area(nd.l_lo, nd.l_hi) * mesh_weight[e.mesh], nd.left, e.mesh});
        if (!(nd.right & PT_LEAF_BIT)) queue.push({area(nd.r_lo, nd.r_hi) * mesh_weight[e.mesh], nd.right, e.mesh});
    }
    const uint32_t n_top = static_cast<uint32_t>(order.size());
    std::vector<uint32_t> new_index(nodes.size(), PT_NO_NODE);
    for (uint32_t k = 0; k < n_top; ++k) new_index[order[k]] = k;
    uint32_t next = n_top;
    for (uint32_t i = 0; i < nodes.size(); ++i)
        if (new_index[i] == PT_NO_NODE) new_index[i] = next++;
    std::vector<ptd::DevNode> moved(nodes.size());
    for (uint32_t i = 0; i < nodes.size(); ++i) {
        ptd::DevNode n = nodes[i];
        if (!(n.left & PT_LEAF_BIT)) n.left = new_index[n.left];
        if (!(n.right & PT_LEAF_BIT)) n.right = new_index[n.right];
        moved[new_index[i]] = n;
    }
    nodes.swap(moved);
    for (uint32_t& r : mesh_root)
        if (!(r & PT_LEAF_BIT)) r = new_index[r];
    for (ptd::DevXform& xf : xforms)
        if (!(xf.root & PT_LEAF_BIT)) xf.root = new_index[xf.root];
    return n_top;
}

// Balanced by count (median split along the widest axis of the box centres), so the depth is ceil(log2(n)).
// Depth-first emission with an explicit work list: a range becomes a node, then its left half, then its right half;
// `skip` = the index after the node's subtree = node + 2 * count - 1.
void build_instance_tree(const std::vector<ptd::DevInstance>& instances, std::vector<ptd::TlasNode>& tlas) {
    tlas.clear();
    if (instances.empty()) return;
    std::vector<uint32_t> order(instances.size());
    for (uint32_t k = 0; k < order.size(); ++k) order[k] = k;
    struct Range {
        uint32_t begin, end;
    };
    std::vector<Range> todo{{0u, static_cast<uint32_t>(order.size())}};
    while (!todo.empty()) {
        const Range r = todo.back();
        todo.pop_back();
        const uint32_t count = r.end - r.begin;
        ptd::TlasNode nd;
        float clo[3], chi[3];
        for (int a = 0; a < 3; ++a) {
            nd.lo[a] = clo[a] = std::numeric_limits<float>::max();
            nd.hi[a] = chi[a] = std::numeric_limits<float>::lowest();
        }
        for (uint32_t i = r.begin; i < r.end; ++i) {
            const float* d = instances[order[i]].dop;
            for (int a = 0; a < 3; ++a) {
                nd.lo[a] = std::min(nd.lo[a], d[2 * a]);
                nd.hi[a] = std::max(nd.hi[a], d[2 * a + 1]);
                const float c = d[2 * a] + d[2 * a + 1];
                clo[a] = std::min(clo[a], c);
                chi[a] = std::max(chi[a], c);
            }
        }
        nd.skip = static_cast<uint32_t>(tlas.size()) + 2u * count - 1u;
        nd.inst = (count == 1) ? order[r.begin] : PT_NO_NODE;
        tlas.push_back(nd);
        if (count == 1) continue;
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
        const uint32_t mid = r.begin + count / 2;
        std::nth_element(order.begin() + r.begin, order.begin() + mid, order.begin() + r.end, [&](uint32_t x, uint32_t y) {
            const float* dx = instances[x].dop;
            const float* dy = instances[y].dop;
            const float cx = dx[2 * axis] + dx[2 * axis + 1], cy = dy[2 * axis] + dy[2 * axis + 1];
            return cx < cy || (cx == cy && x < y);
        });
        todo.push_back({mid, r.end});  // popped after the whole left half
        todo.push_back({r.begin, mid});
    }
}

// Which binary nodes survive as four-wide nodes.  A ray pays one visit per surviving node whose box it enters, so
// the expected number of visits is the summed surface area of the surviving nodes; that sum is minimised exactly by
// dynamic programming over the binary tree (Ylitie, Karras, Laine: Efficient incoherent ray traversal on GPUs
// through compressed wide BVHs, 2017, section 4.1): cost[n][k] = least area below n when n's subtree is
// represented by at most k+1 roots (k = 0: n itself is a wide node), part[n][k] = how many of them its left child gets
// (0 = the solution for one root fewer is as good).  `greedy` selects the older rule instead (always pull up the
// child with the largest box).
bool collapse_to_wide(const std::vector<ptd::DevNode>& nodes, const std::vector<uint32_t>& binary_root, uint32_t dummy_leaf, bool greedy,
                      std::vector<ptd::Node4>& wide, std::vector<uint32_t>& mesh_root4, uint32_t* max_depth4_out, std::string* err) {
    int code = 0;
    wide.clear();
    mesh_root4.assign(binary_root.size(), PT_NO_NODE);
    std::vector<uint32_t> depth_of;  // per wide node
    uint32_t max_depth4 = 0;
    struct Todo {
        uint32_t binary, wide_index;
    };
    std::vector<Todo> todo;
    auto make = [&](uint32_t binary, uint32_t depth) -> uint32_t {
        const uint32_t w = static_cast<uint32_t>(wide.size());
        wide.emplace_back();
        depth_of.push_back(depth);
        max_depth4 = std::max(max_depth4, depth);
        todo.push_back({binary, w});
        return w;
    };
    std::vector<float> cost;
    std::vector<uint8_t> part;
    auto cost_of = [&](uint32_t ref, int roots) { return (ref & PT_LEAF_BIT) ? 0.0f : cost[static_cast<size_t>(ref) * 4 + (roots - 1)]; };
    if (!greedy) {
        cost.assign(nodes.size() * 4, 0.0f);
        part.assign(nodes.size() * 4, 0);
        std::vector<uint32_t> order;  // every node after its children: reverse of a pre-order walk
        for (uint32_t root : binary_root) {
            if (root == PT_NO_NODE || (root & PT_LEAF_BIT)) continue;
            const size_t first = order.size();
            order.push_back(root);
            for (size_t i = first; i < order.size(); ++i) {
                const ptd::DevNode& nd = nodes[order[i]];
                if (!(nd.left & PT_LEAF_BIT)) order.push_back(nd.left);
                if (!(nd.right & PT_LEAF_BIT)) order.push_back(nd.right);
            }
        }
        for (size_t i = order.size(); i-- > 0;) {
            const uint32_t n = order[i];
            const ptd::DevNode& nd = nodes[n];
            float lo[3], hi[3];
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(nd.l_lo[a], nd.r_lo[a]);
                hi[a] = std::max(nd.l_hi[a], nd.r_hi[a]);
            }
            const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
            float own = dx * dy + dx * dz + dy * dz;
            // (validate_model keeps coordinates finite and small enough for this product; a value that is not a usable
            // number all the same is counted as "infinitely expensive", which every comparison below orders correctly)
            if (!(own >= 0.0f) || !(own < std::numeric_limits<float>::infinity())) own = std::numeric_limits<float>::max();
            float below[5] = {0, 0, 0, 0, 0};  // below[k]: children represented by at most k roots, k = 2..4
            uint8_t left_share[5] = {0, 1, 1, 1, 1};  // (k roots can always be split 1 : k - 1)
            for (int k = 2; k <= 4; ++k) {
                below[k] = std::numeric_limits<float>::infinity();
                for (int l = 1; l < k; ++l) {
                    const float c = cost_of(nd.left, l) + cost_of(nd.right, k - l);
                    if (c < below[k] || l == 1) {
                        below[k] = c;
                        left_share[k] = static_cast<uint8_t>(l);
                    }
                }
            }
            float* c = &cost[static_cast<size_t>(n) * 4];
            uint8_t* p4 = &part[static_cast<size_t>(n) * 4];
            c[0] = own + below[4];  // one root: n is a wide node, its (at most four) children come from below[4]
            p4[0] = left_share[4];
            for (int k = 2; k <= 4; ++k) {  // k roots: n dissolves into its children's roots, unless k - 1 roots do as well
                if (c[k - 2] <= below[k]) {
                    c[k - 1] = c[k - 2];
                    p4[k - 1] = 0;
                } else {
                    c[k - 1] = below[k];
                    p4[k - 1] = left_share[k];
                }
            }
        }
    }
    for (uint32_t m = 0; m < binary_root.size(); ++m) {
        const uint32_t root = binary_root[m];
        if (root == PT_NO_NODE) continue;  // mesh without an instance
        if (root & PT_LEAF_BIT) {
            mesh_root4[m] = root;
            continue;
        }
        mesh_root4[m] = make(root, 1);
        while (!todo.empty()) {
            const Todo t = todo.back();
            todo.pop_back();
            Child ch[4];
            int n_ch = 2;
            children_of(nodes, t.binary, ch);
            if (!greedy) {
                // follow the recorded decisions: `roots` of the subtree below a child become children of this wide node
                n_ch = 0;
                struct Open {
                    Child c;
                    int roots;
                };
                Open open[8];
                int n_open = 0;
                Child two[2];
                children_of(nodes, t.binary, two);
                const int l = part[static_cast<size_t>(t.binary) * 4];
                if (l < 1 || l > 3) return fail(err, &code, PTGPU_E_INVALID, "four-wide collapse: node %u records a split of %d", t.binary, l);
                open[n_open++] = {two[1], 4 - l};
                open[n_open++] = {two[0], l};
                while (n_open) {
                    const Open o = open[--n_open];
                    if (o.roots < 1 || o.roots > 3) return fail(err, &code, PTGPU_E_INVALID, "four-wide collapse: %d roots for a child of node %u", o.roots, t.binary);
                    if ((o.c.ref & PT_LEAF_BIT) || o.roots == 1) {
                        if (n_ch >= 4) return fail(err, &code, PTGPU_E_INVALID, "four-wide collapse: more than four children below node %u", t.binary);
                        ch[n_ch++] = o.c;
                        continue;
                    }
                    const int share = part[static_cast<size_t>(o.c.ref) * 4 + (o.roots - 1)];
                    if (share == 0) {
                        open[n_open++] = {o.c, o.roots - 1};
                        continue;
                    }
                    if (share >= o.roots || n_open + 2 > 8)
                        return fail(err, &code, PTGPU_E_INVALID, "four-wide collapse: inconsistent split %d of %d below node %u", share, o.roots, t.binary);
                    children_of(nodes, o.c.ref, two);
                    open[n_open++] = {two[1], o.roots - share};
                    open[n_open++] = {two[0], share};
                }
                if (n_ch < 2) return fail(err, &code, PTGPU_E_INVALID, "four-wide collapse: node %u kept %d children", t.binary, n_ch);
            }
            while (greedy && n_ch < 4) {  // pull up the inner child with the largest box
                int best = -1;
                float best_area = -1.0f;
                for (int k = 0; k < n_ch; ++k)
                    if (!(ch[k].ref & PT_LEAF_BIT) && (best < 0 || half_area(ch[k]) > best_area)) {
                        best_area = half_area(ch[k]);
                        best = k;
                    }
                if (best < 0) break;
                Child two[2];
                children_of(nodes, ch[best].ref, two);
                ch[best] = two[0];
                ch[n_ch++] = two[1];
            }
            ptd::Node4 nd;
            std::memset(&nd, 0, sizeof(nd));
            // the node's own box, and per axis the step that reaches its far side in 255 steps (plane = origin + step * q,
            // evaluated in binary32 as the kernel's slow path does; low planes round down, high planes up)
            for (int a = 0; a < 3; ++a) {
                float lo = ch[0].lo[a], hi = ch[0].hi[a];
                for (int k = 1; k < n_ch; ++k) {
                    lo = std::min(lo, ch[k].lo[a]);
                    hi = std::max(hi, ch[k].hi[a]);
                }
                float step = (hi - lo) / 255.0f;
                if (!(step > 0.0f) || !(step < std::numeric_limits<float>::max())) step = 0.0f;
                // (far from the origin one unit in the last place of `hi` is many of `step`: grow geometrically)
                for (float grow = 1.0f / 1024.0f; step > 0.0f && lo + step * 255.0f < hi; grow *= 2.0f)
                    step = std::max(std::nextafter(step, std::numeric_limits<float>::max()), step * (1.0f + grow));
                nd.origin[a] = lo;
                nd.scale[a] = step;
                // (clamped before the conversion: with coordinates that validation refuses the quotient can be infinite or
                // not a number, which no int holds)
                const auto to_step = [](float x) { return !(x > 0.0f) ? 0 : (x >= 255.0f ? 255 : static_cast<int>(x)); };
                for (int k = 0; k < n_ch; ++k) {
                    int ql = 0, qh = 0;
                    if (step > 0.0f) {
                        ql = to_step(std::floor((ch[k].lo[a] - lo) / step));
                        while (ql > 0 && lo + step * static_cast<float>(ql) > ch[k].lo[a]) --ql;
                        qh = to_step(std::ceil((ch[k].hi[a] - lo) / step));
                        while (qh < 255 && lo + step * static_cast<float>(qh) < ch[k].hi[a]) ++qh;
                    }
                    nd.qlo[a] |= static_cast<uint32_t>(ql) << (8 * k);
                    nd.qhi[a] |= static_cast<uint32_t>(qh) << (8 * k);
                }
            }
            // An unused slot is not marked "absent": it names the padding record after the last triangle (one degenerate
            // triangle, never hit) as a leaf, inside an inverted box (low planes 255, high planes 0).  A ray with a finite
            // inverse direction never enters such a box (its near plane lies behind its far plane on every axis), except in
            // a node of zero extent, where the degenerate triangle is tested and missed; the kernel's path for rays with
            // a zero direction component, whose box test is symmetric in the two planes, skips inverted slots explicitly.
            // The fast path so needs no test for a slot's existence.
            for (int k = n_ch; k < 4; ++k)
                for (int a = 0; a < 3; ++a) {
                    nd.qlo[a] |= 255u << (8 * k);
                    nd.qhi[a] &= ~(255u << (8 * k));
                }
            for (int k = 0; k < 4; ++k)
                nd.ref[k] = (k >= n_ch) ? dummy_leaf : ((ch[k].ref & PT_LEAF_BIT) ? ch[k].ref : make(ch[k].ref, depth_of[t.wide_index] + 1));
            wide[t.wide_index] = nd;
        }
    }
    if (max_depth4_out) *max_depth4_out = max_depth4;
    return true;
}

void make_leaf_records(const std::vector<ptd::DevTri>& tris, uint32_t n_real, std::vector<ptd::Node4>& wide, std::vector<uint32_t>& mesh_root4,
                       std::vector<ptd::Leaf64>& out) {
    out.clear();
    std::vector<uint32_t> record_of(static_cast<size_t>(n_real) + 1u, PT_NO_NODE);  // first triangle of a leaf -> its first record
    auto same = [](const float* a, const float* b) { return std::memcmp(a, b, 12) == 0; };
    auto emit_leaf = [&](uint32_t first) -> uint32_t {
        if (record_of[first] != PT_NO_NODE) return record_of[first];
        const uint32_t at = static_cast<uint32_t>(out.size());
        record_of[first] = at;
        if (first >= n_real) {  // the dummy leaf of unused child slots: one degenerate triangle (all corners in one point, never hit)
            ptd::Leaf64 r;
            std::memset(&r, 0, sizeof(r));
            out.push_back(r);
            return at;
        }
        const uint32_t count = tris[first].leaf_count ? tris[first].leaf_count : 1u;
        for (uint32_t i = 0; i < count;) {
            const ptd::DevTri& a = tris[first + i];
            ptd::Leaf64 r;
            std::memset(&r, 0, sizeof(r));
            std::memcpy(r.v[0], a.p1, 12);
            std::memcpy(r.v[1], a.p2, 12);
            std::memcpy(r.v[2], a.p3, 12);
            r.index0 = a.index;
            uint32_t used = 1;
            if (i + 1 < count) {  // the next triangle rides along when at most one of its corners is new
                const ptd::DevTri& b = tris[first + i + 1];
                const float* corner[3] = {b.p1, b.p2, b.p3};
                uint32_t sel[3];
                bool have_new = false, fits = true;
                for (int k = 0; k < 3; ++k) {
                    sel[k] = 3u;
                    for (uint32_t j = 0; j < 3; ++j)
                        if (same(corner[k], r.v[j])) {
                            sel[k] = j;
                            break;
                        }
                    if (sel[k] != 3u) continue;
                    if (!have_new) {
                        std::memcpy(r.v[3], corner[k], 12);
                        have_new = true;
                    } else if (!same(corner[k], r.v[3])) {
                        fits = false;  // a second new corner: no room in the record
                    }
                }
                if (fits) {
                    r.index1 = b.index;
                    r.info |= 1u | (sel[0] << 8) | (sel[1] << 10) | (sel[2] << 12);
                    used = 2;
                } else {
                    std::memset(r.v[3], 0, 12);
                }
            }
            i += used;
            if (i < count) r.info |= 2u;
            out.push_back(r);
        }
        return at;
    };
    auto move = [&](uint32_t ref) -> uint32_t {
        if (ref == PT_NO_NODE || !(ref & PT_LEAF_BIT)) return ref;
        return PT_LEAF_BIT | emit_leaf(ref & ~PT_LEAF_BIT);
    };
    // in the order a traversal meets them: the leaves of one node next to each other
    for (ptd::Node4& nd : wide)
        for (int k = 0; k < 4; ++k) nd.ref[k] = move(nd.ref[k]);
    for (uint32_t& r : mesh_root4) r = move(r);
}

uint32_t number_top_nodes4(std::vector<ptd::Node4>& wide, std::vector<uint32_t>& mesh_root4, const std::vector<float>& mesh_weight,
                           uint32_t capacity) {
    std::priority_queue<TopEntry> queue;
    for (uint32_t m = 0; m < mesh_root4.size(); ++m)
        if (mesh_root4[m] != PT_NO_NODE && !(mesh_root4[m] & PT_LEAF_BIT) && mesh_weight[m] > 0.0f)
            queue.push({std::numeric_limits<float>::max(), mesh_root4[m], m});
    std::vector<uint32_t> order;
    while (!queue.empty() && order.size() < capacity) {
        const TopEntry e = queue.top();
        queue.pop();
        order.push_back(e.node);
        const ptd::Node4& nd = wide[e.node];
        for (int k = 0; k < 4; ++k)
            if (nd.ref[k] != PT_NO_NODE && !(nd.ref[k] & PT_LEAF_BIT)) {
                float ext[3];
                for (int a = 0; a < 3; ++a)
                    ext[a] = nd.scale[a] * static_cast<float>(static_cast<int>((nd.qhi[a] >> (8 * k)) & 0xffu) -
                                                              static_cast<int>((nd.qlo[a] >> (8 * k)) & 0xffu));
                queue.push({(ext[0] * ext[1] + ext[0] * ext[2] + ext[1] * ext[2]) * mesh_weight[e.mesh], nd.ref[k], e.mesh});
            }
    }
    const uint32_t n_top4 = static_cast<uint32_t>(order.size());
    std::vector<uint32_t> new_index(wide.size(), PT_NO_NODE);
    for (uint32_t k = 0; k < n_top4; ++k) new_index[order[k]] = k;
    uint32_t next = n_top4;
    for (uint32_t i = 0; i < wide.size(); ++i)
        if (new_index[i] == PT_NO_NODE) new_index[i] = next++;
    std::vector<ptd::Node4> moved(wide.size());
    for (uint32_t i = 0; i < wide.size(); ++i) {
        ptd::Node4 n = wide[i];
        for (int k = 0; k < 4; ++k)
            if (n.ref[k] != PT_NO_NODE && !(n.ref[k] & PT_LEAF_BIT)) n.ref[k] = new_index[n.ref[k]];
        moved[new_index[i]] = n;
    }
    wide.swap(moved);
    for (uint32_t& r : mesh_root4)
        if (r != PT_NO_NODE && !(r & PT_LEAF_BIT)) r = new_index[r];
    return n_top4;
}

bool build_scene(const ptgpu_model& model, const BuildOptions& opt, BuiltScene& out, std::string* err, int* code) {
    out = BuiltScene{};
    const bool naive = opt.naive;
    // Four-wide nodes are for throughput, not for the reference's visiting order: their trees get small leaves (fewer
    // triangle tests per ray, a few more node visits; measured best at 2), whatever maxTrianglesPerBVHLeaf says ...
    const bool want_wide = opt.want_wide && !naive;
    const uint32_t leaf_setting = opt.max_leaf ? opt.max_leaf : 1;
    const uint32_t max_leaf = want_wide ? std::min(leaf_setting, kWideLeaf) : leaf_setting;
    // ... and are collapsed from a tree of lower expected cost than the reference's (2 % fewer node visits per ray)
    const bool tuned_splits = want_wide && !opt.reference_splits;

    // ---- acceleration structure (BVH_AS::BVH_AS, acceleration_structure.cc:352-390) ----
    InstanceTable table;
    build_instances(model, table);
    std::vector<ptd::DevNode> all_nodes;
    std::vector<ptd::DevTri> all_tris;
    std::vector<uint32_t> mesh_root(model.n_meshes);
    out.bvh_info.assign(model.n_meshes, ptgpu_bvh_info{});
    for (uint32_t i = 0; i < model.n_meshes; ++i) {
        const ptgpu_mesh& m = model.meshes[i];
        MeshTree tree;
        if (opt.mesh_builder) {
            std::string e;
            if (!opt.mesh_builder(model.vertices + m.first_vertex, model.indices + m.first_index, m.n_indices, max_leaf, tree, &e))
                return fail(err, code, PTGPU_E_HIP, "mesh %u: %s", i, e.c_str());
        } else {
            build_mesh_tree(model.vertices + m.first_vertex, model.indices + m.first_index, m.n_indices, max_leaf, naive, tuned_splits, tree,
                            opt.build_threads);
        }
        if (all_tris.size() + tree.tris.size() >= (1ull << 31) - 2 || all_nodes.size() + tree.nodes.size() >= (1ull << 31))
            return fail(err, code, PTGPU_E_INVALID, "scene exceeds 2^31 triangles or nodes");
        relocate(tree, static_cast<uint32_t>(all_nodes.size()), static_cast<uint32_t>(all_tris.size()));
        mesh_root[i] = tree.root;
        out.bvh_info[i] = tree.info;
        out.max_depth = std::max(out.max_depth, tree.info.max_depth);
        all_nodes.insert(all_nodes.end(), tree.nodes.begin(), tree.nodes.end());
        all_tris.insert(all_tris.end(), tree.tris.begin(), tree.tris.end());
    }
    for (ptd::DevXform& xf : table.xforms) {
        xf.root = mesh_root[xf.mesh_index];
        const ptgpu_material& mat = model.materials[model.meshes[xf.mesh_index].material_index];
        const bool double_sided = mat.double_sided || mat.has_volume;  // :377-378
        xf.cull = double_sided ? 0u : 1u;
    }
    std::memcpy(out.scene_min, table.scene_min, 12);
    std::memcpy(out.scene_max, table.scene_max, 12);

    std::vector<float> mesh_weight(model.n_meshes, 0.0f);  // summed squared scale of a mesh's instances
    for (const ptd::DevXform& xf : table.xforms) {
        const float* m = xf.model;  // columns c0 c1 c2 (c3 = translation)
        const float det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[3] * (m[1] * m[8] - m[2] * m[7]) + m[6] * (m[1] * m[5] - m[2] * m[4]);
        const float scale = std::cbrt(std::fabs(det));
        mesh_weight[xf.mesh_index] += scale * scale;
    }
    out.n_top = number_top_nodes(all_nodes, mesh_root, mesh_weight, table.xforms, static_cast<uint32_t>(ptd::kTopNodes));

    // ---- a tree over the instances' boxes when there are many of them (the instance pass then culls before it tests) ----
    if (table.instances.size() > ptd::kTlasLinearMax && !opt.linear_tlas) build_instance_tree(table.instances, out.tlas);

    // ---- optional four-wide nodes: inner children pulled up into their parent ----
    if (want_wide && !all_nodes.empty()) {
        std::vector<uint32_t> binary_root(model.n_meshes, PT_NO_NODE);
        for (const ptd::DevXform& xf : table.xforms) binary_root[xf.mesh_index] = xf.root;
        std::vector<uint32_t> mesh_root4;
        const uint32_t dummy_leaf = PT_LEAF_BIT | static_cast<uint32_t>(all_tris.size());
        std::string e;
        if (!collapse_to_wide(all_nodes, binary_root, dummy_leaf, opt.greedy_collapse, out.nodes4, mesh_root4, &out.max_depth4, &e))
            return fail(err, code, PTGPU_E_INVALID, "%s", e.c_str());
        // Which form of the four-wide kernel: seven workgroups per CU when nodes and triangles fit the Infinity Cache with room
        // to spare (the kernel is then limited by the latency of a step), six otherwise (limited by the gather path: more
        // wavefronts only add to it).
        const uint64_t geometry_bytes = static_cast<uint64_t>(out.nodes4.size()) * sizeof(ptd::Node4) + (all_tris.size() + 2) * sizeof(ptd::TriQ);
        out.seven_waves = geometry_bytes <= (128ull << 20);
        if (opt.traverse_waves == 6 || opt.traverse_waves == 7) out.seven_waves = opt.traverse_waves == 7;
        const uint32_t capacity = out.seven_waves ? static_cast<uint32_t>(ptd::top_nodes7_for(static_cast<uint32_t>(table.instances.size())))
                                                  : static_cast<uint32_t>(ptd::kTopNodes4);
        out.n_top4 = number_top_nodes4(out.nodes4, mesh_root4, mesh_weight, capacity);
        // Leaves as one aligned 64-byte record each instead of two 48-byte triangle records that straddle two or three memory
        // sectors: for scenes beyond the caches (the six-workgroup form: limited by the sectors a ray touches; 10 M-triangle
        // frame: traversal -12.5 %) and for deep closest-hit traversals of large cache-resident scenes (readme frame -1.9 %).
        // Short traversals and any-hit rays, which often need only a leaf's first triangle, pay for the edge vectors the kernel
        // then forms itself (ambient occlusion +1.5 %, 247 K-triangle material scene +1.3 %) and keep the triangle records.
        uint64_t instanced_tris = 0;
        for (uint32_t i = 0; i < model.n_objects; ++i) instanced_tris += model.meshes[model.objects[i].mesh_index].n_indices / 3;
        const bool deep = opt.program_mode != PTGPU_PROGRAM_AO && instanced_tris >= 500000;
        const bool leaf_records = opt.leaf_records > 0 || (opt.leaf_records == 0 && (!out.seven_waves || deep));
        if (leaf_records) {
            make_leaf_records(all_tris, static_cast<uint32_t>(all_tris.size()), out.nodes4, mesh_root4, out.leaves64);
        }
        for (ptd::DevXform& xf : table.xforms) xf.root4 = mesh_root4[xf.mesh_index];
        // a visit pushes up to three children: the stack can hold 3 entries per level
        out.max_depth = std::max(out.max_depth, 3u * out.max_depth4 + 1u);
        if (opt.verbose)
            std::fprintf(stderr, "[ptgpu] four-wide nodes: %zu (from %zu binary), depth %u, kernel form: %d workgroups per CU\n", out.nodes4.size(),
                         all_nodes.size(), out.max_depth4, out.seven_waves ? 7 : 6);
    }

    // ---- shading data ----
    // corners: every triangle's three vertices de-indexed, in mesh order (the hit record's triangle index / 3)
    std::vector<uint64_t> mesh_tri_base(model.n_meshes);
    uint64_t n_shade_tris = 0;
    for (uint32_t i = 0; i < model.n_meshes; ++i) {
        mesh_tri_base[i] = n_shade_tris;
        n_shade_tris += model.meshes[i].n_indices / 3;
    }
    if (n_shade_tris >= (1ull << 32)) return fail(err, code, PTGPU_E_INVALID, "scene exceeds 2^32 triangles");
    out.corners.resize(static_cast<size_t>(n_shade_tris) * 3);
    for (uint32_t i = 0; i < model.n_meshes; ++i) {
        const ptgpu_mesh& m = model.meshes[i];
        ptgpu_vertex* dst = out.corners.data() + mesh_tri_base[i] * 3;
        const uint32_t n_corner = m.n_indices - m.n_indices % 3;
        for (uint32_t k = 0; k < n_corner; ++k) {
            const uint32_t v = model.indices[m.first_index + k];
            if (v >= m.n_vertices) return fail(err, code, PTGPU_E_INVALID, "mesh %u: index %u beyond its %u vertices", i, v, m.n_vertices);
            dst[k] = model.vertices[m.first_vertex + v];
        }
    }
    auto texture_of = [&](int32_t image, ptd::DevTexture& t) -> bool {
        t = ptd::DevTexture{0, 0, 0u};
        if (image < 0) return true;
        if (static_cast<uint32_t>(image) >= model.n_images) return false;
        const ptgpu_image& im = model.images[image];
        if (im.width <= 0 || im.height <= 0 || im.texel_offset + static_cast<uint64_t>(im.width) * im.height > model.n_texels ||
            im.texel_offset >= (1ull << 32))
            return false;
        t = ptd::DevTexture{im.width, im.height, static_cast<uint32_t>(im.texel_offset)};
        return true;
    };
    out.objects.resize(model.n_objects);
    out.object_mesh.resize(model.n_objects);
    for (uint32_t i = 0; i < model.n_objects; ++i) {
        const ptgpu_object& o = model.objects[i];
        ptd::DevObject& d = out.objects[i];
        std::memset(&d, 0, sizeof(d));
        for (int col = 0; col < 4; ++col)
            for (int r = 0; r < 3; ++r) {
                d.model[col * 3 + r] = o.model_matrix[col * 4 + r];
                d.normal[col * 3 + r] = o.normal_matrix[col * 4 + r];
            }
        d.tri_base = static_cast<uint32_t>(mesh_tri_base[o.mesh_index]);
        out.object_mesh[i] = o.mesh_index;
        const ptgpu_material& m = model.materials[model.meshes[o.mesh_index].material_index];
        ptd::DevMaterial& dm = d.mat;
        std::memcpy(dm.base_color, m.base_color_factor, 16);
        std::memcpy(dm.emissive, m.emissive_factor, 12);
        dm.emissive_strength = m.emissive_strength;
        std::memcpy(dm.attenuation, m.attenuation_factor, 12);
        dm.metallic = m.metallic_factor;
        dm.roughness = m.roughness_factor;
        dm.ior = m.ior;
        dm.f0 = m.dielectric_f0;
        dm.transmission = m.transmission_factor;
        dm.alpha_cutoff = m.alpha_cutoff;
        dm.double_sided = m.double_sided != 0;
        dm.has_volume = m.has_volume != 0;
        dm.alpha_blending = m.alpha_blending != 0;
        if (!texture_of(m.base_color_texture, dm.base_color_tex) || !texture_of(m.metallic_roughness_texture, dm.mr_tex) ||
            !texture_of(m.normal_texture, dm.normal_tex) || !texture_of(m.transmission_texture, dm.transmission_tex) ||
            !texture_of(m.emissive_texture, dm.emissive_tex))
            return fail(err, code, PTGPU_E_INVALID, "object %u: its material names an image that does not exist or lies outside the texel array", i);
    }
    static_assert(sizeof(ptgpu_vertex) == 48, "vertex is three float4");
    static_assert(sizeof(ptd::DevObject) == 256, "object record is 256 bytes");
    for (int i = 0; i < 256; ++i) out.srgb_lut[i] = srgb_to_linear_host(static_cast<float>(i) / 255.0f);

    // traversal forms: planes grouped per axis, triangle edges precomputed (binary32 subtraction, as the test does)
    out.nodes.resize(all_nodes.size());
    for (size_t i = 0; i < all_nodes.size(); ++i) {
        const ptd::DevNode& n = all_nodes[i];
        ptd::NodeQ& q = out.nodes[i];
        float* axes[3] = {q.x, q.y, q.z};
        for (int a = 0; a < 3; ++a) {
            axes[a][0] = n.l_lo[a];
            axes[a][1] = n.r_lo[a];
            axes[a][2] = n.l_hi[a];
            axes[a][3] = n.r_hi[a];
        }
        q.left = n.left;
        q.right = n.right;
        q.pad0 = q.pad1 = 0;
    }
    // + two records of padding: the four-wide kernel reads triangles in pairs, and the first padding record is itself a leaf
    // (one degenerate triangle: the filler of unused child slots)
    out.tris.resize(all_tris.size() + 2);
    for (size_t k = all_tris.size(); k < out.tris.size(); ++k) {
        std::memset(&out.tris[k], 0, sizeof(ptd::TriQ));
        out.tris[k].leaf_count = 1;
    }
    uint32_t in_leaf = 0;  // the builder marks a leaf's first triangle with the leaf size; every triangle gets the rest count
    for (size_t i = 0; i < all_tris.size(); ++i) {
        const ptd::DevTri& t = all_tris[i];
        ptd::TriQ& q = out.tris[i];
        for (int a = 0; a < 3; ++a) {
            q.p1[a] = t.p1[a];
            q.e1[a] = t.p2[a] - t.p1[a];
            q.e2[a] = t.p3[a] - t.p1[a];
        }
        q.index = t.index;
        if (t.leaf_count) in_leaf = t.leaf_count;
        q.leaf_count = in_leaf ? in_leaf-- : 1u;
        q.pad = 0;
    }
    out.instances = std::move(table.instances);
    out.xforms = std::move(table.xforms);
    {   // deep closest-hit traversals (large scenes, every program but ambient occlusion with its any-hit rays) give leaves the
        // heavier vote in the traversal kernel's choice of phase
        uint64_t instanced = 0;
        for (uint32_t i = 0; i < model.n_objects; ++i) instanced += model.meshes[model.objects[i].mesh_index].n_indices / 3;
        out.leaf_vote = (opt.program_mode != PTGPU_PROGRAM_AO && instanced >= 500000) ? 4u : 1u;
        if (opt.leaf_vote >= 1 && opt.leaf_vote <= 16) out.leaf_vote = opt.leaf_vote;
    }
    ++g_builds;
    return true;
}

}  // namespace pth
