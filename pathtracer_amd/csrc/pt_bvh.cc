// pt_bvh.cc — host builder of the per-mesh BVH and the instance table (see pt_bvh.h).
// Float arithmetic is written operation by operation like the reference (built with contraction off),
// so the tree is the one BVH_AS builds: same bins, same split planes, same partitions.
#include "pt_bvh.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>

namespace pth {
namespace {

constexpr float kInf = std::numeric_limits<float>::infinity();
constexpr float kMax = std::numeric_limits<float>::max();
constexpr float kLowest = std::numeric_limits<float>::lowest();

inline float lo_sel(float a, float b) { return (b < a) ? b : a; }  // glm::min
inline float hi_sel(float a, float b) { return (a < b) ? b : a; }  // glm::max

struct Box {
    float lo[3] = {kMax, kMax, kMax};
    float hi[3] = {kLowest, kLowest, kLowest};
    bool empty() const { return hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2]; }
    void add_point(const float* p) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = lo_sel(lo[a], p[a]);
            hi[a] = hi_sel(hi[a], p[a]);
        }
    }
    void add_box(const Box& b) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = lo_sel(lo[a], b.lo[a]);
            hi[a] = hi_sel(hi[a], b.hi[a]);
        }
    }
    float half_area() const {  // BBox::surface_area, acceleration_structure.cc:124-127 (the 2x is left out there too)
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dx * dz + dy * dz;
    }
};

inline Box range_box(const ptd::DevTri* b, const ptd::DevTri* e) {  // triangles_to_bbox, :572-585
    Box all;
    for (; b != e; ++b) {
        Box t;
        std::memcpy(t.lo, b->p1, 12);
        std::memcpy(t.hi, b->p1, 12);
        t.add_point(b->p2);
        t.add_point(b->p3);
        all.add_box(t);
    }
    return all;
}

inline int widest_axis(const Box& b) {  // get_longest_axis, :51-59
    const float ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
    if (ex >= ey && ex >= ez) return 0;
    if (ey >= ex && ey >= ez) return 1;
    return 2;
}

// `int bin = float_value;` (:472) has no defined result for NaN or out-of-range values; the x86
// conversion instruction every build of the reference uses yields INT_MIN there.
inline int to_int_like_x86(float f) {
    if (!(f == f) || f >= 2147483648.0f || f < -2147483648.0f) return std::numeric_limits<int>::min();
    return static_cast<int>(f);
}

inline float centroid3(const ptd::DevTri& t, int axis) { return t.p1[axis] + t.p2[axis] + t.p3[axis]; }

// split_triangles (:449-536): returns the index of the first triangle of the right part, or `end`.
uint32_t choose_split(std::vector<ptd::DevTri>& tris, uint32_t begin, uint32_t end, const Box& box, uint32_t max_leaf) {
    constexpr int kBins = 32;
    struct Bin {
        Box box;
        int n = 0;
    };
    const int n = static_cast<int>(end - begin);
    const float parent_cost = box.empty() ? kInf : box.half_area() * n;
    float best_cost = parent_cost;
    const float dim[3] = {box.hi[0] - box.lo[0], box.hi[1] - box.lo[1], box.hi[2] - box.lo[2]};
    const float to_bin[3] = {(kBins * 1.0f) / dim[0], (kBins * 1.0f) / dim[1], (kBins * 1.0f) / dim[2]};
    int best_axis = widest_axis(box);
    float plane = 0.0f;
    ptd::DevTri* first = tris.data() + begin;
    ptd::DevTri* last = tris.data() + end;

    for (int axis = 0; axis < 3; ++axis) {
        std::array<Bin, kBins> bins;
        for (ptd::DevTri* t = first; t != last; ++t) {
            const float c = centroid3(*t, axis) / 3.0f;
            int b = to_int_like_x86((c - box.lo[axis]) * to_bin[axis]);
            b = std::clamp(b, 0, kBins - 1);
            ++bins[b].n;
            bins[b].box.add_point(t->p1);
            bins[b].box.add_point(t->p2);
            bins[b].box.add_point(t->p3);
        }
        std::array<Bin, kBins - 1> from_left, from_right;
        from_left[0] = bins[0];
        for (int i = 1; i < kBins - 1; ++i) {
            from_left[i] = from_left[i - 1];
            from_left[i].box.add_box(bins[i].box);
            from_left[i].n += bins[i].n;
        }
        from_right[kBins - 2] = bins[kBins - 1];
        for (int i = kBins - 3; i >= 0; --i) {
            // the reference accumulates bins[i] here (not bins[i + 1]); kept so the tree is the same (:490-494)
            from_right[i] = from_right[i + 1];
            from_right[i].box.add_box(bins[i].box);
            from_right[i].n += bins[i].n;
        }
        for (int i = 0; i < kBins - 1; ++i) {
            const Bin& l = from_left[i];
            const Bin& r = from_right[i];
            const float cost = (l.box.empty() || r.box.empty()) ? kInf : l.box.half_area() * l.n + r.box.half_area() * r.n;
            if (cost < best_cost) {
                plane = box.lo[axis] + (i + 1) * dim[axis] * (1.0f / kBins);
                best_cost = cost;
                best_axis = axis;
            }
        }
    }
    const bool improved = (best_cost != parent_cost);
    if (!improved) {
        best_axis = widest_axis(box);
        plane = box.lo[best_axis] + 0.5f * dim[best_axis];
    }
    ptd::DevTri* mid = std::partition(first, last, [plane, best_axis](const ptd::DevTri& t) {
        return centroid3(t, best_axis) / 3.0f < plane;
    });
    // A split plane found by the bins can leave every centroid on its right once compared in binary32 (meshes far from
    // their origin): an empty left part, and a right part equal to the parent.  The reference recurses without end
    // there (:443-444); this builder takes the median split in that case — its only deviation from the reference's tree.
    const bool empty_left = improved && mid == first;
    if (improved && !empty_left) return begin + static_cast<uint32_t>(mid - first);
    if (!empty_left && static_cast<uint32_t>(n) <= max_leaf) return end;
    if (mid == first || mid == last || mid == first + 1 || mid + 1 == last) {
        std::sort(first, last, [best_axis](const ptd::DevTri& a, const ptd::DevTri& b) {
            return centroid3(a, best_axis) < centroid3(b, best_axis);
        });
        mid = first + n / 2;
    }
    return begin + static_cast<uint32_t>(mid - first);
}

// The split used under the four-wide nodes, where the tree need not be the reference's: surface-area heuristic over
// bins that span the CENTROIDS of the range (not its box), right-hand sums taken over exactly the bins right of the
// plane, and every split position tried (three sorted sweeps) once a range is small.  Always splits a range of two or
// more triangles; returns the index of the first triangle of the right part.
inline Box tri_box(const ptd::DevTri& t) {
    Box b;
    b.add_point(t.p1);
    b.add_point(t.p2);
    b.add_point(t.p3);
    return b;
}
uint32_t choose_split_tuned(std::vector<ptd::DevTri>& tris, uint32_t begin, uint32_t end) {
    constexpr int kBins = 32;
    constexpr int kSweepBelow = 48;
    const int n = static_cast<int>(end - begin);
    ptd::DevTri* first = tris.data() + begin;
    float clo[3] = {kMax, kMax, kMax}, chi[3] = {kLowest, kLowest, kLowest};
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            const float c = centroid3(first[i], a);
            clo[a] = lo_sel(clo[a], c);
            chi[a] = hi_sel(chi[a], c);
        }
    float best_cost = kInf;
    if (n <= kSweepBelow) {
        int order[3][kSweepBelow];
        int best_axis = -1, best_at = 0;
        float right_area[kSweepBelow];
        for (int a = 0; a < 3; ++a) {
            if (!(chi[a] > clo[a])) continue;
            int* o = order[a];
            for (int i = 0; i < n; ++i) o[i] = i;
            std::stable_sort(o, o + n, [&](int x, int y) { return centroid3(first[x], a) < centroid3(first[y], a); });
            Box acc;
            for (int i = n - 1; i > 0; --i) {
                acc.add_box(tri_box(first[o[i]]));
                right_area[i] = acc.half_area();
            }
            Box left;
            for (int i = 1; i < n; ++i) {  // i triangles on the left
                left.add_box(tri_box(first[o[i - 1]]));
                const float cost = left.half_area() * i + right_area[i] * (n - i);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = a;
                    best_at = i;
                }
            }
        }
        if (best_axis < 0) return begin + static_cast<uint32_t>(n / 2);
        ptd::DevTri tmp[kSweepBelow];
        for (int i = 0; i < n; ++i) tmp[i] = first[order[best_axis][i]];
        for (int i = 0; i < n; ++i) first[i] = tmp[i];
        return begin + static_cast<uint32_t>(best_at);
    }
    struct Bin {
        Box box;
        int n = 0;
    };
    int best_axis = -1, best_bin = 0;
    float best_scale = 0.0f;
    for (int a = 0; a < 3; ++a) {
        if (!(chi[a] > clo[a])) continue;
        const float scale = static_cast<float>(kBins) / (chi[a] - clo[a]);
        std::array<Bin, kBins> bins;
        for (int i = 0; i < n; ++i) {
            const int b = std::clamp(to_int_like_x86((centroid3(first[i], a) - clo[a]) * scale), 0, kBins - 1);
            ++bins[b].n;
            bins[b].box.add_box(tri_box(first[i]));
        }
        float right_area[kBins];
        int right_n[kBins];
        Box acc;
        int cnt = 0;
        for (int b = kBins - 1; b > 0; --b) {
            acc.add_box(bins[b].box);
            cnt += bins[b].n;
            right_area[b] = acc.empty() ? 0.0f : acc.half_area();
            right_n[b] = cnt;
        }
        Box left;
        int left_n = 0;
        for (int b = 1; b < kBins; ++b) {  // plane between bin b - 1 and bin b
            left.add_box(bins[b - 1].box);
            left_n += bins[b - 1].n;
            if (left_n == 0 || right_n[b] == 0) continue;
            const float cost = left.half_area() * left_n + right_area[b] * right_n[b];
            if (cost < best_cost) {
                best_cost = cost;
                best_axis = a;
                best_bin = b;
                best_scale = scale;
            }
        }
    }
    if (best_axis < 0) return begin + static_cast<uint32_t>(n / 2);  // every centroid in one place
    const float lo = clo[best_axis];
    ptd::DevTri* mid = std::partition(first, first + n, [=](const ptd::DevTri& t) {
        return std::clamp(to_int_like_x86((centroid3(t, best_axis) - lo) * best_scale), 0, kBins - 1) < best_bin;
    });
    return begin + static_cast<uint32_t>(mid - first);
}

inline void fnv_u32(uint64_t& h, uint32_t v) {
    for (int i = 0; i < 4; ++i) {
        h ^= (v >> (8 * i)) & 0xffu;
        h *= 1099511628211ull;
    }
}

// A range of triangles waiting to become a subtree, and the slot of its parent that receives box and reference.
struct Task {
    uint32_t begin, end;
    uint32_t parent;  // node index, or PT_NO_NODE for the root of this build
    bool right;
};

// parse (:406-447) with an explicit work list: left ranges are processed before right ranges, so nodes are numbered
// in depth-first order.  Ranges of at most `defer_below` triangles (when non-zero) are recorded in `deferred` with their
// box already stored in the parent, instead of being expanded.
void build_range(std::vector<ptd::DevTri>& tris, Task first, uint32_t max_leaf, bool tuned, std::vector<ptd::DevNode>& nodes,
                 uint32_t* root, uint32_t defer_below, std::vector<Task>* deferred) {
    std::vector<Task> work{first};
    while (!work.empty()) {
        const Task tk = work.back();
        work.pop_back();
        const Box box = range_box(tris.data() + tk.begin, tris.data() + tk.end);
        const uint32_t count = tk.end - tk.begin;
        uint32_t ref = PT_NO_NODE;
        const bool defer = defer_below && deferred && tk.parent != PT_NO_NODE && count <= defer_below && count > max_leaf;
        if (defer) {
            deferred->push_back(tk);
        } else {
            uint32_t mid = tk.end;
            if (count > max_leaf) mid = tuned ? choose_split_tuned(tris, tk.begin, tk.end) : choose_split(tris, tk.begin, tk.end, box, max_leaf);
            if (mid == tk.end) {
                tris[tk.begin].leaf_count = count;
                ref = PT_LEAF_BIT | tk.begin;
            } else {
                ref = static_cast<uint32_t>(nodes.size());
                nodes.emplace_back();
                work.push_back({mid, tk.end, ref, true});
                work.push_back({tk.begin, mid, ref, false});
            }
        }
        if (tk.parent == PT_NO_NODE) {
            *root = ref;
        } else {
            ptd::DevNode& p = nodes[tk.parent];
            if (tk.right) {
                std::memcpy(p.r_lo, box.lo, 12);
                std::memcpy(p.r_hi, box.hi, 12);
                p.right = ref;
            } else {
                std::memcpy(p.l_lo, box.lo, 12);
                std::memcpy(p.l_hi, box.hi, 12);
                p.left = ref;
            }
        }
    }
}

}  // namespace

void collect_info(MeshTree& t, uint32_t n_real_tris) {  // collect_tree_info, :538-570
    ptgpu_bvh_info r{};
    r.min_depth = 0xffffffffu;
    r.min_tris_in_leaf = 0xffffffffu;
    uint64_t depth_sum = 0, tri_sum = 0, h = 1469598103934665603ull;
    struct Item {
        uint32_t ref, depth;
    };
    std::vector<Item> todo{{t.root, 0u}};
    while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        if (!(it.ref & PT_LEAF_BIT)) {
            const ptd::DevNode& nd = t.nodes[it.ref];
            ++r.n_nodes;
            todo.push_back({nd.right, it.depth + 1});
            todo.push_back({nd.left, it.depth + 1});
            continue;
        }
        const uint32_t first = it.ref & ~PT_LEAF_BIT;
        const uint32_t count = n_real_tris ? t.tris[first].leaf_count : 0u;
        ++r.n_leaves;
        r.min_depth = std::min(r.min_depth, it.depth);
        r.max_depth = std::max(r.max_depth, it.depth);
        depth_sum += it.depth;
        r.min_tris_in_leaf = std::min(r.min_tris_in_leaf, count);
        r.max_tris_in_leaf = std::max(r.max_tris_in_leaf, count);
        tri_sum += count;
        fnv_u32(h, count);
        for (uint32_t k = 0; k < count; ++k) fnv_u32(h, t.tris[first + k].index);
    }
    r.n_triangles = n_real_tris;
    r.mean_depth = static_cast<float>(depth_sum) / r.n_leaves;
    r.mean_tris_in_leaf = static_cast<float>(tri_sum) / r.n_leaves;
    r.checksum = h;
    t.info = r;
}

void build_mesh_tree(const ptgpu_vertex* vertices, const uint32_t* indices, uint32_t n_indices, uint32_t max_leaf,
                     bool single_leaf, bool tuned, MeshTree& out, unsigned build_threads) {
    out.nodes.clear();
    out.tris.clear();
    const uint32_t n_tris = n_indices / 3;
    out.tris.reserve(std::max<uint32_t>(n_tris, 1));
    for (uint32_t i = 0; i + 2 < n_indices; i += 3) {  // MeshBVHData ctor, :396-401
        ptd::DevTri t{};
        std::memcpy(t.p1, vertices[indices[i]].position, 12);
        std::memcpy(t.p2, vertices[indices[i + 1]].position, 12);
        std::memcpy(t.p3, vertices[indices[i + 2]].position, 12);
        t.index = i;
        out.tris.push_back(t);
    }
    if (n_tris == 0) {
        // an empty mesh is a leaf without triangles in the reference; the device leaf loop always tests
        // one triangle, so give it a degenerate one (det == 0: never hit)
        ptd::DevTri t{};
        t.leaf_count = 1;
        out.tris.push_back(t);
        out.root = PT_LEAF_BIT | 0u;
        collect_info(out, 0);
        return;
    }
    if (single_leaf || n_tris <= max_leaf) {
        out.tris[0].leaf_count = n_tris;
        out.root = PT_LEAF_BIT | 0u;
        collect_info(out, n_tris);
        return;
    }
    // Top-down build with an explicit work list (build_range).  Ranges of at most `defer_below` triangles are not
    // expanded by the first, serial pass: they are independent subtrees (choose_split only permutes triangles inside
    // its own range), built by a few threads into private node arrays and appended in left-to-right order.  The
    // tree — boxes, split planes, triangle order, leaves — is the one a single-threaded build produces; only the
    // numbering of the nodes differs (top part first, then the subtrees).
    std::vector<Task> deferred;
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const unsigned n_threads = build_threads ? std::min(build_threads, 64u) : hw;
    const uint32_t defer_below = (n_threads > 1 && n_tris >= (1u << 16)) ? std::max<uint32_t>(n_tris / (n_threads * 8u), max_leaf) : 0u;
    build_range(out.tris, Task{0u, n_tris, PT_NO_NODE, false}, max_leaf, tuned, out.nodes, &out.root, defer_below, &deferred);
    if (!deferred.empty()) {
        struct Subtree {
            std::vector<ptd::DevNode> nodes;
            uint32_t root = 0;
        };
        std::vector<Subtree> subs(deferred.size());
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < deferred.size();)
                build_range(out.tris, Task{deferred[i].begin, deferred[i].end, PT_NO_NODE, false}, max_leaf, tuned, subs[i].nodes, &subs[i].root, 0,
                            nullptr);
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(worker);
        worker();
        for (std::thread& t : pool) t.join();
        for (size_t i = 0; i < deferred.size(); ++i) {
            const uint32_t base = static_cast<uint32_t>(out.nodes.size());
            auto move = [base](uint32_t ref) { return (ref & PT_LEAF_BIT) ? ref : ref + base; };
            for (ptd::DevNode n : subs[i].nodes) {
                n.left = move(n.left);
                n.right = move(n.right);
                out.nodes.push_back(n);
            }
            // the parent slot already holds the subtree's box (written by the serial pass); only the reference is open
            ptd::DevNode& p = out.nodes[deferred[i].parent];
            (deferred[i].right ? p.right : p.left) = move(subs[i].root);
        }
    }
    collect_info(out, n_tris);
}

void relocate(MeshTree& t, uint32_t node_base, uint32_t tri_base) {
    auto move = [&](uint32_t ref) {
        return (ref & PT_LEAF_BIT) ? (PT_LEAF_BIT | ((ref & ~PT_LEAF_BIT) + tri_base)) : ref + node_base;
    };
    for (ptd::DevNode& n : t.nodes) {
        n.left = move(n.left);
        n.right = move(n.right);
    }
    t.root = move(t.root);
}

void build_instances(const ptgpu_model& model, InstanceTable& out) {
    static const float s3 = 0.577350269189625764509148780501957456f;  // std::numbers::inv_sqrt3_v<float>
    static const float axes[7][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {s3, s3, s3}, {s3, s3, -s3}, {s3, -s3, s3}, {-s3, s3, s3}};
    struct Entry {
        ptd::DevInstance inst;
        ptd::DevXform xf;
        uint32_t complexity;
        bool hidden;
    };
    constexpr float kMaxMatrixEntry = 1e18f;  // (what validate_model asks of a model matrix)
    std::vector<Entry> entries(model.n_objects);
    for (uint32_t i = 0; i < model.n_objects; ++i) {
        const ptgpu_object& o = model.objects[i];
        const ptgpu_mesh& m = model.meshes[o.mesh_index];
        Entry& e = entries[i];
        std::memset(&e, 0, sizeof(e));
        for (int a = 0; a < 7; ++a) {
            e.inst.dop[2 * a] = kMax;
            e.inst.dop[2 * a + 1] = kLowest;
        }
        const float* M = o.model_matrix;
        for (uint32_t v = 0; v < m.n_vertices; ++v) {  // object_to_ws_dop, :208-220
            const float* p = model.vertices[m.first_vertex + v].position;
            float w[3];
            for (int r = 0; r < 3; ++r) w[r] = (M[r] * p[0] + M[4 + r] * p[1]) + (M[8 + r] * p[2] + M[12 + r] * 1.0f);
            for (int a = 0; a < 7; ++a) {
                const float pr = w[0] * axes[a][0] + w[1] * axes[a][1] + w[2] * axes[a][2];
                e.inst.dop[2 * a] = std::min(e.inst.dop[2 * a], pr);
                e.inst.dop[2 * a + 1] = std::max(e.inst.dop[2 * a + 1], pr);
            }
        }
        e.inst.object_index = i;
        e.inst.mesh_index = o.mesh_index;
        // invModelMatrix = transpose(NormalMatrix) (:363); keep rows 0..2 of every column
        const float* N = o.normal_matrix;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 3; ++r) {
                e.xf.inv[c * 3 + r] = N[r * 4 + c];
                e.xf.model[c * 3 + r] = M[c * 4 + r];
            }
        e.xf.object_index = i;
        e.xf.mesh_index = o.mesh_index;
        e.complexity = m.n_indices;
        // A model matrix that cannot be inverted (an object scaled to nothing, the usual way of hiding one) comes with a
        // normal matrix of infinities and NaNs.  The reference keeps such an object and never hits it: every box test of a
        // ray taken through that matrix compares NaNs and fails.  Here it is left out of the instance table — the same
        // picture without the visit — but still counts for the scene's bounds.
        for (int k = 0; k < 16; ++k)
            if (!(std::fabs(N[k]) <= kMaxMatrixEntry)) e.hidden = true;
    }
    std::sort(entries.begin(), entries.end(), [](const Entry& a, const Entry& b) { return a.complexity > b.complexity; });
    out.instances.clear();
    out.xforms.clear();
    float dop[14];
    for (int a = 0; a < 7; ++a) {
        dop[2 * a] = kMax;
        dop[2 * a + 1] = kLowest;
    }
    for (const Entry& e : entries) {
        if (!e.hidden) {
            out.instances.push_back(e.inst);
            out.xforms.push_back(e.xf);
        }
        for (int a = 0; a < 7; ++a) {
            dop[2 * a] = std::min(dop[2 * a], e.inst.dop[2 * a]);
            dop[2 * a + 1] = std::max(dop[2 * a + 1], e.inst.dop[2 * a + 1]);
        }
    }
    for (int a = 0; a < 3; ++a) {
        out.scene_min[a] = dop[2 * a];
        out.scene_max[a] = dop[2 * a + 1];
    }
}

}  // namespace pth
