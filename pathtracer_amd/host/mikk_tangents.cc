// mikk_tangents.cc — tangents for meshes that come with uv but without TANGENT.
//
// The reference hands such a mesh to the MikkTSpace library (compute_tangents.cc:9-62: genTangSpaceDefault with
// callbacks that read position / normal / uv through the index buffer and write the result into the INDEXED vertex,
// so of the corners that share a vertex the one of the highest face wins).  The library itself is not part of the
// reference's tree (an empty submodule), so this file restates its published algorithm (mikktspace.c, Morten S.
// Mikkelsen; default angular threshold of 180 degrees) for triangle meshes:
//   1. corners with identical position, normal and uv (bit for bit) are welded;
//   2. triangles with two equal corner positions are degenerate: they take no part and get a neighbour's result;
//   3. per triangle: the uv-gradient vectors vOs / vOt, their magnitudes, and whether the uv mapping preserves
//      orientation; a triangle without uv area may join any group;
//   4. groups: around every welded vertex, triangles connected through shared edges and of the same orientation;
//   5. per group and member face: the sub-group of faces whose projected vOs / vOt are within the threshold (with
//      180 degrees: practically all), evaluated as the angle-weighted sum of the members' vOs / vOt projected into
//      the plane of the vertex normal, in ascending face order;
//   6. every corner reports (vOs, orientation ? +1 : -1) to its indexed vertex, faces in ascending order.
// No reference output exists to pin this against (the library is absent and the reference's tests do not cover it);
// tests/test_gltf_loader_cpu.py checks analytic cases.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "app_model.h"

namespace app {
namespace {

struct V3 {
    float x = 0, y = 0, z = 0;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(float s, V3 v) { return {s * v.x, s * v.y, s * v.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(V3 v) { return std::sqrt(dot(v, v)); }
inline bool not_zero(float f) { return std::fabs(f) > FLT_MIN; }
inline bool not_zero(V3 v) { return not_zero(v.x) || not_zero(v.y) || not_zero(v.z); }
inline V3 normalized(V3 v) { return (1.0f / length(v)) * v; }
inline bool same(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline V3 pos_of(const vertex& v) { return {v.position.x, v.position.y, v.position.z}; }
inline V3 nrm_of(const vertex& v) { return {v.normal.x, v.normal.y, v.normal.z}; }

enum : uint32_t { kDegenerate = 1u, kGroupWithAny = 4u, kOrientPreserving = 8u };

struct Group {
    std::vector<int> faces;
    int vertex = 0;  // welded index the group is built around
    bool orient = false;
};
struct TriInfo {
    int neighbour[3] = {-1, -1, -1};
    int group[3] = {-1, -1, -1};
    V3 os, ot;
    float mag_s = 0, mag_t = 0;
    uint32_t flags = 0;
};
struct TSpace {
    V3 os{1.0f, 0.0f, 0.0f}, ot{0.0f, 1.0f, 0.0f};
    float mag_s = 1.0f, mag_t = 1.0f;
    int counter = 0;
    bool orient = false;
};
struct VertexKey {
    uint32_t w[8];
    bool operator==(const VertexKey& o) const { return std::memcmp(w, o.w, sizeof(w)) == 0; }
};
struct VertexKeyHash {
    size_t operator()(const VertexKey& k) const {
        uint64_t h = 1469598103934665603ull;
        for (uint32_t v : k.w) h = (h ^ v) * 1099511628211ull;
        return static_cast<size_t>(h);
    }
};

struct Builder {
    const Mesh& mesh;
    int n_tris;
    std::vector<int> corner;      // welded index per (triangle, corner): the first corner with the same data
    std::vector<int> corner_src;  // vertex of the mesh behind a welded index
    std::vector<TriInfo> tri;
    std::vector<Group> groups;

    explicit Builder(const Mesh& m) : mesh(m), n_tris(static_cast<int>(m.indices.size() / 3)) {}
    const vertex& vert(int welded) const { return mesh.vertices[corner_src[welded]]; }

    void weld() {
        corner.resize(static_cast<size_t>(n_tris) * 3);
        corner_src.resize(corner.size());
        std::unordered_map<VertexKey, int, VertexKeyHash> first;
        first.reserve(corner.size());
        for (size_t c = 0; c < corner.size(); ++c) {
            const vertex& v = mesh.vertices[mesh.indices[c]];
            VertexKey k;
            std::memcpy(k.w, &v.position.x, 12);
            std::memcpy(k.w + 3, &v.normal.x, 12);
            std::memcpy(k.w + 6, &v.uv.x, 8);
            for (uint32_t& w : k.w)
                if (w == 0x80000000u) w = 0;  // -0 equals +0 in a float comparison
            const int id = first.emplace(k, static_cast<int>(c)).first->second;
            corner[c] = id;
            corner_src[c] = static_cast<int>(mesh.indices[c]);
        }
    }
    void init_tri_info() {
        tri.assign(n_tris, TriInfo{});
        for (int f = 0; f < n_tris; ++f) {
            TriInfo& t = tri[f];
            const vertex &a = vert(corner[3 * f]), &b = vert(corner[3 * f + 1]), &c = vert(corner[3 * f + 2]);
            const V3 p0 = pos_of(a), p1 = pos_of(b), p2 = pos_of(c);
            if (same(p0, p1) || same(p0, p2) || same(p1, p2)) t.flags |= kDegenerate;
            t.flags |= kGroupWithAny;
            const float t21x = b.uv.x - a.uv.x, t21y = b.uv.y - a.uv.y, t31x = c.uv.x - a.uv.x, t31y = c.uv.y - a.uv.y;
            const V3 d1 = p1 - p0, d2 = p2 - p0;
            const float area2 = t21x * t31y - t21y * t31x;
            V3 os = (t31y * d1) - (t21y * d2);
            V3 ot = ((-t31x) * d1) + (t21x * d2);
            if (area2 > 0.0f) t.flags |= kOrientPreserving;
            if (not_zero(area2)) {
                const float abs_area = std::fabs(area2), len_os = length(os), len_ot = length(ot);
                const float s = (t.flags & kOrientPreserving) ? 1.0f : -1.0f;
                if (not_zero(len_os)) os = (s / len_os) * os;
                if (not_zero(len_ot)) ot = (s / len_ot) * ot;
                t.mag_s = len_os / abs_area;
                t.mag_t = len_ot / abs_area;
                if (not_zero(t.mag_s) && not_zero(t.mag_t)) t.flags &= ~kGroupWithAny;
            }
            t.os = os;
            t.ot = ot;
        }
    }
    void build_neighbours() {  // the other good triangle on each edge (edges as unordered pairs of welded indices)
        struct Edge {
            int a, b, f, e;
        };
        std::vector<Edge> edges;
        edges.reserve(static_cast<size_t>(n_tris) * 3);
        for (int f = 0; f < n_tris; ++f) {
            if (tri[f].flags & kDegenerate) continue;
            for (int e = 0; e < 3; ++e) {
                const int i0 = corner[3 * f + e], i1 = corner[3 * f + (e + 1) % 3];
                edges.push_back({std::min(i0, i1), std::max(i0, i1), f, e});
            }
        }
        std::sort(edges.begin(), edges.end(), [](const Edge& x, const Edge& y) {
            if (x.a != y.a) return x.a < y.a;
            if (x.b != y.b) return x.b < y.b;
            return x.f < y.f;
        });
        for (size_t i = 0; i < edges.size(); ++i) {
            const Edge& x = edges[i];
            if (tri[x.f].neighbour[x.e] != -1) continue;
            for (size_t j = i + 1; j < edges.size() && edges[j].a == x.a && edges[j].b == x.b; ++j) {
                const Edge& y = edges[j];
                if (tri[y.f].neighbour[y.e] == -1) {  // first unassigned partner on this edge
                    tri[x.f].neighbour[x.e] = y.f;
                    tri[y.f].neighbour[y.e] = x.f;
                    break;
                }
            }
        }
    }
    bool assign(int f, int g) {  // AssignRecur, iteratively in the same left-then-right order
        std::vector<int> todo{f};
        bool first_result = false, first = true;
        while (!todo.empty()) {
            const int my = todo.back();
            todo.pop_back();
            TriInfo& t = tri[my];
            bool ok = false;
            int i = -1;
            for (int k = 0; k < 3; ++k)
                if (corner[3 * my + k] == groups[g].vertex) { i = k; break; }
            if (i >= 0) {
                if (t.group[i] == g) {
                    ok = true;
                } else if (t.group[i] == -1) {
                    if ((t.flags & kGroupWithAny) && t.group[0] == -1 && t.group[1] == -1 && t.group[2] == -1) {
                        t.flags &= ~kOrientPreserving;  // a triangle without uv area takes the orientation of its first group
                        if (groups[g].orient) t.flags |= kOrientPreserving;
                    }
                    if (((t.flags & kOrientPreserving) != 0) == groups[g].orient) {
                        groups[g].faces.push_back(my);
                        t.group[i] = g;
                        ok = true;
                        const int left = t.neighbour[i], right = t.neighbour[i > 0 ? i - 1 : 2];
                        if (right >= 0) todo.push_back(right);  // popped after the whole left branch, like the recursion
                        if (left >= 0) todo.push_back(left);
                    }
                }
            }
            if (first) { first_result = ok; first = false; }
        }
        return first_result;
    }
    void build_groups() {
        for (int f = 0; f < n_tris; ++f) {
            if (tri[f].flags & kDegenerate) continue;
            for (int i = 0; i < 3; ++i) {
                if ((tri[f].flags & kGroupWithAny) || tri[f].group[i] != -1) continue;
                const int g = static_cast<int>(groups.size());
                groups.emplace_back();
                groups[g].vertex = corner[3 * f + i];
                groups[g].orient = (tri[f].flags & kOrientPreserving) != 0;
                groups[g].faces.push_back(f);
                tri[f].group[i] = g;
                const int left = tri[f].neighbour[i], right = tri[f].neighbour[i > 0 ? i - 1 : 2];
                if (left >= 0) assign(left, g);
                if (right >= 0) assign(right, g);
            }
        }
    }
    int corner_of(int f, int welded) const {
        for (int k = 0; k < 3; ++k)
            if (corner[3 * f + k] == welded) return k;
        return -1;
    }
    V3 projected(V3 v, V3 n) const {
        V3 r = v - (dot(n, v) * n);
        if (not_zero(r)) r = normalized(r);
        return r;
    }
    TSpace eval(const std::vector<int>& faces, int welded) const {  // EvalTspace
        TSpace res;
        res.os = res.ot = V3{};
        res.mag_s = res.mag_t = 0.0f;
        float angle_sum = 0.0f;
        for (int f : faces) {
            if (tri[f].flags & kGroupWithAny) continue;
            const int i = corner_of(f, welded);
            if (i < 0) continue;
            const V3 n = nrm_of(vert(corner[3 * f + i]));
            const V3 os = projected(tri[f].os, n), ot = projected(tri[f].ot, n);
            const V3 p0 = pos_of(vert(corner[3 * f + (i > 0 ? i - 1 : 2)])), p1 = pos_of(vert(corner[3 * f + i])),
                     p2 = pos_of(vert(corner[3 * f + (i < 2 ? i + 1 : 0)]));
            const V3 v1 = projected(p0 - p1, n), v2 = projected(p2 - p1, n);
            float c = dot(v1, v2);
            c = c > 1.0f ? 1.0f : (c < -1.0f ? -1.0f : c);
            const float angle = static_cast<float>(std::acos(static_cast<double>(c)));
            res.os = res.os + (angle * os);
            res.ot = res.ot + (angle * ot);
            res.mag_s += angle * tri[f].mag_s;
            res.mag_t += angle * tri[f].mag_t;
            angle_sum += angle;
        }
        if (not_zero(res.os)) res.os = normalized(res.os);
        if (not_zero(res.ot)) res.ot = normalized(res.ot);
        if (angle_sum > 0.0f) {
            res.mag_s /= angle_sum;
            res.mag_t /= angle_sum;
        }
        return res;
    }
    static TSpace average(const TSpace& a, const TSpace& b) {  // AvgTSpace
        TSpace r;
        if (a.mag_s == b.mag_s && a.mag_t == b.mag_t && same(a.os, b.os) && same(a.ot, b.ot)) {
            r = a;
        } else {
            r.mag_s = 0.5f * (a.mag_s + b.mag_s);
            r.mag_t = 0.5f * (a.mag_t + b.mag_t);
            r.os = a.os + b.os;
            r.ot = a.ot + b.ot;
            if (not_zero(r.os)) r.os = normalized(r.os);
            if (not_zero(r.ot)) r.ot = normalized(r.ot);
        }
        return r;
    }
    std::vector<TSpace> generate() {  // GenerateTSpaces with cos(180 degrees) as threshold
        const float thres_cos = static_cast<float>(std::cos(3.14159265358979323846));
        std::vector<TSpace> out(static_cast<size_t>(n_tris) * 3);
        std::vector<int> members;
        for (const Group& g : groups) {
            std::vector<std::vector<int>> sub_members;
            std::vector<TSpace> sub_space;
            for (int f : g.faces) {
                const int index = corner_of(f, g.vertex);
                if (index < 0) continue;
                const V3 n = nrm_of(vert(g.vertex));
                const V3 os = projected(tri[f].os, n), ot = projected(tri[f].ot, n);
                members.clear();
                for (int t : g.faces) {
                    const V3 os2 = projected(tri[t].os, n), ot2 = projected(tri[t].ot, n);
                    const bool any = ((tri[f].flags | tri[t].flags) & kGroupWithAny) != 0;
                    const float cos_s = dot(os, os2), cos_t = dot(ot, ot2);
                    if (any || f == t || (cos_s > thres_cos && cos_t > thres_cos)) members.push_back(t);
                }
                std::sort(members.begin(), members.end());
                size_t l = 0;
                while (l < sub_members.size() && sub_members[l] != members) ++l;
                if (l == sub_members.size()) {
                    sub_members.push_back(members);
                    sub_space.push_back(eval(members, g.vertex));
                }
                TSpace& dst = out[static_cast<size_t>(f) * 3 + index];
                if (dst.counter == 1) {
                    dst = average(dst, sub_space[l]);
                    dst.counter = 2;
                    dst.orient = g.orient;
                } else {
                    dst = sub_space[l];
                    dst.counter = 1;
                    dst.orient = g.orient;
                }
            }
        }
        return out;
    }
    void degenerate_epilogue(std::vector<TSpace>& ts) const {  // a degenerate triangle's corners copy a good corner at the same welded vertex
        std::unordered_map<int, size_t> good;  // welded index -> a corner of a good triangle
        for (int f = 0; f < n_tris; ++f)
            if (!(tri[f].flags & kDegenerate))
                for (int i = 0; i < 3; ++i) good.emplace(corner[3 * f + i], static_cast<size_t>(f) * 3 + i);
        for (int f = 0; f < n_tris; ++f)
            if (tri[f].flags & kDegenerate)
                for (int i = 0; i < 3; ++i) {
                    const auto it = good.find(corner[3 * f + i]);
                    if (it != good.end()) ts[static_cast<size_t>(f) * 3 + i] = ts[it->second];
                }
    }
};

}  // namespace

// TangentSpaceHelper::compute_tangents (compute_tangents.cc:58-62) with setTSpaceBasic's write rule (:36-42)
void compute_tangents_mikk(Mesh& mesh) {
    Builder b(mesh);
    if (b.n_tris == 0) return;
    b.weld();
    b.init_tri_info();
    b.build_neighbours();
    b.build_groups();
    std::vector<TSpace> ts = b.generate();
    b.degenerate_epilogue(ts);
    for (int f = 0; f < b.n_tris; ++f)
        for (int i = 0; i < 3; ++i) {
            const TSpace& t = ts[static_cast<size_t>(f) * 3 + i];
            vertex& v = mesh.vertices[mesh.indices[static_cast<size_t>(f) * 3 + i]];
            v.tangent = fvec4{t.os.x, t.os.y, t.os.z, t.orient ? 1.0f : -1.0f};
        }
}

}  // namespace app
