// pt_oracle.cc — CPU parity oracle, part 2: two-level traversal, shading programs, the frame loop
// and the C interface.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see pt_oracle.h).
// Citations are file:line in the reference tree (zarond/PathTracer).

#include "pt_oracle_core.h"

using namespace orc;

namespace {

// ------------------------------------------------------------------------------------------
// material sampling, src/brdf.cc
// ------------------------------------------------------------------------------------------
struct SceneData {
    std::vector<Mesh> meshes;
    std::vector<Object> objects;
    std::vector<orc_material> materials;
    std::vector<Tex8> images;
    TexF env;
    // owned copies of the caller's buffers
    std::vector<orc_vertex> vertex_store;
    std::vector<uint32_t> index_store;
    std::vector<uint8_t> texel_store;
    std::vector<float> env_store;
};

inline const Tex8* tex_or_null(const SceneData& sc, int idx) { return idx >= 0 ? &sc.images[idx] : nullptr; }
inline V4 sample_srgba(V4 factor, const Tex8* t, V2 uv) {  // brdf.cc:11-17
    if (t) factor = factor * sample_bilinear(*t, uv, true);
    return factor;
}
inline V4 sample_rgba(V4 factor, const Tex8* t, V2 uv) {  // brdf.cc:18-24
    if (t) factor = factor * sample_bilinear(*t, uv, false);
    return factor;
}
inline V4 sample_albedo(const SceneData& sc, const orc_material& m, V2 uv) {  // brdf.cc:36-39
    return sample_srgba(V4(m.base_color_factor[0], m.base_color_factor[1], m.base_color_factor[2], m.base_color_factor[3]),
                        tex_or_null(sc, m.base_color_texture), uv);
}
inline V4 sample_roughness_metallic(const SceneData& sc, const orc_material& m, V2 uv) {  // brdf.cc:40-44
    return sample_rgba(V4(1.0f, m.roughness_factor, m.metallic_factor, 1.0f), tex_or_null(sc, m.metallic_roughness_texture), uv);
}
inline V4 sample_normals(const SceneData& sc, const orc_material& m, V2 uv) {  // brdf.cc:45-52
    if (const Tex8* t = tex_or_null(sc, m.normal_texture)) return sample_bilinear(*t, uv, false);
    return V4(0.5f, 0.5f, 1.0f, 0.0f);
}
inline V4 sample_emissive(const SceneData& sc, const orc_material& m, V2 uv) {  // brdf.cc:53-57
    return sample_srgba(V4(m.emissive_factor[0], m.emissive_factor[1], m.emissive_factor[2], 1.0f),
                        tex_or_null(sc, m.emissive_texture), uv) * m.emissive_strength;
}
inline float sample_transmission(const SceneData& sc, const orc_material& m, V2 uv) {  // brdf.cc:25-31, 58-61
    float f = m.transmission_factor;
    if (const Tex8* t = tex_or_null(sc, m.transmission_texture)) f *= sample_bilinear(*t, uv, false).x;
    return f;
}
inline V4 sample_environment(V3 dir, const TexF& env, float rot, const Mathx& mx) {  // brdf.cc:62-69
    V2 uv = V2{mx.atan2(-dir.z, -dir.x) + rot, -2.0f * mx.asin(dir.y)} * (1.0f / kPi);
    uv = uv * 0.5f + V2{0.5f, 0.5f};
    return sample_bilinear(env, uv);
}
inline float fibonacci1d(float i) { return std::fmod((i + 1.0f) * 1.618034f, 1.0f); }  // brdf.cc:9, 73-74
inline V2 fibonacci2d(int i, float inv_n) {                                            // brdf.cc:75-78
    float f = static_cast<float>(i);
    return V2{(f + 0.5f) * inv_n, fibonacci1d(f)};
}

// ------------------------------------------------------------------------------------------
// acceleration structures, src/acceleration_structure.cc:222-348 (Naive) and :352-716 (BVH)
// ------------------------------------------------------------------------------------------
struct NaiveMesh {
    std::vector<V3> verts;  // three per triangle
    bool double_sided = false;
};
struct Accel {
    bool bvh = true;
    std::vector<ObjectData> objects;  // sorted by complexity, descending (:370-374)
    std::vector<MeshBvh> bvh_meshes;
    std::vector<NaiveMesh> naive_meshes;

    void build(const SceneData& sc, bool use_bvh, uint32_t max_leaf) {
        bvh = use_bvh;
        for (uint32_t i = 0; i < sc.objects.size(); ++i) {
            const Object& o = sc.objects[i];
            const Mesh& m = sc.meshes[o.mesh];
            ObjectData od;
            for (uint32_t v = 0; v < m.n_vertices; ++v)  // object_to_ws_dop, :208-220
                od.volume.grow(xyz(mul(o.model, V4(vpos(m.vertices[v]), 1.0f))));
            od.model = o.model;
            od.inv_model = transpose(o.normal);  // :363
            od.object = i;
            od.mesh = o.mesh;
            od.complexity = m.n_indices;
            objects.push_back(od);
        }
        std::sort(objects.begin(), objects.end(),
                  [](const ObjectData& a, const ObjectData& b) { return a.complexity > b.complexity; });
        for (const Mesh& m : sc.meshes) {
            const orc_material& mat = sc.materials[m.material];
            bool ds = mat.double_sided || mat.has_volume;  // :377-378
            if (bvh) {
                bvh_meshes.emplace_back();
                bvh_meshes.back().build(m, ds, max_leaf);
            } else {
                NaiveMesh nm;
                nm.double_sided = ds;
                nm.verts.reserve(m.n_indices);
                for (uint32_t k = 0; k < m.n_indices; ++k) nm.verts.push_back(vpos(m.vertices[m.indices[k]]));
                naive_meshes.push_back(std::move(nm));
            }
        }
    }

    // BVH_AS::mesh_ray_intersection, :635-708
    template <bool ANY>
    TriHit trace_bvh(const Ray& ws, const ObjectData& od) const {
        const MeshBvh& mesh = bvh_meshes[od.mesh];
        V3 o_os = xyz(mul(od.inv_model, V4(ws.origin, 1.0f)));
        V3 d_os = normalize(xyz(mul(od.inv_model, V4(ws.direction, 0.0f))));
        V3 inv_dir = 1.0f / d_os;
        Ray r{o_os, d_os};
        TriHit hit;
        static thread_local std::vector<uint32_t> stack;
        stack.clear();
        stack.push_back(0);
        Counters& cnt = tl_counters;
        while (!stack.empty()) {
            uint32_t cur = stack.back();
            stack.pop_back();
            const BvhNode& nd = mesh.nodes[cur];
            if (nd.leaf) {
                for (uint32_t k = 0; k < nd.count; ++k) {
                    const BvhTri& tr = mesh.tris[nd.first + k];
                    ++cnt.tris;
                    TriHit th = ray_triangle(r, tr.p1, tr.p2, tr.p3, !mesh.double_sided);
                    if (!th.hit || th.t < 0.0f) continue;
                    if (th.t >= hit.t) continue;
                    hit.A = th.A; hit.B = th.B; hit.t = th.t; hit.hit = th.hit; hit.backface = th.backface;
                    hit.triangle = tr.index;
                    if (ANY) break;
                }
            } else {
                ++cnt.nodes;
                VolumeHit vl = mesh.nodes[nd.left].box.test(r, inv_dir);
                VolumeHit vr = mesh.nodes[nd.right].box.test(r, inv_dir);
                float dl = vl.forward_distance(), dr = vr.forward_distance();
                if (dl < dr) {
                    if (vr.hit && hit.t > dr) stack.push_back(nd.right);
                    if (vl.hit && hit.t > dl) stack.push_back(nd.left);
                } else {
                    if (vl.hit && hit.t > dl) stack.push_back(nd.left);
                    if (vr.hit && hit.t > dr) stack.push_back(nd.right);
                }
            }
            if (ANY && hit.hit) break;
        }
        if (!hit.hit) return hit;
        V3 p = r.origin + r.direction * hit.t;  // :704-706
        V3 p_ws = xyz(mul(od.model, V4(p, 1.0f)));
        hit.t = length(p_ws - ws.origin);
        return hit;
    }
    // NaiveAS::mesh_ray_intersection, :299-340
    template <bool ANY>
    TriHit trace_naive(const Ray& ws, const ObjectData& od) const {
        const NaiveMesh& mesh = naive_meshes[od.mesh];
        V3 o_os = xyz(mul(od.inv_model, V4(ws.origin, 1.0f)));
        V3 d_os = xyz(mul(od.inv_model, V4(ws.direction, 0.0f)));
        Ray r{o_os, normalize(d_os)};
        TriHit hit;
        Counters& cnt = tl_counters;
        for (size_t k = 0; k + 2 < mesh.verts.size(); k += 3) {
            ++cnt.tris;
            TriHit th = ray_triangle(r, mesh.verts[k], mesh.verts[k + 1], mesh.verts[k + 2], !mesh.double_sided);
            if (!th.hit || th.t < 0.0f) continue;
            if (th.t >= hit.t) continue;
            hit.A = th.A; hit.B = th.B; hit.t = th.t; hit.hit = th.hit; hit.backface = th.backface;
            hit.triangle = static_cast<uint32_t>(k);
            if (ANY) break;
        }
        if (!hit.hit) return hit;
        V3 p = r.origin + r.direction * hit.t;
        V3 p_ws = xyz(mul(od.model, V4(p, 1.0f)));
        hit.t = length(p_ws - ws.origin);
        return hit;
    }
    // DOP::ray_volume_intersection(const ray&), :174-190 (the NaiveAS top level recomputes projections per DOP)
    static VolumeHit dop_test_direct(const Dop& d, const Ray& r) {
        float tmin = std::numeric_limits<float>::lowest(), tmax = std::numeric_limits<float>::max();
        for (int i = 0; i < 7; ++i) {
            float op = dot(r.origin, kDopAxes[i]);
            float ip = 1.0f / dot(r.direction, kDopAxes[i]);
            float t0 = (d.mm[i].x - op) * ip, t1 = (d.mm[i].y - op) * ip;
            if (t0 > t1) std::swap(t0, t1);
            tmin = std::max(t0, tmin);
            tmax = std::min(t1, tmax);
            if (tmax < tmin) return {false, 0.0f, 0.0f};
        }
        return {true, tmin, tmax};
    }
    // BVH_AS::intersect_ray :587-633 / NaiveAS::intersect_ray :257-297
    TriHit intersect(const Ray& ray, bool any_hit) const {
        Counters& cnt = tl_counters;
        ++cnt.rays;
        TriHit hit;
        static thread_local std::vector<VolumeCandidate> cand;
        cand.clear();
        std::array<V2, 7> proj{};
        if (bvh) proj = dop_projections(ray);
        for (uint32_t i = 0; i < objects.size(); ++i) {
            ++cnt.inst_tested;
            VolumeHit vh = bvh ? objects[i].volume.test(proj) : dop_test_direct(objects[i].volume, ray);
            cand.push_back({vh, i});
        }
        if (!any_hit) std::make_heap(cand.begin(), cand.end());
        while (!cand.empty()) {
            const VolumeCandidate& top = (!any_hit) ? cand.front() : cand.back();
            if (top.hit.forward() && top.hit.forward_distance() < hit.t) {
                const ObjectData& od = objects[top.slot];
                ++cnt.inst_entered;
                TriHit mh = bvh ? (any_hit ? trace_bvh<true>(ray, od) : trace_bvh<false>(ray, od))
                                : (any_hit ? trace_naive<true>(ray, od) : trace_naive<false>(ray, od));
                if (mh.forward() && mh.t < hit.t) {
                    hit = mh;
                    hit.object = od.object;
                    hit.mesh = od.mesh;
                    if (any_hit) break;
                }
            } else if (!any_hit) {
                break;
            }
            if (!any_hit) std::pop_heap(cand.begin(), cand.end());
            cand.pop_back();
        }
        return hit;
    }
    BBox scene_bounds() const {  // :710-716
        Dop d;
        for (const ObjectData& o : objects) d.grow(o.volume);
        BBox b;
        b.lo = V3(d.mm[0].x, d.mm[1].x, d.mm[2].x);
        b.hi = V3(d.mm[0].y, d.mm[1].y, d.mm[2].y);
        return b;
    }
};

// ------------------------------------------------------------------------------------------
// shading programs, src/ray_program.cc
// ------------------------------------------------------------------------------------------
constexpr float kEpsilon = 1e-8f;  // ray_program.cc:13
constexpr float kMaxBRDF = 10.0f;  // ray_program.cc:14

struct Sampler {
    int mode = ORC_RNG_COUNTER;
    std::minstd_rand gen;
    std::uniform_real_distribution<float> dist{0.0f, 1.0f};
    float draw(const RayP& r, uint32_t j) { return mode == ORC_RNG_COUNTER ? rng_uniform(r.key, j) : dist(gen); }
};

inline V3 cos_dir(V2 xi, const Mathx& mx) {  // ImportanceSampleCosDir, ray_program.cc:16-27
    float c2 = 1.0f - xi.x;
    float c = std::sqrt(c2), s = std::sqrt(xi.x);
    float phi = 2.0f * xi.y * kPi;
    float sp, cp;
    mx.sincos(phi, &sp, &cp);
    return V3(s * cp, s * sp, c);
}
inline V3 ggx_dir(V2 xi, float a, const Mathx& mx) {  // importanceSampleGGX, ray_program.cc:28-39
    float c2 = (1.0f - xi.x) / (1.0f + (a * a - 1.0f) * xi.x);
    float c = std::sqrt(c2), s = std::sqrt(1.0f - c2);
    float phi = 2.0f * xi.y * kPi;
    float sp, cp;
    mx.sincos(phi, &sp, &cp);
    return V3(s * cp, s * sp, c);
}
inline V3 fresnel_schlick(V3 f0, V3 f90, float c) {  // ray_program.cc:40-45
    float x = 1.0f - c;
    float x2 = x * x;
    float x5 = x2 * x2 * x;
    return mix(f0, f90, x5);
}
inline float v_smith_ggx(float NoV, float NoL, float a) {  // V_SmithGGXCorrelated, ray_program.cc:46-57
    float a2 = a * a;
    float gv = NoL * std::sqrt((-NoV * a2 + NoV) * NoV + a2);
    float gl = NoV * std::sqrt((-NoL * a2 + NoL) * NoL + a2);
    return 2.0f / (gv + gl);
}
struct HitVertices {
    Vertex p1, p2, p3;
};
inline HitVertices fetch_vertices(const Mesh& m, const TriHit& h) {  // get_vertex_data, :75-80
    return {load_vertex(m.vertices[m.indices[h.triangle]]), load_vertex(m.vertices[m.indices[h.triangle + 1]]),
            load_vertex(m.vertices[m.indices[h.triangle + 2]])};
}
inline V4 lin3(V4 a, V4 b, V4 c, float A, float B, float C) { return a * A + b * B + c * C; }
inline V3 lin3(V3 a, V3 b, V3 c, float A, float B, float C) { return a * A + b * B + c * C; }
inline Vertex interpolate(const HitVertices& v, const TriHit& h) {  // interpolate_vertex_data, :81-89
    float C = h.C();
    Vertex r;
    r.position = lin3(v.p1.position, v.p2.position, v.p3.position, h.A, h.B, C);
    r.normal = lin3(v.p1.normal, v.p2.normal, v.p3.normal, h.A, h.B, C);
    r.tangent = lin3(v.p1.tangent, v.p2.tangent, v.p3.tangent, h.A, h.B, C);
    r.uv = v.p1.uv * h.A + v.p2.uv * h.B + v.p3.uv * C;
    return r;
}
inline V3 normals_to_world(const Object& o, Vertex& p) {  // convert_normals_to_world_space, :94-101
    float sign = p.tangent.w;
    p.tangent.w = 0.0f;
    p.tangent = mul(o.model, p.tangent);
    p.normal = xyz(mul(o.normal, V4(p.normal, 0.0f)));
    return cross(p.normal, xyz(p.tangent)) * sign;
}
inline M3 make_tbn(V3 tangent, V3 bitangent, V3 normal) {  // construct_TBN, :110-115
    V3 n = normalize(normal);
    V3 t = normalize(tangent - n * dot(n, tangent));
    V3 b = normalize(bitangent - n * dot(n, bitangent) - t * dot(t, bitangent));
    M3 m;
    m.c[0] = t; m.c[1] = b; m.c[2] = n;
    return m;
}
inline V3 normal_map_to_world(V3 sample, const M3& tbn) {  // normal_map_sample_to_world, :116-120
    V3 n(std::fma(sample.x, 2.0f, -1.0f), std::fma(sample.y, 2.0f, -1.0f), std::fma(sample.z, 2.0f, -1.0f));
    n.y *= -1.0f;
    return mul(tbn, n);
}
inline V3 geometric_normal(const HitVertices& v, bool ds_material, bool backface) {  // :102-109
    V3 g = normalize(cross(v.p2.position - v.p1.position, v.p3.position - v.p1.position));
    if (ds_material && backface) g = g * -1.0f;
    return g;
}
M3 tbn_for_hit(const Object& o, V4 nmap, const Vertex& p, V3 bitangent, V3 v, bool ds_material, bool exiting,
               bool backface, const HitVertices& hv) {  // handle_TBN_creation, :121-142
    M3 tbn = make_tbn(xyz(p.tangent), bitangent, p.normal);
    const bool has_map = (nmap.w != 0.0f);
    V3 n = has_map ? normal_map_to_world(xyz(nmap), tbn) : tbn.c[2];
    const bool impossible = (dot((!exiting) ? v : -v, n) < 0.0);
    if (impossible) {
        n = geometric_normal(hv, ds_material, backface);
        n = xyz(mul(o.normal, V4(n, 0.0f)));
    }
    if (has_map || impossible) tbn = make_tbn(tbn.c[0], tbn.c[1], n);
    return tbn;
}

struct Programs {
    const SceneData* sc = nullptr;
    float env_rot = 0.0f;
    uint32_t ao_samples = 0;
    float inv_ao = 0.0f;
    Mathx mx;

    // RayCasterProgram::on_hit, ray_program.cc:150-163
    V3 raycast(const RayP& ray, const TriHit& hit) const {
        if (!hit.forward()) return xyz(sample_environment(ray.direction, sc->env, env_rot, mx));
        const Mesh& mesh = sc->meshes[hit.mesh];
        HitVertices hv = fetch_vertices(mesh, hit);
        V2 uv = hv.p1.uv * hit.A + hv.p2.uv * hit.B + hv.p3.uv * hit.C();
        return xyz(sample_albedo(*sc, sc->materials[mesh.material], uv));
    }
    // AOProgram::on_hit, ray_program.cc:172-225
    V3 ao(const RayP& ray, const TriHit& hit, std::vector<RayP>& out, Sampler& smp) const {
        if (!hit.forward()) return ray.payload;
        if (ray.depth == 0) return V3(0.0f);
        const Object& obj = sc->objects[hit.object];
        const Mesh& mesh = sc->meshes[hit.mesh];
        HitVertices hv = fetch_vertices(mesh, hit);
        Vertex p = interpolate(hv, hit);
        V3 bitangent = normals_to_world(obj, p);
        const orc_material& mat = sc->materials[mesh.material];
        if (hit.backface && mat.double_sided) p.normal = p.normal * -1.0f;
        M3 tbn = make_tbn(xyz(p.tangent), bitangent, p.normal);
        V4 nmap = sample_normals(*sc, mat, p.uv);
        if (nmap.w != 0.0f) {
            V3 n = normal_map_to_world(xyz(nmap), tbn);
            tbn = make_tbn(tbn.c[0], tbn.c[1], n);
        }
        V3 ws_pos = ray.origin + ray.direction * hit.t;
        V3 new_pos = ws_pos + p.normal * 1e-5f;
        uint8_t new_depth = static_cast<uint8_t>(ray.depth - 1);
        float jx = smp.draw(ray, 0), jy = smp.draw(ray, 1);
        for (uint32_t i = 0; i < ao_samples; ++i) {
            V2 r = fibonacci2d(static_cast<int>(i), inv_ao);
            r.x = std::fmod(r.x + jx, 1.0f);
            r.y = std::fmod(r.y + jy, 1.0f);
            V3 d = mul(tbn, cos_dir(r, mx));
            RayP nr;
            nr.origin = new_pos;
            nr.direction = d;
            nr.payload = V3(inv_ao);
            nr.depth = new_depth;
            nr.any_hit = true;
            nr.key = rng_child(ray.key, LOBE_AO0 + i);
            out.push_back(nr);
        }
        return V3(0.0f);
    }
    // PBRProgram::on_hit, ray_program.cc:234-391
    V3 pbr(const RayP& ray, const TriHit& hit, std::vector<RayP>& out, Sampler& smp) const {
        if (!hit.forward()) return ray.payload * xyz(sample_environment(ray.direction, sc->env, env_rot, mx));
        const Object& obj = sc->objects[hit.object];
        const Mesh& mesh = sc->meshes[hit.mesh];
        HitVertices hv = fetch_vertices(mesh, hit);
        Vertex p = interpolate(hv, hit);
        p.position = xyz(mul(obj.model, V4(p.position, 1.0f)));  // convert_position_to_world_space, :90-93
        V3 bitangent = normals_to_world(obj, p);
        const orc_material& mat = sc->materials[mesh.material];

        bool exiting = false;
        if (hit.backface) {
            if (mat.double_sided) p.normal = p.normal * -1.0f;
            else if (mat.has_volume) exiting = true;
        }
        V4 albedo = sample_albedo(*sc, mat, p.uv);
        float alpha = albedo.w;
        if (!mat.alpha_blending) alpha = (alpha < mat.alpha_cutoff) ? 0.0f : 1.0f;
        if (alpha != 1.0f) {  // pass-through ray, :265-271
            // Not in the reference, shared with the GPU library (ptgpu_stats.rays_abandoned): the new origin is offset along
            // the INTERPOLATED normal, and where that normal and the triangle's plane disagree about the side the ray came
            // from the ray hits the same triangle again, for ever, spawning the other lobes every time.  A pass-through ray
            // carries the number (0..7) of pass-through steps in a row that ended closer than 0.001, a 5-bit tag of the
            // triangle it passed and one bit "already a repeated visit".  The chain ends at the third visit in a row of one
            // triangle, and in any case at the eighth close step in a row (a ray on an edge may alternate between the
            // triangles that share it).  No scene whose normals agree with its geometry gets there.
            const uint32_t h = pass_hash(hit.object, hit.triangle);
            const uint32_t close_steps = ray.pass_tag >> 9;
            const bool near = hit.t < 1e-3f;
            const bool again = near && close_steps && ((ray.pass_tag & 0xffu) == h);
            if (near && ((again && (ray.pass_tag & 0x100u)) || close_steps >= 7u)) {
                ++chains_cut;
            } else {
                RayP nr = ray;
                nr.origin = p.position + (exiting ? 1.0f : -1.0f) * p.normal * 1e-5f;
                nr.payload = nr.payload * (1.0f - alpha);
                nr.key = rng_child(ray.key, LOBE_PASS);
                nr.pass_tag = ((near ? close_steps + 1u : 0u) << 9) | (again ? 0x100u : 0u) | h;
                out.push_back(nr);
            }
        }
        if (alpha == 0.0f) return V3(0.0f);

        V3 att_f(mat.attenuation_factor[0], mat.attenuation_factor[1], mat.attenuation_factor[2]);
        V3 atten(1.0f);
        if (exiting) {  // :278-279
            V3 e = -att_f * hit.t;
            atten = gmin(V3(mx.exp(e.x), mx.exp(e.y), mx.exp(e.z)), V3(1.0f));
        }
        V3 emissive = xyz(sample_emissive(*sc, mat, p.uv));
        if (ray.depth == 0) return ray.payload * emissive * atten * alpha;

        V3 v = -ray.direction;
        V4 nmap = sample_normals(*sc, mat, p.uv);
        M3 tbn = tbn_for_hit(obj, nmap, p, bitangent, v, mat.double_sided != 0, exiting, hit.backface, hv);

        float transmission = sample_transmission(*sc, mat, p.uv);
        V4 orm = sample_roughness_metallic(*sc, mat, p.uv);
        V3 diffuse_color = (1.0f - orm.z) * xyz(albedo);
        V3 f0 = mix(V3(mat.dielectric_f0), xyz(albedo), orm.z);
        const V3 f90(1.0f);
        const float roughness = orm.y;
        const float a_lin = roughness * roughness;
        uint8_t new_depth = static_cast<uint8_t>(ray.depth - 1);

        float u[4];
        for (uint32_t j = 0; j < 4; ++j) u[j] = smp.draw(ray, j);

        if (diffuse_color * (1.0f - transmission) != V3(0.0f)) {  // diffuse lobe, :306-323
            V3 l = normalize(mul(tbn, cos_dir(V2{u[0], u[1]}, mx)));
            V3 h = normalize(v + l);
            float LdH = clampf(dot(l, h), 0.0f, 1.0f);
            V3 F = V3(1.0f) - fresnel_schlick(f0, f90, LdH);
            RayP nr;
            nr.origin = p.position + p.normal * 1e-5f;
            nr.direction = l;
            nr.payload = F * diffuse_color * (1.0f - transmission) * atten * ray.payload * alpha;
            nr.depth = new_depth;
            nr.any_hit = false;
            nr.key = rng_child(ray.key, LOBE_DIFFUSE);
            out.push_back(nr);
        }
        {   // shared micro-normal, :324-389
            V3 m = mul(tbn, ggx_dir(V2{u[2], u[3]}, a_lin, mx));
            V3 N = tbn.c[2];
            V3 offset_dir = p.normal;
            if (exiting) {
                m = m * -1.0f;
                N = N * -1.0f;
                offset_dir = offset_dir * -1.0f;
            }
            const float eta = (!exiting) ? 1.0f / mat.ior : mat.ior;
            V3 l = refract(-v, m, eta);
            const float VdN = clampf(dot(N, v), kEpsilon, 1.0f);
            const float VdM = clampf(dot(v, m), 0.0f, 1.0f);
            const float LdM = clampf(dot(l, -m), 0.0f, 1.0f);
            V3 F = fresnel_schlick(f0, f90, (!exiting) ? VdM : LdM);

            if ((diffuse_color * transmission != V3(0.0f)) && (l != V3(0.0f))) {  // transmission lobe, :349-370
                V3 h = normalize(-(eta * v + l));
                float LdN = clampf(dot(-N, l), 0.0f, 1.0f);
                float G = v_smith_ggx(VdN, LdN, a_lin);
                float NdM = clampf(dot(N, m), kEpsilon, 1.0f);
                float VdH = clampf(dot((!exiting) ? v : -v, h), 0.0f, 1.0f);
                V3 brdf = (V3(1.0f) - F) * (G * VdH * LdN / NdM);
                brdf = gmin(brdf, V3(kMaxBRDF));
                RayP nr;
                nr.origin = p.position - offset_dir * 1e-5f;
                nr.direction = normalize(l);
                nr.payload = diffuse_color * transmission * atten * brdf * ray.payload * alpha;
                nr.depth = new_depth;
                nr.any_hit = false;
                nr.key = rng_child(ray.key, LOBE_TRANSMIT);
                out.push_back(nr);
            }
            {   // specular reflection lobe, :371-388
                V3 h = m;
                l = normalize(reflect(-v, h));
                float LdH = clampf(dot(l, h), 0.0f, 1.0f);
                float LdN = clampf(dot(N, l), 0.0f, 1.0f);
                float NdH = clampf(dot(N, h), kEpsilon, 1.0f);
                float G = v_smith_ggx(VdN, LdN, a_lin);
                V3 brdf = F * (G * LdN * LdH / NdH);
                brdf = gmin(brdf, V3(kMaxBRDF));
                RayP nr;
                nr.origin = p.position + offset_dir * 1e-5f;
                nr.direction = l;
                nr.payload = atten * brdf * ray.payload * alpha;
                nr.depth = new_depth;
                nr.any_hit = false;
                nr.key = rng_child(ray.key, LOBE_REFLECT);
                out.push_back(nr);
            }
        }
        return ray.payload * emissive * atten * alpha;
    }
    // Single-path PBR (no counterpart in the reference, whose estimator follows every lobe): of the rays one hit
    // would spawn, one is kept, chosen with probability proportional to the largest component of its payload, and
    // its payload is divided by that probability — the same expectation with at most one ray per hit, which is
    // what makes deep bounce limits affordable.  The BRDF clamp is already inside the payloads when they are
    // weighed.  Order of the candidates: pass-through, diffuse, transmission, reflection.
    static float lobe_weight(V3 p) {
        float w = std::fabs(p.x);
        if (std::fabs(p.y) > w) w = std::fabs(p.y);
        if (std::fabs(p.z) > w) w = std::fabs(p.z);
        return ((w > 0.0f) && (w < kInf)) ? w : 0.0f;  // NaN, infinite and zero payloads are never chosen
    }
    V3 pbr_single(const RayP& ray, const TriHit& hit, std::vector<RayP>& out, Sampler& smp) const {
        static thread_local std::vector<RayP> cand;
        cand.clear();
        V3 radiance = pbr(ray, hit, cand, smp);
        if (cand.empty()) return radiance;
        float w[4] = {0.0f, 0.0f, 0.0f, 0.0f}, c[4];
        float total = 0.0f;
        for (size_t k = 0; k < cand.size(); ++k) {
            w[k] = lobe_weight(cand[k].payload);
            total = total + w[k];
            c[k] = total;
        }
        if (!(total > 0.0f)) return radiance;  // nothing the children could add
        const float x = smp.draw(ray, 4) * total;
        size_t pick = cand.size() - 1;
        for (size_t k = cand.size(); k-- > 0;)
            if (x < c[k]) pick = k;
        while (w[pick] == 0.0f) --pick;  // only when x landed on the total by rounding: the last candidate with weight
        RayP nr = cand[pick];
        nr.payload = nr.payload * (total / w[pick]);
        out.push_back(nr);
        return radiance;
    }
    bool single_path = false;
    mutable std::atomic<uint64_t> chains_cut{0};  // pass-through chains that did not advance (see pbr())
    V3 on_hit(int mode, const RayP& ray, const TriHit& hit, std::vector<RayP>& out, Sampler& smp) const {
        switch (mode) {
            case 1: return ao(ray, hit, out, smp);
            case 2: return single_path ? pbr_single(ray, hit, out, smp) : pbr(ray, hit, out, smp);
            default: return raycast(ray, hit);
        }
    }
};

// ------------------------------------------------------------------------------------------
// frame loop, src/renderer.cc
// ------------------------------------------------------------------------------------------
std::vector<V2> subsample_positions(uint32_t spp) {  // Renderer::generate_subsample_positions, :135-155
    std::vector<V2> pos(spp);
    int s = static_cast<int>(sqrtf(static_cast<float>(spp)));
    if (static_cast<uint32_t>(s * s) == spp) {
        for (uint32_t i = 0; i < spp; ++i)
            pos[i] = V2{(static_cast<float>(i / s) + 0.5f) / static_cast<float>(s),
                        (static_cast<float>(i % s) + 0.5f) / static_cast<float>(s)};
    } else {
        float inv = 1.0f / static_cast<float>(spp);
        for (uint32_t i = 0; i < spp; ++i) pos[i] = fibonacci2d(static_cast<int>(i), inv);
    }
    return pos;
}

}  // namespace

struct orc_scene {
    SceneData data;
    Accel accel;
    Programs programs;
    orc_settings settings{};
    int rng_mode = ORC_RNG_COUNTER;
    M4 ndc2world;
    V3 origin;
    bool camera_set = false;
};

namespace {

inline V3 div_scalar(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }

RayP camera_ray(const orc_scene& s, int x, int y, float inv_w, float inv_h, V2 sub, uint32_t pixel, uint32_t sample) {
    // ndc_from_pixel + Renderer::generate_camera_ray, renderer.cc:17-21, 66-75
    V2 pc = V2{static_cast<float>(x), static_cast<float>(y)} + sub;
    float nx = pc.x * inv_w * 2.0f - 1.0f;
    float ny = -pc.y * inv_h * 2.0f + 1.0f;
    V3 d = xyz(mul(s.ndc2world, V4(nx, ny, 0.0f, 1.0f)));
    RayP r;
    r.origin = s.origin;
    r.direction = normalize(d);
    r.payload = V3(1.0f);
    r.depth = static_cast<uint8_t>(s.settings.max_ray_bounces);
    r.any_hit = false;
    r.key = rng_camera_key(s.settings.seed, pixel, sample);
    return r;
}

enum class Out { Mean, Sum };

int render_rows(orc_scene* s, int W, int H, int y0, int y1, float* out, int n_threads, orc_stats* stats, Out mode) {
    if (!s || !out || W <= 0 || H <= 0 || !s->camera_set) return -1;
    const orc_settings st = s->settings;
    const uint32_t spp = st.samples_per_pixel ? st.samples_per_pixel : 1;
    const uint32_t step = st.sample_step ? st.sample_step : 1;
    const uint32_t begin = st.sample_begin;
    const std::vector<V2> sub = subsample_positions(spp);
    const float inv_w = 1.0f / static_cast<float>(W), inv_h = 1.0f / static_cast<float>(H);
    const int mode_prog = st.program_mode;
    if (n_threads <= 0) n_threads = static_cast<int>(std::thread::hardware_concurrency());
    if (n_threads <= 0) n_threads = 1;
    n_threads = std::min(n_threads, std::max(1, y1 - y0));

    std::atomic<int> next_row{y0};
    s->programs.chains_cut = 0;
    std::vector<Counters> per_thread(n_threads);
    auto t_start = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        tl_counters = Counters{};
        std::vector<RayP> rays;  // LIFO, renderer.cc:100
        Sampler smp;
        smp.mode = s->rng_mode;
        for (;;) {
            int y = next_row.fetch_add(1);
            if (y >= y1) break;
            if (smp.mode == ORC_RNG_MINSTD) smp.gen.seed(st.seed * 7919u + static_cast<uint32_t>(y) + 1u);
            for (int x = 0; x < W; ++x) {
                V3 mean(0.0f), sum(0.0f);
                float count = 0.0f;
                uint32_t n_samples = 0;
                for (uint32_t i = begin; i < spp; i += step) {
                    V3 col(0.0f);
                    rays.push_back(camera_ray(*s, x, y, inv_w, inv_h, sub[i], static_cast<uint32_t>(y) * W + x, i));
                    while (!rays.empty()) {  // renderer.cc:111-117
                        RayP r = rays.back();
                        rays.pop_back();
                        TriHit h = s->accel.intersect(r, r.any_hit);
                        col = col + s->programs.on_hit(mode_prog, r, h, rays, smp);
                    }
                    // Welford mean, renderer.h:24-30
                    count += 1.0f;
                    V3 delta = col - mean;
                    mean = mean + div_scalar(delta, count);
                    sum = sum + col;
                    ++n_samples;
                }
                float* px = out + (static_cast<size_t>(y) * W + x) * 4;
                if (mode == Out::Mean) {
                    px[0] = mean.x; px[1] = mean.y; px[2] = mean.z; px[3] = 1.0f;  // renderer.cc:120
                } else {
                    px[0] = sum.x; px[1] = sum.y; px[2] = sum.z; px[3] = static_cast<float>(n_samples);
                }
            }
        }
        per_thread[tid] = tl_counters;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& t : pool) t.join();
    if (stats) {
        orc_stats r{};
        for (const Counters& c : per_thread) {
            r.rays += c.rays; r.nodes_visited += c.nodes; r.triangles_tested += c.tris;
            r.instances_tested += c.inst_tested; r.instances_entered += c.inst_entered;
        }
        uint64_t per_px = 0;
        for (uint32_t i = begin; i < spp; i += step) ++per_px;
        r.camera_samples = per_px * static_cast<uint64_t>(W) * static_cast<uint64_t>(y1 - y0);
        r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        r.chains_cut = s->programs.chains_cut.load();
        *stats = r;
    }
    return 0;
}

void fill_hit(const TriHit& h, orc_hit* o) {
    o->t = h.forward() ? h.t : kInf;
    o->A = h.forward() ? h.A : 0.0f;
    o->B = h.forward() ? h.B : 0.0f;
    o->object_index = h.forward() ? h.object : 0;
    o->mesh_index = h.forward() ? h.mesh : 0;
    o->triangle_index = h.forward() ? h.triangle : 0;
    o->hit = h.forward() ? 1u : 0u;
    o->backface = (h.forward() && h.backface) ? 1u : 0u;
}
TriHit to_trihit(const orc_hit& h) {
    TriHit t;
    t.hit = h.hit != 0;
    t.t = h.hit ? h.t : kInf;
    t.A = h.A; t.B = h.B;
    t.object = h.object_index; t.mesh = h.mesh_index; t.triangle = h.triangle_index;
    t.backface = h.backface != 0;
    return t;
}
V3 v3(const float* p) { return V3(p[0], p[1], p[2]); }
void store3(V3 v, float* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

}  // namespace

extern "C" {

orc_scene* orc_scene_create(const orc_model* model, const orc_envmap* env, const orc_settings* settings) {
    if (!model || !settings) return nullptr;
    auto* s = new orc_scene;
    SceneData& d = s->data;
    d.vertex_store.assign(model->vertices, model->vertices + model->n_vertices);
    d.index_store.assign(model->indices, model->indices + model->n_indices);
    if (model->n_texels) d.texel_store.assign(model->texels, model->texels + model->n_texels * 4);
    for (uint32_t i = 0; i < model->n_meshes; ++i) {
        const orc_mesh& m = model->meshes[i];
        d.meshes.push_back({d.index_store.data() + m.first_index, m.n_indices, d.vertex_store.data() + m.first_vertex,
                            m.n_vertices, m.material_index});
    }
    for (uint32_t i = 0; i < model->n_objects; ++i) {
        const orc_object& o = model->objects[i];
        d.objects.push_back({load_m4(o.model_matrix), load_m4(o.normal_matrix), o.mesh_index});
    }
    d.materials.assign(model->materials, model->materials + model->n_materials);
    for (uint32_t i = 0; i < model->n_images; ++i) {
        const orc_image& im = model->images[i];
        d.images.push_back({d.texel_store.data() + im.texel_offset * 4, im.width, im.height});
    }
    if (env && env->rgba && env->width > 0 && env->height > 0) {
        d.env_store.assign(env->rgba, env->rgba + static_cast<size_t>(env->width) * env->height * 4);
        d.env = {d.env_store.data(), env->width, env->height};
    } else {  // CPUTexture<hdr_pixel>::create_white_texture, cpu_framebuffer.cc:104
        d.env_store.assign(4, 1.0f);
        d.env = {d.env_store.data(), 1, 1};
    }
    s->settings = *settings;
    s->accel.build(d, settings->accel_struct_type == 1,
                   settings->max_triangles_per_bvh_leaf ? settings->max_triangles_per_bvh_leaf : 1);
    s->programs.sc = &s->data;
    s->programs.env_rot = settings->envmap_rotation;
    s->programs.ao_samples = settings->max_new_rays_per_bounce;
    s->programs.inv_ao = 1.0f / static_cast<float>(settings->max_new_rays_per_bounce);  // ray_program.cc:167
    s->programs.single_path = (settings->flags & ORC_FLAG_SINGLE_PATH) != 0;
    return s;
}
void orc_scene_destroy(orc_scene* s) { delete s; }
void orc_set_settings(orc_scene* s, const orc_settings* st) {
    s->settings.samples_per_pixel = st->samples_per_pixel;
    s->settings.max_ray_bounces = st->max_ray_bounces;
    s->settings.seed = st->seed;
    s->settings.sample_begin = st->sample_begin;
    s->settings.sample_step = st->sample_step;
    s->settings.flags = st->flags;
    s->programs.single_path = (st->flags & ORC_FLAG_SINGLE_PATH) != 0;
}
void orc_set_rng_mode(orc_scene* s, int m) { s->rng_mode = m; }
void orc_set_math_mode(orc_scene* s, int m) { s->programs.mx.mode = m; }

void orc_update_camera_transform_state(orc_scene* s, const float position[3], const float direction[3],
                                       const float up[3], float yfov, float aspect, float znear, float zfar) {
    // Renderer::update_camera_transform_state, renderer.cc:27-39
    V3 pos = v3(position), dir = v3(direction);
    s->origin = pos;
    M4 view = look_at_rh(pos, pos + dir, v3(up));
    M4 proj = perspective_rh(yfov, aspect > 0.0f ? aspect : 1.77777777777777777f, znear, zfar > 0.0f ? zfar : 1000.f);
    view.c[3] = V4(0.0f, 0.0f, 0.0f, 1.0f);
    s->ndc2world = inverse(mul(proj, view));
    s->camera_set = true;
}
void orc_set_camera_raw(orc_scene* s, const float ndc2world[16], const float origin[3]) {
    s->ndc2world = load_m4(ndc2world);
    s->origin = v3(origin);
    s->camera_set = true;
}
void orc_get_camera_raw(const orc_scene* s, float ndc2world[16], float origin[3]) {
    std::memcpy(ndc2world, &s->ndc2world, 64);
    store3(s->origin, origin);
}
void orc_get_scene_bound(const orc_scene* s, float bmin[3], float bmax[3]) {
    BBox b = s->accel.scene_bounds();
    store3(b.lo, bmin);
    store3(b.hi, bmax);
}
int orc_render_frame(orc_scene* s, int w, int h, float* out, int n_threads, orc_stats* stats) {
    return render_rows(s, w, h, 0, h, out, n_threads, stats, Out::Mean);
}
int orc_render_frame_sums(orc_scene* s, int w, int h, float* out, int n_threads, orc_stats* stats) {
    return render_rows(s, w, h, 0, h, out, n_threads, stats, Out::Sum);
}
int orc_render_rows(orc_scene* s, int w, int h, int y0, int y1, float* out, int n_threads, orc_stats* stats) {
    if (y0 < 0 || y1 > h || y0 >= y1) return -1;
    return render_rows(s, w, h, y0, y1, out, n_threads, stats, Out::Mean);
}
int orc_render_hits(orc_scene* s, int W, int H, uint32_t sample, orc_hit* out, int n_threads) {
    if (!s || !out || !s->camera_set) return -1;
    const uint32_t spp = s->settings.samples_per_pixel ? s->settings.samples_per_pixel : 1;
    if (sample >= spp) return -1;
    const std::vector<V2> sub = subsample_positions(spp);
    const float inv_w = 1.0f / static_cast<float>(W), inv_h = 1.0f / static_cast<float>(H);
    if (n_threads <= 0) n_threads = static_cast<int>(std::thread::hardware_concurrency());
    n_threads = std::max(1, std::min(n_threads, H));
    std::atomic<int> next_row{0};
    auto worker = [&]() {
        for (;;) {
            int y = next_row.fetch_add(1);
            if (y >= H) break;
            for (int x = 0; x < W; ++x) {
                RayP r = camera_ray(*s, x, y, inv_w, inv_h, sub[sample], static_cast<uint32_t>(y) * W + x, sample);
                fill_hit(s->accel.intersect(r, false), out + static_cast<size_t>(y) * W + x);
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    return 0;
}
int orc_intersect_rays(orc_scene* s, uint64_t n, const float* o, const float* d, const uint8_t* any, orc_hit* out) {
    if (!s || !out) return -1;
    for (uint64_t i = 0; i < n; ++i) {
        Ray r{v3(o + 3 * i), v3(d + 3 * i)};
        bool a = any && any[i];
        TriHit h = s->accel.intersect(r, a);
        fill_hit(h, out + i);
    }
    return 0;
}
int orc_get_bvh_info(const orc_scene* s, uint32_t mesh, orc_bvh_info* out) {
    if (!s || !s->accel.bvh || mesh >= s->accel.bvh_meshes.size()) return -1;
    s->accel.bvh_meshes[mesh].info(out);
    return 0;
}
void orc_subsample_positions(uint32_t spp, float* out_xy) {
    std::vector<V2> p = subsample_positions(spp);
    for (uint32_t i = 0; i < spp; ++i) {
        out_xy[2 * i] = p[i].x;
        out_xy[2 * i + 1] = p[i].y;
    }
}

// ---- known-answer entry points ---------------------------------------------------------------
void orc_kat_ray_triangle(const float o[3], const float d[3], const float p1[3], const float p2[3], const float p3[3],
                          int cull, float out[5]) {
    TriHit h = ray_triangle(Ray{v3(o), v3(d)}, v3(p1), v3(p2), v3(p3), cull != 0);
    out[0] = h.A; out[1] = h.B; out[2] = h.t; out[3] = h.hit ? 1.0f : 0.0f; out[4] = h.backface ? 1.0f : 0.0f;
}
void orc_kat_bbox(const float bmin[3], const float bmax[3], const float o[3], const float d[3], float out[4]) {
    BBox b;
    b.lo = v3(bmin);
    b.hi = v3(bmax);
    Ray r{v3(o), v3(d)};
    VolumeHit h = b.test(r, 1.0f / r.direction);
    out[0] = h.hit ? 1.0f : 0.0f; out[1] = h.t0; out[2] = h.t1; out[3] = h.forward_distance();
}
void orc_kat_dop(const float* pts, int n, const float o[3], const float d[3], float out[4]) {
    Dop dop;
    for (int i = 0; i < n; ++i) dop.grow(v3(pts + 3 * i));
    Ray r{v3(o), v3(d)};
    VolumeHit h = dop.test(dop_projections(r));
    out[0] = h.hit ? 1.0f : 0.0f; out[1] = h.t0; out[2] = h.t1; out[3] = h.forward_distance();
}
void orc_kat_sample_bilinear_rgba8(const uint8_t* texels, int w, int h, float u, float v, int srgb, float out[4]) {
    V4 c = sample_bilinear(Tex8{texels, w, h}, V2{u, v}, srgb != 0);
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}
void orc_kat_sample_environment(const float* rgba, int w, int h, const float dir[3], float rot, int math_mode, float out[4]) {
    Mathx mx;
    mx.mode = math_mode;
    V4 c = sample_environment(v3(dir), TexF{rgba, w, h}, rot, mx);
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}
void orc_kat_cos_dir(float x, float y, int math_mode, float out[3]) {
    Mathx mx;
    mx.mode = math_mode;
    store3(cos_dir(V2{x, y}, mx), out);
}
void orc_kat_ggx(float x, float y, float a, int math_mode, float out[3]) {
    Mathx mx;
    mx.mode = math_mode;
    store3(ggx_dir(V2{x, y}, a, mx), out);
}
void orc_kat_schlick(const float f0[3], const float f90[3], float c, float out[3]) { store3(fresnel_schlick(v3(f0), v3(f90), c), out); }
float orc_kat_v_smith(float NoV, float NoL, float a) { return v_smith_ggx(NoV, NoL, a); }
void orc_kat_refract(const float I[3], const float N[3], float eta, float out[3]) { store3(refract(v3(I), v3(N), eta), out); }
void orc_kat_fibonacci2d(int i, float inv_n, float out[2]) {
    V2 p = fibonacci2d(i, inv_n);
    out[0] = p.x; out[1] = p.y;
}
float orc_kat_srgb_to_linear(float c) { return srgb_to_linear(c); }
float orc_kat_linear_to_srgb(float c) { return linear_to_srgb(c); }
void orc_kat_float_pixel_to_srgb8(const float rgba[4], uint8_t out[4]) {  // float_pixel_to_srgb8, cpu_framebuffer.h:52-59
    for (int i = 0; i < 4; ++i) {
        float c = clampf(rgba[i], 0.0f, 1.0f);
        out[i] = static_cast<uint8_t>(linear_to_srgb(c) * 255 + 0.5f);
    }
}
uint32_t orc_kat_rng_camera_key(uint32_t seed, uint32_t pixel, uint32_t sample) { return rng_camera_key(seed, pixel, sample); }
float orc_kat_rng_uniform(uint32_t key, uint32_t j) { return rng_uniform(key, j); }
uint32_t orc_kat_rng_child(uint32_t key, uint32_t lobe) { return rng_child(key, lobe); }

int orc_kat_on_hit(orc_scene* s, const orc_ray* ray, const orc_hit* hit, float radiance[3], orc_ray* out_rays, int max_out) {
    RayP r;
    r.origin = v3(ray->origin);
    r.direction = v3(ray->direction);
    r.payload = v3(ray->payload);
    r.depth = static_cast<uint8_t>(ray->depth);
    r.any_hit = ray->any_hit != 0;
    r.key = ray->key;
    std::vector<RayP> out;
    Sampler smp;
    smp.mode = ORC_RNG_COUNTER;
    V3 c = s->programs.on_hit(s->settings.program_mode, r, to_trihit(*hit), out, smp);
    store3(c, radiance);
    int n = static_cast<int>(out.size());
    for (int i = 0; i < n && i < max_out; ++i) {
        store3(out[i].origin, out_rays[i].origin);
        store3(out[i].direction, out_rays[i].direction);
        store3(out[i].payload, out_rays[i].payload);
        out_rays[i].depth = out[i].depth;
        out_rays[i].any_hit = out[i].any_hit ? 1u : 0u;
        out_rays[i].key = out[i].key;
    }
    return n;
}

}  // extern "C"
