// pt_scene_build.h — everything ptgpu_load_scene does on the host, as plain functions over plain arrays: validation of
// the model, the two-level acceleration structure (pt_bvh.h), the numbering of the most visited nodes, the tree over the
// instance bounds, the collapse of the binary trees into four-wide quantised nodes, and the shading tables.  No HIP call
// in here (the device-side tree builder is handed in as a callback), so the whole of it runs under the host sanitizers.
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "pt_bvh.h"

namespace pth {

struct BuildOptions {
    uint32_t max_leaf = 8;            // maxTrianglesPerBVHLeaf (four-wide nodes: capped at 2)
    bool naive = false;               // PTGPU_AS_NAIVE: one leaf per mesh
    bool want_wide = false;           // PTGPU_FLAG_WIDE_NODES (ignored with `naive`)
    bool reference_splits = false;    // keep the reference's splits under the four-wide nodes
    bool greedy_collapse = false;     // four-wide collapse: always pull up the child with the largest box
    bool linear_tlas = false;         // no tree over the instance bounds whatever their number
    uint32_t traverse_waves = 0;      // 0: by the size of the geometry; 6 or 7: that form of the four-wide kernel
    uint32_t leaf_vote = 0;           // 0: by scene size and program; 1..16
    int leaf_records = 0;             // four-wide leaves as aligned 64-byte records: 0 by the size of the geometry, 1 always, -1 never
    uint32_t build_threads = 0;       // 0: up to 16
    int program_mode = PTGPU_PROGRAM_PBR;
    bool verbose = false;
    // the device-side tree builder (PTGPU_AS_LBVH), or empty: host binned-SAH builder
    std::function<bool(const ptgpu_vertex*, const uint32_t*, uint32_t, uint32_t, MeshTree&, std::string*)> mesh_builder;
};

// What a scene is on the device, still in host memory: one array per device buffer plus the scalars of SceneDev.
struct BuiltScene {
    std::vector<ptd::NodeQ> nodes;
    std::vector<ptd::TriQ> tris;           // + two padding records (the first is the dummy leaf of unused wide slots)
    std::vector<ptd::Node4> nodes4;        // empty without four-wide nodes
    std::vector<ptd::Leaf64> leaves64;     // four-wide trees' leaves as 64-byte records (then nodes4 refers to these), or empty
    std::vector<ptd::TlasNode> tlas;       // empty: every ray tests every instance bound
    std::vector<ptd::DevInstance> instances;
    std::vector<ptd::DevXform> xforms;
    std::vector<ptd::DevObject> objects;
    std::vector<ptgpu_vertex> corners;     // three de-indexed vertices per triangle, mesh order
    std::vector<uint32_t> object_mesh;
    std::vector<ptgpu_bvh_info> bvh_info;  // per mesh (of the binary tree)
    float srgb_lut[256];
    uint32_t n_top = 0, n_top4 = 0;
    uint32_t max_depth = 0;                // stack entries a traversal can need
    uint32_t max_depth4 = 0;               // depth of the deepest four-wide tree (0 without)
    bool seven_waves = false;
    uint32_t leaf_vote = 1;
    float scene_min[3] = {0, 0, 0}, scene_max[3] = {0, 0, 0};
};

// Everything the kernels index with is inside its array, every coordinate and matrix entry is finite and small enough
// that box areas do not overflow.  Returns false with a message.
bool validate_model(const ptgpu_model& model, std::string* err);

// The whole host side of Renderer::load_scene (renderer.cc:41-64; BVH_AS::BVH_AS, acceleration_structure.cc:352-390).
// `model` must have passed validate_model.  Returns false with a message (and PTGPU_E_* in *code) on failure.
bool build_scene(const ptgpu_model& model, const BuildOptions& opt, BuiltScene& out, std::string* err, int* code);

// Number of build_scene calls this process has completed (a scene shared by several device contexts is built once).
uint64_t host_build_count();

// ---- the steps, exposed for the sanitizer driver and the CPU tests ----
// Greedy choice of the `capacity` nodes expected to be visited most (half-area x the mesh's summed squared instance scale),
// renumbered to the front of the array; returns how many were chosen.  `roots` are updated.
uint32_t number_top_nodes(std::vector<ptd::DevNode>& nodes, std::vector<uint32_t>& mesh_root, const std::vector<float>& mesh_weight,
                          std::vector<ptd::DevXform>& xforms, uint32_t capacity);
// Count-balanced threaded tree over the instances' boxes (depth first, skip links).
void build_instance_tree(const std::vector<ptd::DevInstance>& instances, std::vector<ptd::TlasNode>& out);
// Binary trees -> four-wide quantised nodes.  `binary_root[m]` = root reference of mesh m (PT_NO_NODE: no instance).
// `dummy_leaf` = reference of the padding triangle record that fills unused child slots.  Returns false if the recorded
// decisions are inconsistent (they cannot be for finite boxes; the check is what keeps bad input from corrupting memory).
bool collapse_to_wide(const std::vector<ptd::DevNode>& nodes, const std::vector<uint32_t>& binary_root, uint32_t dummy_leaf,
                      bool greedy, std::vector<ptd::Node4>& wide, std::vector<uint32_t>& mesh_root4, uint32_t* max_depth4,
                      std::string* err);
// The leaves of the four-wide trees (runs of one or two 48-byte triangle records) as 64-byte records; the leaf references
// of `wide` and `mesh_root4` are rewritten to record numbers.  `tris` = binary-order triangle records with leaf counts,
// `n_real` of them real (the dummy leaf at tris[n_real] becomes a record of its own).
void make_leaf_records(const std::vector<ptd::DevTri>& tris, uint32_t n_real, std::vector<ptd::Node4>& wide, std::vector<uint32_t>& mesh_root4,
                       std::vector<ptd::Leaf64>& out);
uint32_t number_top_nodes4(std::vector<ptd::Node4>& wide, std::vector<uint32_t>& mesh_root4, const std::vector<float>& mesh_weight,
                           uint32_t capacity);

}  // namespace pth
