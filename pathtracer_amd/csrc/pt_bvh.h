// pt_bvh.h — host-side construction of the two-level acceleration structure and its flattening into
// the device layout of pt_device.h.
//
// The tree is the reference's: a top-down binned-SAH binary BVH per mesh (32 bins x 3 axes on the
// triangle centroid, midpoint / median fallbacks; BVH_AS::MeshBVHData, acceleration_structure.cc:392-536)
// and a world-space 14-DOP per object (object_to_ws_dop, :208-220), objects sorted by triangle count,
// descending (:370-374).  Only the storage differs: an inner node stores both children's boxes and
// references (64 B), leaves are contiguous runs of 48-byte triangles.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "ptgpu.h"
#include "pt_device.h"

namespace pth {

struct MeshTree {
    std::vector<ptd::DevNode> nodes;  // child references are local to this mesh until relocate()
    std::vector<ptd::DevTri> tris;
    uint32_t root = 0;                // reference of the root (leaf bit set when the whole mesh is one leaf)
    ptgpu_bvh_info info{};
};

// Build the BVH of one mesh.  `max_leaf` = maxTrianglesPerBVHLeaf; `single_leaf` puts every triangle
// into one leaf in mesh order (the brute-force NaiveAS of acceleration_structure.cc:299-340).
// `tuned_splits` leaves the reference's tree for one of lower expected cost (bins over the centroid range, exact
// right-hand sums, every split position tried in small ranges): for the four-wide nodes, which do not keep the
// reference's visiting order anyway.  Same closest hits, up to ties in distance.
// `build_threads`: host threads that build independent subtrees of a large mesh (0 = one per core, at most 16); the tree
// does not depend on it.
void build_mesh_tree(const ptgpu_vertex* vertices, const uint32_t* indices, uint32_t n_indices, uint32_t max_leaf,
                     bool single_leaf, bool tuned_splits, MeshTree& out, unsigned build_threads = 0);

// Device-side alternative (pt_lbvh.hip): a linear BVH over Morton-ordered triangles, built on the GPU of the current
// device and read back into the same structure.  Different tree, same closest hits (up to ties in distance).
// Returns false with a message in *err when a HIP call fails.
bool build_mesh_tree_lbvh(const ptgpu_vertex* vertices, const uint32_t* indices, uint32_t n_indices, uint32_t max_leaf, hipStream_t st,
                          MeshTree& out, std::string* err);

// Statistics of a finished tree (collect_tree_info, acceleration_structure.cc:538-570) into t.info;
// n_real_tris = 0 for the placeholder leaf of an empty mesh.
void collect_info(MeshTree& t, uint32_t n_real_tris);

// Add base offsets to every reference so the tree can live at nodes[node_base...] / tris[tri_base...].
void relocate(MeshTree& t, uint32_t node_base, uint32_t tri_base);

struct InstanceTable {
    std::vector<ptd::DevInstance> instances;  // sorted by complexity, descending
    std::vector<ptd::DevXform> xforms;        // same order; root/cull filled by the caller
    float scene_min[3], scene_max[3];         // get_scene_bounds (:710-716)
};
void build_instances(const ptgpu_model& model, InstanceTable& out);

}  // namespace pth
