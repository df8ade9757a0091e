// scene_build.hpp -- host-side construction of one mesh object's flattened hierarchy.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "layout.h"

namespace mi355rt {

struct BuildOptions {
   int max_depth = 20;
   int max_children = 5;
   bool use_kdtree = true;
   bool build_tree = true;   // false: validate, normals and bounds only (the hierarchy comes from gpu_build.hip)
   bool synth_normals = true;   // false: a mesh without normals keeps none (gpu_build.hip makes them on the device)
};

struct BuiltObject {
   std::vector<NodeRec> nodes;        // internal nodes, DFS pre-order
   std::vector<int32_t> slot_face;    // face id of every leaf slot, leaves in DFS order
   std::vector<int32_t> slot_remaining;   // faces left in the slot's leaf, this slot included
   std::vector<float> normals;        // per-vertex normals actually used (given or synthesised)
   float bmin[3], bmax[3];            // KDTree.minMax
   float shadow_distance = 0.f;       // KDTree.shadowDistance
   int32_t root_link = 0, root_cnt = 0, root_is_leaf = 1;
   uint64_t n_leaves = 0;
   int32_t max_depth = 0;
};

// Builds the reference's hierarchy (bounds/KDTree.java:24-56, bounds/KDNode.java:29-90) for one mesh.
// `normals` may be null (synthesised per sceneObjects/TriangleMesh.java:96-126).
// Returns an empty string on success, else an error message.
std::string build_object(const float *vertices, uint32_t n_vertices, const float *normals, const int32_t *indices,
                         uint32_t n_faces, const BuildOptions &opts, BuiltObject &out);

// ---- leaf groups (layout.h "LEAVES"): what the walker reads instead of the raw leaf slots.
// The reference's builder copies a face into every leaf whose box it straddles (6.9 leaves per face on the Dragon-class
// mesh), so a ray that walks a handful of neighbouring leaves meets the same faces again and again -- 62 % of the
// walker's triangle tests repeated one of the ray's last 32 faces.  A test that fails, fails in every leaf, so it need
// only run once per ray.  The hierarchy itself must stay the reference's (which leaves a ray reaches decides its
// answer); what changes is how a leaf names its faces: small subtrees are made GROUPS of at most GROUP_FACES distinct
// faces, the group's faces are stored once, and each leaf of the group is a bit mask over them.  A lane keeps one
// mask of the group faces its ray has already missed and skips them (trace_kernels.hip).
struct GroupedLeaves {
   std::vector<LeafDesc> descs;        // leaves in DFS order (a leaf of n > GROUP_FACES faces: ceil(n / GROUP_FACES) consecutive descriptors)
   std::vector<int32_t> gface;         // face id of every group-face entry
   uint64_t n_groups = 0, n_shared = 0;   // groups in all; groups that span more than one leaf
};

// Rewrites the leaf links of the records (first slot -> index of the leaf's first descriptor; node links and counts
// stay) and fills `out`.  records: n_records NodeRec or DevNodeRec (`stride` bytes each) in DFS pre-order, links relative to this object;
// slot_face: face id per leaf slot.  root_link / root_cnt: the root (node 0, or a leaf root's first slot and face count;
// a leaf root is rewritten too).
void group_leaves(void *records, size_t n_records, size_t stride, const int32_t *slot_face, int32_t &root_link, int32_t root_cnt, GroupedLeaves &out);

// TriangleMesh.computeNormals, no-normals branch.
void synthesise_normals(const float *vertices, uint32_t n_vertices, const int32_t *indices, uint32_t n_faces, float *out);

}   // namespace mi355rt
