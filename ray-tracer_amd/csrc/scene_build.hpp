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

// TriangleMesh.computeNormals, no-normals branch.
void synthesise_normals(const float *vertices, uint32_t n_vertices, const int32_t *indices, uint32_t n_faces, float *out);

}   // namespace mi355rt
