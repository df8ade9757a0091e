// gpu_build.hpp -- the reference's spatial-median hierarchy (bounds/KDNode.java:29-90) built on the GPU, level by
// level, into exactly the arrays scene_build.cpp produces on the host.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>

#include "layout.h"
#include "scene_build.hpp"

namespace mi355rt {

struct GpuBuiltObject {
   DevNodeRec *nodes = nullptr;   // n_nodes records in DFS pre-order; links relative to this object (node index / leaf first slot)
   int32_t *slot_ids = nullptr;   // face id per leaf slot, leaves in DFS order, faces in the reference's stored order
   uint64_t n_nodes = 0, n_leaves = 0, n_slots = 0;
   int32_t max_depth = 0;
};

// Preconditions (checked by the caller): use_kdtree, n_faces > max_children, max_depth >= 1, every vertex finite,
// indices in range.  bmin / bmax = the root box over ALL vertices (bounds/KDTree.java:35-44).  On success `out`
// owns two device buffers (release with gpu_build_free).
hipError_t gpu_build_object(const float *vertices, uint32_t n_vertices, const int32_t *indices, uint32_t n_faces, const float bmin[3],
                            const float bmax[3], const BuildOptions &opts, GpuBuiltObject &out, std::string &err);
void gpu_build_free(GpuBuiltObject &o);

// Gathers the corners of the n faces listed in `gface` (host) into 48-byte group-face entries at dst_dev (layout.h "gfaces").
hipError_t gpu_emit_gfaces(const float *vertices, uint32_t n_vertices, const int32_t *indices, uint32_t n_faces, const int32_t *gface, size_t n,
                           float *dst_dev);

}   // namespace mi355rt
