// gpu_build.hpp -- the reference's spatial-median hierarchy (bounds/KDNode.java:29-90) built on the GPU, level by
// level, into exactly the arrays scene_build.cpp produces on the host.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>

#include "layout.h"
#include "scene_build.hpp"

namespace mi355rt {

// where one object's records go in the scene's concatenated arrays (layout.h: links are absolute)
struct PartBases {
   size_t node = 0, slot = 0, desc = 0, gface = 0;
};

struct GpuBuiltObject {
   // everything below is device memory in the final, uploaded form (absolute links: the bases are folded in)
   DevNodeRec *nodes = nullptr;   // n_nodes records in DFS pre-order
   int32_t *slot_ids = nullptr;   // face id per leaf slot, leaves in DFS order, faces in the reference's stored order
   LeafDesc *descs = nullptr;     // n_descs leaf descriptors (scene_build.hpp "leaf groups", grouped on the device)
   float *gfaces = nullptr;       // 12 floats per group face
   float *fnrm = nullptr;         // 12 floats per face: its three vertex normals
   uint64_t n_nodes = 0, n_leaves = 0, n_slots = 0, n_descs = 0, n_gfaces = 0, n_groups = 0, n_shared = 0;
   int32_t max_depth = 0;
   float max_normal_len = 0.0f;   // longest vertex normal; NaN if any is (TriangleMesh.computeNormals makes them for degenerate faces)
   double seconds[4] = { 0, 0, 0, 0 };   // upload + normals, hierarchy, leaf groups, records
};

// Preconditions (checked by the caller): use_kdtree, n_faces > max_children, max_depth >= 1, every vertex finite,
// indices in range.  bmin / bmax = the root box over ALL vertices (bounds/KDTree.java:35-44).  normals: the mesh's own
// (n_vertices * 3, host) or null: synthesised on the device as TriangleMesh.computeNormals does (sceneObjects/
// TriangleMesh.java:96-126: face normals summed per vertex in face order).  On success `out` owns its device buffers
// (release with gpu_build_free).
hipError_t gpu_build_object(const float *vertices, uint32_t n_vertices, const float *normals, const int32_t *indices, uint32_t n_faces,
                            const float bmin[3], const float bmax[3], const BuildOptions &opts, const PartBases &bases, GpuBuiltObject &out,
                            std::string &err);
void gpu_build_free(GpuBuiltObject &o);

// Gathers the corners of the n faces listed in `gface` (host) into 48-byte group-face entries at dst_dev (layout.h "gfaces").
hipError_t gpu_emit_gfaces(const float *vertices, uint32_t n_vertices, const int32_t *indices, uint32_t n_faces, const int32_t *gface, size_t n,
                           float *dst_dev);

// TriangleMesh.computeNormals (no-normals branch) on the device, bit for bit what synthesise_normals gives on the host:
// host arrays in, n_vertices * 3 floats out (host).
hipError_t gpu_compute_normals(const float *vertices, uint32_t n_vertices, const int32_t *indices, uint32_t n_faces, float *normals_out);

}   // namespace mi355rt
