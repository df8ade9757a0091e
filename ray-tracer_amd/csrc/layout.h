// layout.h -- HBM data layout shared by the host builder and the HIP kernels.
//
// One mesh object (= one KDTree of the reference, bounds/KDTree.java) is three arrays:
//
//   nodes : one 64-byte record per INTERNAL node of the reference hierarchy, in DFS pre-order.
//           A record carries the boxes and links of BOTH children, so one 4 x 128-bit fetch
//           decides both child tests (the reference tests left.minMax / right.minMax at the
//           parent, bounds/KDNode.java:97,101).
//             float4 a = { lmin.x, lmin.y, lmin.z, lmax.x }
//             float4 b = { lmax.y, lmax.z, rmin.x, rmin.y }
//             float4 c = { rmin.z, rmax.x, rmax.y, rmax.z }
//             int4   d = { llink, rlink, lcnt, rcnt }
//           link: cnt == 0 -> index of the child's node record; cnt > 0 -> first leaf slot of a
//           leaf child holding cnt faces; link == LINK_NONE -> the reference's null child.
//   tris  : one 48-byte slot per (leaf, face) pair, leaves in left-to-right DFS order, faces in
//           the reference's stored order (straddlers are duplicated exactly as KDNode.java:58-69
//           duplicates them).  slot = 3 x float4 { v0.xyz, faceId }, { v1.xyz, 0 }, { v2.xyz, 0 }.
//           Because leaves are laid out in DFS order, the first-slot index of a leaf is also the
//           leaf's rank in the reference's fold order (the tie rule of KDNode.java:105,114).
//   fnrm  : 48 bytes per FACE (not per slot): the three vertex normals, fetched once per final hit.
#pragma once
#include <stdint.h>

namespace mi355rt {

constexpr int32_t LINK_NONE = -1;
constexpr int MAX_TREE_DEPTH = 30;   // build_opts.max_depth limit; traversal stack holds depth+2
constexpr int STACK_SLOTS = 32;
constexpr int MAX_OBJECTS = 64;
constexpr int BLOCK_THREADS = 256;

struct NodeRec {   // 64 B
   float lmin[3], lmax[3];
   float rmin[3], rmax[3];
   int32_t llink, rlink, lcnt, rcnt;
};
static_assert(sizeof(NodeRec) == 64, "node record must be 64 bytes");

struct DevMaterial {
   uint32_t kind;
   float diffuse[3], specular[3];
   float shininess;
   int32_t brdf_id;
};

struct DevObject {
   const void *nodes;   // NodeRec[]
   const void *tris;    // float4[3 * n_slots]
   const void *fnrm;    // float4[3 * n_faces]
   float bmin[3], bmax[3];
   int32_t root_link, root_cnt;   // root_cnt > 0 (or n_faces == 0): the root itself is a leaf
   int32_t root_is_leaf;
   float shadow_distance;
   DevMaterial mat;
   int32_t pad_[2];
};

struct RenderParams {
   // camera (Camera.java:40-94, 123-126)
   float origin[3];
   float rot[9];
   float x_min, y_min, x_inc, y_inc, near_plane_cam;
   int32_t width, height;
   // settings
   float epsilon, near_plane, far_plane;
   int32_t multi_samples, brdf_samples, lighting_model, max_mirror_bounces;
   uint64_t rng_seed;
   float light_origin[3], light_emission[3], light_ambient[3];
   // scene
   int32_t n_objs;
   const DevObject *objs;
   const float *brdf_dirs;      // n_dirs * 3, remapped
   const float *brdf_weights;   // n_materials * n_dirs
   int32_t n_dirs, n_materials;
   float scene_abs_max;         // max |coordinate| over the scene bounds (prune margin)
   // tiles
   int32_t tile_size, tile_rank, tile_world, tiles_x, tiles_y, n_owned_tiles, micro_per_tile, n_micro;
   // outputs
   float *rgb;
   uint32_t *argb8;
   int32_t *hit_id, *hit_obj;
   float *hit_w;
   unsigned long long *stats;   // 6 counters or nullptr
};

struct TraceParams {
   float epsilon, near_plane, far_plane;
   int32_t n_objs;
   const DevObject *objs;
   float scene_abs_max;
   uint64_t n;
   const float *origins, *dirs;
   int32_t *hit_id, *hit_obj;
   float *hit_w, *hit_point, *hit_normal;
};

}   // namespace mi355rt
