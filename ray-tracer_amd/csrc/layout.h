// layout.h -- HBM data layout shared by the host side and the HIP kernels of libmi355rt.
//
// SCENE (read-only during a frame).  One mesh object (= one KDTree of the reference,
// bounds/KDTree.java) is five arrays (NodeRec is the host builder's form, DevNodeRec what is uploaded):
//
// All objects' records are concatenated (links absolute); DevObject gives each object's root.
//   nodes : one 80-byte record per INTERNAL node of the reference hierarchy, in DFS pre-order.
//           A record carries the boxes and links of BOTH children, so one 5 x 128-bit fetch
//           decides both child tests (the reference tests left.minMax / right.minMax at the
//           parent, bounds/KDNode.java:97,101).
//             float4 a = { lmin.x, rmin.x, lmin.y, rmin.y }
//             float4 b = { lmin.z, rmin.z, lmax.x, rmax.x }
//             float4 c = { lmax.y, rmax.y, lmax.z, rmax.z }
//             int4   d = { llink, rlink, lleaf, rleaf }
//             int4   e = { lgbase, lmask, rgbase, rmask }   (leaf children: their descriptors' words, see LEAVES)
//           The two children's values of one coordinate sit side by side (an aligned register pair after the
//           128-bit load), so the walker's slab arithmetic runs both children through packed-f32 instructions
//           (v_pk_add_f32 / v_pk_mul_f32: two results per issue slot).
//           link: index of the child's node record, or (leaf == -1; 0 otherwise) of the first LEAF DESCRIPTOR of a leaf
//           child, so that link ^ leaf is the walker's cursor form at once (a leaf is ~descriptor); link == LINK_NONE ->
//           the reference's null child.  (While a record is being built the two words hold the children's face counts.)
//   LEAVES (scene_build.hpp "leaf groups").  The reference's builder copies a face into every leaf whose box it
//   straddles (KDNode.java:58-69: 6.9 leaf slots per face on the Dragon-class mesh).  Which leaves a ray reaches
//   decides its answer, so the leaves stay; but a face that a ray misses, it misses in every leaf.  So:
//   gfaces : 48 bytes per (group, distinct face): 3 x float4 { v0.xyz, faceId } { v1.xyz, 0 } { v2.xyz, 0 }.  A GROUP is
//           a small subtree with at most 32 distinct faces (stored once, by ascending face id), or a single leaf
//           (faces in the reference's stored order; 32 per group for larger leaves).
//   descs : 16 bytes per leaf (per 32 faces of a larger leaf), leaves in left-to-right DFS order:
//           { key, gbase | flags, mask, 0 }: key = the leaf's first SLOT in the reference's layout (one slot per
//           (leaf, face), leaves in DFS order) -- the leaf's rank in the reference's fold order, which is all the tie
//           rule of KDNode.java:105,114 needs --, gbase = the group's first entry in gfaces, mask = which of the
//           group's faces the leaf holds.  The walker keeps, per lane, the faces of the current group its ray has
//           already missed and tests only mask & ~missed.
//   slot_ids : int32 per slot: the face id, in the reference's stored order (only read when two DIFFERENT faces of
//           one leaf are hit at exactly the same distance: the earlier slot wins, KDNode.java:110-118).
//   fnrm  : 48 bytes per FACE (not per slot): the three vertex normals, fetched once per final hit.
//
// FRAME SCRATCH (per scene handle, sized for one pass of at most `cap` samples).  The frame is a
// wavefront pipeline over compacted queues; every record is 16-byte aligned for 128-bit access:
//
//   samples : float4 per sample slot -- the sample's colour (RenderThread's colors[j])
//   rays    : 2 x float4 per queued ray { o.xyz, bits(slot) } { d.xyz, bits(aux) }
//   hits    : 2 x float4 per queued ray { P.xyz, w } { N.xyz, bits(face) }, plus hit_obj int (-1 miss)
//   mat     : float4 per PRIMARY queue entry { material colour rgb, bits(occluded) }
//   shadow  : u32 per shadow ray = index of its primary queue entry
//   refr_*  : RefractionMaterial bookkeeping per primary queue entry (see FrameArgs)
//   counters: queue sizes and the work cursors of the persistent trace launches
#pragma once
#include <stdint.h>

namespace mi355rt {

constexpr int32_t LINK_NONE = -1;
constexpr int32_t STACK_EMPTY = 0x7fffffff;   // traversal cursor: nothing left (also the root entry of an empty mesh)
constexpr int MAX_TREE_DEPTH = 22;   // build_opts.max_depth limit (the reference uses 20)
constexpr int STACK_SLOTS = 22;      // >= MAX_TREE_DEPTH: at most one deferred sibling per level
constexpr int MAX_OBJECTS = 64;
constexpr int BLOCK_THREADS = 256;
constexpr int GFACE_PAD = 2;         // zero entries after the last group face: lanes that sit a trip out read entry 0 / a second face that is not there

struct NodeRec {   // 64 B
   float lmin[3], lmax[3];
   float rmin[3], rmax[3];
   int32_t llink, rlink, lcnt, rcnt;
};
static_assert(sizeof(NodeRec) == 64, "node record must be 64 bytes");

// the same record as uploaded: left / right interleaved per coordinate ([0] left, [1] right), followed by a copy of each
// LEAF child's descriptor words (group base | flags, mask; 0, 0 for a node child): a lane that steps into a leaf child
// needs no fetch of its own for the descriptor, and a leaf none of whose faces is still untested costs nothing at all
struct DevNodeRec {   // 80 B = 5 x float4
   float lo_x[2], lo_y[2], lo_z[2], hi_x[2], hi_y[2], hi_z[2];
   int32_t llink, rlink, lcnt, rcnt;   // (lcnt, rcnt: face counts while the record is built, -1 / 0 "is a leaf" masks as uploaded)
   int32_t lgbase;
   uint32_t lmask;
   int32_t rgbase;
   uint32_t rmask;
};
static_assert(sizeof(DevNodeRec) == 80, "node record must be 80 bytes");
constexpr int NODE_F4 = 5;   // float4 per node record

// leaf descriptors and groups (see LEAVES above)
constexpr int GROUP_FACES = 32;     // bits of a leaf mask
constexpr int GROUP_SLOTS = 224;    // a subtree with more leaf slots than this is not tried as one group
constexpr int32_t DESC_MORE = (int32_t) 0x80000000;    // LeafDesc.gbase: the leaf goes on in the next descriptor
constexpr int32_t DESC_IDENT = 0x40000000;             // LeafDesc.gbase: the group is this leaf (chunk) alone, its faces in slot order
constexpr int32_t DESC_BASE_MASK = 0x3fffffff;
constexpr int DESC_PAD = 3;         // descriptors after the last: the walker reads three float4 where one is the descriptor

struct LeafDesc {   // 16 B, one per leaf (or per GROUP_FACES faces of a larger leaf)
   int32_t key;     // first slot of the leaf: its rank in the reference's fold order (the tie rule) and where its face ids start
   int32_t gbase;   // first entry of the leaf's group in the group-face array | DESC_MORE | DESC_IDENT
   uint32_t mask;   // which of the group's faces the leaf holds
   int32_t pad;
};
static_assert(sizeof(LeafDesc) == 16, "leaf descriptor must be 16 bytes");

struct DevMaterial {
   uint32_t kind;
   float diffuse[3], specular[3];
   float shininess;
   int32_t brdf_id;
};

// Spheres and planes are entries of the same object array with an empty hierarchy
// (root_entry == STACK_EMPTY, so the walker passes over them); their parameters and the parameters of the
// refraction / checkerboard materials sit in a parallel DevObjectExt array read by analytic_kernels.hip and
// the shade kernel only.
enum { OBJ_MESH = 0, OBJ_SPHERE = 1, OBJ_PLANE = 2 };
enum { MAT_COLOR = 0, MAT_BRDF = 1, MAT_REFLECTION = 2, MAT_REFRACTION = 3, MAT_CHECKERBOARD = 4 };
constexpr int MAX_REFRACTION_TERMS = 8;   // MI355RT_MAX_REFRACTION_TERMS

struct DevObjectExt {
   float center[3], radius;      // Sphere.origin, Sphere.radius
   float pA, pB, pC, pD;         // Plane.A .. Plane.D
   float pnormal[3];             // Plane.normal
   float refraction_index;       // RefractionMaterial.refractionIndex
   float color1[3], color2[3];   // CheckerboardMaterial.color1 / color2
   float cell[3];                // cellWidth, cellHeight, cellDepth
};

struct DevObject {
   float bmin[3], bmax[3];     // KDTree.minMax / BoundingSphere.getMinMax / the Plane's AxisAlignedBoundingBox
   int32_t root_entry;         // node index, ~first_slot when the root is a leaf, STACK_EMPTY for an empty mesh
   float shadow_distance;      // KDTree.shadowDistance
   int32_t face_base;          // first row of this object's faces in fnrm
   int32_t node_base, slot_base;
   DevMaterial mat;
   int32_t kind;               // OBJ_*
   int32_t pad_;
};

// what every kernel needs to walk the scene.  All objects share one node array and one slot array:
// node links and leaf slots are absolute, so a wave's lanes may be inside different objects.
struct SceneArgs {
   const DevObject *objs;
   const void *nodes;   // DevNodeRec[]
   const void *gfaces;  // float4[3 * (n_gfaces + GFACE_PAD)]
   const void *descs;   // int4[n_descs]
   const int32_t *slot_ids;   // [n_slots]
   const void *fnrm;    // float4[3 * n_faces]
   int32_t n_objs;
   float epsilon, near_plane, far_plane;
   float scene_abs_max;   // max |coordinate| of the scene bounds (prune margin)
   float max_normal_len;  // longest shading normal of the scene (bounds the shadow-ray origin offset); NaN if unknown
   const DevObjectExt *ext;   // [n_objs]
   int32_t n_analytic;        // spheres + planes among the objects
   int32_t boxes_finite;      // every node box is finite: rays with finite origin and 1/direction may take the packed slab test
   int32_t big_leaves;        // the scene's leaves hold a dozen faces or more on average: the walker tests their chunks wave-wide
};

// Every queue is N_BANDS sub-queues of band_cap slots each: the pass's 8x8 pixel groups are cut into
// N_BANDS consecutive runs (bands of the image) and a ray, its shadow ray and its bounce rays all stay
// in their pixel's band.  One counter per sub-queue keeps the appends off a single hot address, and a
// walker workgroup starts on band blockIdx % N_BANDS (workgroups b and b + 8 share an XCD, hence an
// L2), so each L2 mostly serves one band's geometry.
constexpr int N_BANDS = 8;

// counters[] slots
enum {
   CNT_Q0 = 0,        // primary queue sizes        [N_BANDS]
   CNT_SHADOW = 8,    // shadow list sizes          [N_BANDS]
   CNT_SEC_A = 16,    // secondary queues (ping-pong) [N_BANDS] each
   CNT_SEC_B = 24,
   CNT_STATS = 32,    // N_STATS x 64-bit stats counters (u32 pairs): 6 public, walker diagnostics, per-kernel counts
   N_STATS = 64,
   CNT_CURSOR0 = 160,  // work cursors of the persistent launches: N_BANDS per launch
   CNT_CURSOR_SETS = 8,
   CNT_WORDS = 160 + 8 * 8
};

struct FrameArgs {
   SceneArgs sc;
   // camera (Camera.java:40-94, 123-126)
   float origin[3];
   float rot[9];
   float x_min, y_min, x_inc, y_inc, near_plane_cam;
   int32_t width, height;
   // settings
   int32_t multi_samples, brdf_samples, lighting_model, max_mirror_bounces;
   int32_t max_recursion_depth;   // SceneObject.getColor's cut-off (spheres and planes only)
   uint64_t rng_seed;
   float light_origin[3], light_emission[3], light_ambient[3];
   int32_t shadow_any_hit;   // host-proven: any accepted hit occludes (see api.cpp)
   int32_t fused_shadow;     // the primary launch walked the shadow rays too (walk_kernel MODE_FUSED): mat0[].w is already set
   int32_t phong_in_shade;   // fused_shadow and no bounce rays in the scene: shade_kernel<0> has all PhongLightingModel needs
   int32_t plain_materials;  // every object's material is a ColorInformation: the shade launch takes the kernel without the other materials' code
   const float *brdf_dirs;      // n_dirs * 3, remapped
   const float *brdf_weights;   // n_materials * n_dirs
   int32_t n_dirs, n_materials;
   // tiles: this pass covers 8x8 pixel groups [micro_begin, micro_end) of the rank's enumeration
   int32_t tile_size, tile_rank, tile_world, tiles_x, tiles_y, micro_per_tile;
   // Division by the frame's few fixed divisors without the divide sequence (a 32-bit one is thirty-odd instructions, and
   // the index arithmetic of generate / resolve is mostly that): n / d = mulhi(n, m) >> s for every n < 2^31, with
   // m = ceil(2^(32+s) / d), s = floor(log2 d); a power of two is a shift (m == 0).  fast_div_of() in api.cpp.
   struct FastDiv {
      uint32_t m, s, d, pad;
   } div_ms, div_micro_per_tile, div_tiles_x, div_mpr;   // multi_samples, micro_per_tile, tiles_x, tile_size / 8
   int32_t tile_compact;   // outputs in tile-compact order (settings.tile_compact)
   int32_t micro_begin, micro_end;
   uint32_t band_cap;   // slots per band sub-queue: ceil(groups in pass / N_BANDS) * 64 * multi_samples
   // scratch
   float *samples;      // float4[cap]
   unsigned long long *queued;   // per wave of generate (64 sample slots): which of them became rays -- the others hold the ambient colour without it being stored
   float *rays0;        // 2 x float4[cap]   primary queue
   float *hits0;        // 2 x float4[cap]
   int32_t *hit_obj0;   // [cap]
   float *mat0;         // float4[cap]
   uint32_t *shadow;    // [cap]
   float *rays1[2];     // secondary queues (null when the scene has no mirror)
   float *hits1;        // 2 x float4[cap]
   int32_t *hit_obj1;
   // RefractionMaterial (null when the scene has none): per path the hit point of the refracting surface whose
   // absorption term is pending, and the stack of terms gathered so far (applied deepest first at the path's end)
   float *refr_parent;  // float4[cap] { P.xyz, - }
   float *refr_terms;   // float4[cap * MAX_REFRACTION_TERMS] { exponent rgb, - }
   uint32_t *counters;  // CNT_WORDS
   // outputs
   float *rgb;
   uint32_t *argb8;
   int32_t *hit_id, *hit_obj;
   float *hit_w;
   int32_t stats;   // count rays / tests (walker template level 1)
   int32_t diag;    // walker diagnostics on top (level 2)
};

struct TraceBatchArgs {   // mi355rt_trace_rays
   SceneArgs sc;
   uint32_t n;
   uint32_t band_cap;             // ceil(n / N_BANDS)
   const float *origins, *dirs;   // n*3 each (user layout)
   float *rays;                   // 2 x float4[n]
   float *hits;                   // 2 x float4[n]
   int32_t *hit_obj;              // [n]
   uint32_t *counters;
   int32_t *out_id, *out_obj;
   float *out_w, *out_point, *out_normal;
};

}   // namespace mi355rt
