// gpu_build.hip -- the reference's hierarchy built on the GPU (second half).
//
// What is built is exactly what scene_build.cpp builds on the host, and what KDNode's constructor builds in the
// reference (bounds/KDNode.java:29-90): split while depth < MAX_DEPTH and faces > MAX_CHILDREN at the midpoint of
// the node's box on the axis cycling X -> Y -> Z; a face goes left if its max <= plane, right if its min >= plane,
// otherwise to BOTH sides; faces keep their stored order; a child's box is the tight box of its faces; an empty side
// is a null child.  The recursion is run level by level instead:
//
//   entries   (face, node) pairs of the current level, grouped by node, faces in stored order
//   per level classify every entry against its node's plane -> exclusive scan of the (left, right) flags ->
//             per node: child sizes -> exclusive scan over nodes: child numbering and segment starts ->
//             stable scatter of the entries into the children's segments, tight boxes by atomic min / max on
//             order-preserving integer keys (min and max do not depend on the order); entries of nodes that do
//             not split are parked in the leaf store
//   finally   subtree sizes bottom-up, DFS pre-order numbers and first-slot numbers top-down, then one kernel writes
//             the node records and one the leaf slots' face ids, in the layout of layout.h (the leaf groups follow on the host: scene_build.hpp).
//
// Every float that ends up in a record is produced by the same operation as on the host ((hi - lo) / 2 + lo for the
// plane, min / max for the boxes), so the arrays are byte-identical to the host builder's (tests/test_gpu_build.py
// compares them).  Java's Math.min / Math.max differ from integer-key min / max only for NaN, which the caller
// excludes (non-finite vertices take the host builder).  One small read-back per level (child and entry totals)
// sizes the next level's buffers.
#include "gpu_build.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <vector>

namespace mi355rt {

namespace {

constexpr unsigned NONE = 0xffffffffu;
constexpr int SCAN_THREADS = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;
constexpr float FLOAT_MAX = 3.4028234663852886e38f;

struct S2 {
   unsigned a, b;
};
struct S4 {
   unsigned a, b, c, d;
};
__host__ __device__ inline S2 operator+(S2 x, S2 y) { return S2 { x.a + y.a, x.b + y.b }; }
__host__ __device__ inline S4 operator+(S4 x, S4 y) { return S4 { x.a + y.a, x.b + y.b, x.c + y.c, x.d + y.d }; }
template <typename T> __host__ __device__ inline T zero_of();
template <> __host__ __device__ inline S2 zero_of<S2>() { return S2 { 0u, 0u }; }
template <> __host__ __device__ inline S4 zero_of<S4>() { return S4 { 0u, 0u, 0u, 0u }; }

struct BNode {   // one node of the hierarchy under construction
   unsigned begin, count;   // its entries in its level's entry array
   unsigned box[6];         // tight box as order-preserving keys: lo xyz (min), hi xyz (max)
   unsigned child[2];       // global node indices, NONE = null child
   unsigned split;          // 1: internal node
   unsigned nl, nr;         // entries going left / right
   unsigned store;          // leaf: first entry in the leaf store
   unsigned n_int, n_slots, n_leaves;   // subtree totals: internal nodes, leaf slots, leaves
   unsigned dfs, sbase;     // internal: DFS pre-order number; any: first leaf slot of the subtree
   unsigned parent;         // global node index, NONE for the root
};

// order-preserving map float -> unsigned (-0 < +0, as Java's Math.min / max order them)
__host__ __device__ inline unsigned float_key(float f)
{
   union {
      float f;
      unsigned u;
   } c;
   c.f = f;
   return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}
__host__ __device__ inline float key_float(unsigned k)
{
   union {
      float f;
      unsigned u;
   } c;
   c.u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
   return c.f;
}

// ---------------------------------------------------------------- exclusive scan (n + 1 outputs: out[n] = total)

// exclusive scan of SCAN_THREADS values held one per thread; returns the thread's exclusive prefix, total in `total`
template <typename T> __device__ inline T block_exclusive(T v, T *sh, T &total)
{
   sh[threadIdx.x] = v;
   __syncthreads();
   for (int off = 1; off < SCAN_THREADS; off <<= 1) {   // inclusive Hillis-Steele
      T add = zero_of<T>();
      if ((int) threadIdx.x >= off) add = sh[threadIdx.x - off];
      __syncthreads();
      sh[threadIdx.x] = sh[threadIdx.x] + add;
      __syncthreads();
   }
   const T ex = threadIdx.x ? sh[threadIdx.x - 1] : zero_of<T>();
   total = sh[SCAN_THREADS - 1];
   __syncthreads();
   return ex;
}

template <typename T> __global__ __launch_bounds__(SCAN_THREADS) void scan_block_sums(const T *in, unsigned n, T *bsum)
{
   __shared__ T sh[SCAN_THREADS];
   const unsigned base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
   T acc = zero_of<T>();
   for (int k = 0; k < SCAN_ITEMS; k++)
      if (base + k < n) acc = acc + in[base + k];
   T total;
   (void) block_exclusive(acc, sh, total);
   if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// one workgroup: exclusive scan of the block sums in place; bsum[n_blocks] = grand total
template <typename T> __global__ __launch_bounds__(SCAN_THREADS) void scan_of_sums(T *bsum, unsigned n_blocks)
{
   __shared__ T sh[SCAN_THREADS];
   T carry = zero_of<T>();   // the same in every thread
   for (unsigned base = 0; base < n_blocks; base += SCAN_THREADS) {
      const unsigned i = base + threadIdx.x;
      const T v = i < n_blocks ? bsum[i] : zero_of<T>();
      T total;
      const T ex = block_exclusive(v, sh, total);
      if (i < n_blocks) bsum[i] = carry + ex;
      carry = carry + total;
   }
   if (threadIdx.x == 0) bsum[n_blocks] = carry;
}

template <typename T> __global__ __launch_bounds__(SCAN_THREADS) void scan_apply(const T *in, unsigned n, const T *bsum, unsigned n_blocks, T *out)
{
   __shared__ T sh[SCAN_THREADS];
   const unsigned base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
   T item[SCAN_ITEMS];
   T acc = zero_of<T>();
   for (int k = 0; k < SCAN_ITEMS; k++) {
      item[k] = base + k < n ? in[base + k] : zero_of<T>();
      acc = acc + item[k];
   }
   T total;
   T run = bsum[blockIdx.x] + block_exclusive(acc, sh, total);
   for (int k = 0; k < SCAN_ITEMS; k++) {
      if (base + k < n) out[base + k] = run;
      run = run + item[k];
   }
   if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = bsum[n_blocks];
}

// ---------------------------------------------------------------- set-up

__global__ void face_bounds_kernel(const float *vertices, const int32_t *indices, unsigned n_faces, float *fb, unsigned *ent_face, unsigned *ent_node)
{
   const unsigned f = blockIdx.x * blockDim.x + threadIdx.x;
   if (f >= n_faces) return;
   float lo[3] = { FLOAT_MAX, FLOAT_MAX, FLOAT_MAX }, hi[3] = { -FLOAT_MAX, -FLOAT_MAX, -FLOAT_MAX };
   for (int k = 0; k < 3; k++) {
      const float *p = vertices + 3 * (size_t) indices[3 * (size_t) f + k];
      for (int c = 0; c < 3; c++) {   // RTStatics.getMinMax per face (RTStatics.java:244-264); finite values: min / max as Java's
         lo[c] = float_key(p[c]) < float_key(lo[c]) ? p[c] : lo[c];
         hi[c] = float_key(p[c]) > float_key(hi[c]) ? p[c] : hi[c];
      }
   }
   for (int c = 0; c < 3; c++) {
      fb[6 * (size_t) f + c] = lo[c];
      fb[6 * (size_t) f + 3 + c] = hi[c];
   }
   ent_face[f] = f;
   ent_node[f] = 0u;
}

__global__ void root_kernel(BNode *nodes, unsigned n_faces, float lx, float ly, float lz, float hx, float hy, float hz)
{
   BNode r;
   r.begin = 0;
   r.count = n_faces;
   r.box[0] = float_key(lx);
   r.box[1] = float_key(ly);
   r.box[2] = float_key(lz);
   r.box[3] = float_key(hx);
   r.box[4] = float_key(hy);
   r.box[5] = float_key(hz);
   r.child[0] = r.child[1] = NONE;
   r.split = r.nl = r.nr = r.store = r.n_int = r.n_slots = r.n_leaves = r.dfs = r.sbase = 0;
   r.parent = NONE;
   nodes[0] = r;
}

// ---------------------------------------------------------------- one level

// KDNode.java:40: does this node split?  (depth and the two limits are the same for the whole level)
__global__ void level_split_kernel(BNode *nodes, unsigned level_off, unsigned n_nodes, int can_split, unsigned max_children)
{
   const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_nodes) return;
   BNode &nd = nodes[level_off + i];
   nd.split = (can_split && nd.count > max_children) ? 1u : 0u;
}

// KDNode.java:45-69: which side(s) does each face of a splitting node go to
__global__ void level_classify_kernel(const BNode *nodes, unsigned level_off, const unsigned *ent_face, const unsigned *ent_node, unsigned n_entries,
                                      const float *fb, int axis, S2 *flags)
{
   const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
   if (e >= n_entries) return;
   const BNode &nd = nodes[level_off + ent_node[e]];
   S2 fl = { 0u, 0u };
   if (nd.split) {
      const float lo = key_float(nd.box[axis]), hi = key_float(nd.box[3 + axis]);
      const float plane = (hi - lo) / 2.0f + lo;
      const float *b = fb + 6 * (size_t) ent_face[e];
      if (b[3 + axis] <= plane) fl.a = 1u;
      else if (b[axis] >= plane) fl.b = 1u;
      else fl.a = fl.b = 1u;
   }
   flags[e] = fl;
}

// per node: sizes of the two sides; what the node contributes to the next level and to the leaf store
__global__ void level_count_kernel(BNode *nodes, unsigned level_off, unsigned n_nodes, const S2 *scan_e, S4 *contrib)
{
   const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_nodes) return;
   BNode &nd = nodes[level_off + i];
   S4 c = { 0u, 0u, 0u, 0u };
   if (nd.split) {
      const S2 a = scan_e[nd.begin], b = scan_e[nd.begin + nd.count];
      nd.nl = b.a - a.a;
      nd.nr = b.b - a.b;
      c.a = (nd.nl ? 1u : 0u) + (nd.nr ? 1u : 0u);   // children
      c.b = nd.nl + nd.nr;                          // entries of the next level
      c.d = 1u;
   } else {
      c.c = nd.count;                               // entries parked in the leaf store
   }
   contrib[i] = c;
}

__device__ inline void init_child(BNode &ch, unsigned begin, unsigned count, unsigned parent)
{
   ch.parent = parent;
   ch.begin = begin;
   ch.count = count;
   ch.box[0] = ch.box[1] = ch.box[2] = 0xffffffffu;   // min keys
   ch.box[3] = ch.box[4] = ch.box[5] = 0u;            // max keys
   ch.child[0] = ch.child[1] = NONE;
   ch.split = ch.nl = ch.nr = ch.store = ch.n_int = ch.n_slots = ch.n_leaves = ch.dfs = ch.sbase = 0;
}

__global__ void level_children_kernel(BNode *nodes, unsigned level_off, unsigned n_nodes, unsigned next_off, const S4 *scan_n, unsigned store_base)
{
   const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_nodes) return;
   BNode &nd = nodes[level_off + i];
   const S4 s = scan_n[i];
   if (!nd.split) {
      nd.store = store_base + s.c;
      return;
   }
   unsigned ci = next_off + s.a;
   if (nd.nl) {
      init_child(nodes[ci], s.b, nd.nl, level_off + i);
      nd.child[0] = ci++;
   }
   if (nd.nr) {
      init_child(nodes[ci], s.b + nd.nl, nd.nr, level_off + i);
      nd.child[1] = ci;
   }
}

// Folds the bounds of one face per active lane into its child's box.  Entries are grouped by node and keep their order,
// so near the root a whole wave feeds the same child: then the wave reduces its 64 boxes with shuffles and issues
// 6 atomics instead of 384 on the same words (the first levels of a 28-million-face mesh otherwise serialise on two
// nodes' boxes).  Every lane of the wave must call it.
__device__ inline void grow_box_wave(BNode *nodes, bool active, unsigned child, const float *b)
{
   const unsigned long long m = __ballot(active);
   if (m == 0ull) return;
   const int leader = __ffsll((unsigned long long) m) - 1;
   const unsigned c0 = (unsigned) __shfl((int) child, leader, 64);
   unsigned k[6];
   for (int c = 0; c < 3; c++) {
      k[c] = active ? float_key(b[c]) : 0xffffffffu;
      k[3 + c] = active ? float_key(b[3 + c]) : 0u;
   }
   if (__ballot(active && child != c0) == 0ull) {
      for (int off = 32; off > 0; off >>= 1) {
         for (int c = 0; c < 3; c++) {
            k[c] = min(k[c], (unsigned) __shfl_xor((int) k[c], off, 64));
            k[3 + c] = max(k[3 + c], (unsigned) __shfl_xor((int) k[3 + c], off, 64));
         }
      }
      if ((int) (threadIdx.x & 63) == leader) {
         BNode &ch = nodes[c0];
         for (int c = 0; c < 3; c++) {
            atomicMin(&ch.box[c], k[c]);
            atomicMax(&ch.box[3 + c], k[3 + c]);
         }
      }
   } else if (active) {
      BNode &ch = nodes[child];
      for (int c = 0; c < 3; c++) {
         atomicMin(&ch.box[c], k[c]);
         atomicMax(&ch.box[3 + c], k[3 + c]);
      }
   }
}

// stable scatter into the children's segments (faces keep their stored order, KDNode.java:58-69), tight child boxes
// (KDNode.java:71-85), and the entries of leaves into the leaf store
__global__ __launch_bounds__(256) void level_scatter_kernel(BNode *nodes, unsigned level_off, unsigned next_off, const unsigned *ent_face,
                                                            const unsigned *ent_node, unsigned n_entries, const S2 *scan_e, const float *fb,
                                                            unsigned *next_face, unsigned *next_node, unsigned *store_face, unsigned *store_node)
{
   const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
   const bool live = e < n_entries;
   bool go[2] = { false, false };
   unsigned child[2] = { 0u, 0u };
   const float *b = fb;
   if (live) {
      const unsigned g = level_off + ent_node[e];
      const BNode &nd = nodes[g];
      const unsigned face = ent_face[e];
      b = fb + 6 * (size_t) face;
      if (!nd.split) {
         const unsigned j = nd.store + (e - nd.begin);
         store_face[j] = face;
         store_node[j] = g;
      } else {
         const S2 here = scan_e[e], next = scan_e[e + 1], first = scan_e[nd.begin];
         go[0] = next.a != here.a;
         go[1] = next.b != here.b;
         child[0] = nd.child[0];
         child[1] = nd.child[1];
         const unsigned pos[2] = { here.a - first.a, here.b - first.b };
         for (int s = 0; s < 2; s++) {
            if (!go[s]) continue;
            const unsigned dst = nodes[child[s]].begin + pos[s];
            next_face[dst] = face;
            next_node[dst] = child[s] - next_off;
         }
      }
   }
   grow_box_wave(nodes, go[0], child[0], b);
   grow_box_wave(nodes, go[1], child[1], b);
}

// ---------------------------------------------------------------- layout

__global__ void subtree_sizes_kernel(BNode *nodes, unsigned level_off, unsigned n_nodes)
{
   const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_nodes) return;
   BNode &nd = nodes[level_off + i];
   if (!nd.split) {
      nd.n_int = 0;
      nd.n_slots = nd.count;
      nd.n_leaves = 1;
      return;
   }
   unsigned ni = 1, ns = 0, nl = 0;
   for (int s = 0; s < 2; s++) {
      if (nd.child[s] == NONE) continue;
      const BNode &ch = nodes[nd.child[s]];
      ni += ch.n_int;
      ns += ch.n_slots;
      nl += ch.n_leaves;
   }
   nd.n_int = ni;
   nd.n_slots = ns;
   nd.n_leaves = nl;
}

// DFS pre-order: a node, then its whole left subtree, then its right subtree (scene_build.cpp's emit order)
__global__ void dfs_numbers_kernel(BNode *nodes, unsigned level_off, unsigned n_nodes)
{
   const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_nodes) return;
   const BNode &nd = nodes[level_off + i];
   if (!nd.split) return;
   unsigned dfs = nd.dfs + 1, sbase = nd.sbase;
   for (int s = 0; s < 2; s++) {
      if (nd.child[s] == NONE) continue;
      BNode &ch = nodes[nd.child[s]];
      ch.dfs = dfs;
      ch.sbase = sbase;
      dfs += ch.n_int;
      sbase += ch.n_slots;
   }
}

// node records in the uploaded form: links absolute (node_base / desc_base folded in); a leaf child's link is the index
// of its first descriptor (slot_base[first slot].a, from the grouping pass below)
__global__ void emit_nodes_kernel(const BNode *nodes, unsigned n_all, const S2 *slot_base, unsigned node_base, unsigned desc_base, DevNodeRec *out)
{
   const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= n_all) return;
   const BNode &nd = nodes[g];
   if (!nd.split) return;
   DevNodeRec r;
   int32_t link[2] = { LINK_NONE, LINK_NONE }, cnt[2] = { 0, 0 };
   for (int s = 0; s < 2; s++) {
      float lo[3] = { FLOAT_MAX, FLOAT_MAX, FLOAT_MAX }, hi[3] = { -FLOAT_MAX, -FLOAT_MAX, -FLOAT_MAX };   // scene_build.cpp's null child
      if (nd.child[s] != NONE) {
         const BNode &ch = nodes[nd.child[s]];
         for (int c = 0; c < 3; c++) {
            lo[c] = key_float(ch.box[c]);
            hi[c] = key_float(ch.box[3 + c]);
         }
         link[s] = (int32_t) (ch.split ? ch.dfs + node_base : slot_base[ch.sbase].a + desc_base);
         cnt[s] = ch.split ? 0 : (int32_t) ch.count;
      }
      r.lo_x[s] = lo[0];
      r.lo_y[s] = lo[1];
      r.lo_z[s] = lo[2];
      r.hi_x[s] = hi[0];
      r.hi_y[s] = hi[1];
      r.hi_z[s] = hi[2];
   }
   r.llink = link[0];
   r.rlink = link[1];
   r.lcnt = cnt[0];
   r.rcnt = cnt[1];
   r.lgbase = r.rgbase = 0;   // leaf_words_kernel, once the descriptors are complete
   r.lmask = r.rmask = 0u;
   out[nd.dfs] = r;
}

// a leaf child's descriptor words ride in its parent's record (layout.h)
__global__ void leaf_words_kernel(DevNodeRec *recs, unsigned n, const LeafDesc *descs, unsigned desc_base)
{
   const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   DevNodeRec &r = recs[i];
   if (r.llink != LINK_NONE && r.lcnt > 0) {
      const LeafDesc d = descs[(unsigned) r.llink - desc_base];
      r.lgbase = d.gbase;
      r.lmask = d.mask;
   }
   if (r.rlink != LINK_NONE && r.rcnt > 0) {
      const LeafDesc d = descs[(unsigned) r.rlink - desc_base];
      r.rgbase = d.gbase;
      r.rmask = d.mask;
   }
   r.lcnt = r.lcnt > 0 ? -1 : 0;   // the uploaded form keeps "is a leaf" as a mask (layout.h)
   r.rcnt = r.rcnt > 0 ? -1 : 0;
}

// leaf slots in the reference's order: the face id of every (leaf, face) pair, and the first slot of the slot's leaf
__global__ void emit_slots_kernel(const BNode *nodes, const unsigned *store_face, const unsigned *store_node, unsigned n_store, int32_t *slot_ids,
                                  unsigned *slot_leaf)
{
   const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
   if (j >= n_store) return;
   const BNode &nd = nodes[store_node[j]];
   slot_ids[nd.sbase + (j - nd.store)] = (int32_t) store_face[j];
   slot_leaf[nd.sbase + (j - nd.store)] = nd.sbase;
}

// ---------------------------------------------------------------- leaf groups (scene_build.hpp "leaf groups", Grouper)
// The host's rule, top down: a subtree of at most GROUP_SLOTS leaf slots whose distinct faces number at most GROUP_FACES
// is ONE group (its faces stored once, ascending; every leaf a mask over them); a leaf outside any such subtree is its
// own group (faces in slot order, GROUP_FACES per descriptor).  Being such a subtree is inherited by every subtree of
// it, so the groups are the qualifying nodes whose parent does not qualify.  Descriptors follow the leaves' DFS order
// (= slot order) and group faces the groups' DFS order, so both places come from one scan over the slots.

constexpr int GROUP_WAVES = 4;   // nodes per workgroup: one wave each

// qual[g] = number of distinct faces of subtree g if it qualifies as a group, else 0
__global__ __launch_bounds__(64 * GROUP_WAVES) void group_qualify_kernel(const BNode *nodes, unsigned n_all, const int32_t *slot_ids, unsigned *qual)
{
   __shared__ int32_t sh_ids[GROUP_WAVES][GROUP_SLOTS];
   const unsigned g = blockIdx.x * GROUP_WAVES + (threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   if (g >= n_all) return;
   const BNode &nd = nodes[g];
   if (!nd.split || nd.n_slots > (unsigned) GROUP_SLOTS) {
      if (lane == 0) qual[g] = 0u;
      return;
   }
   int32_t *ids = sh_ids[threadIdx.x >> 6];
   const int n = (int) nd.n_slots;
   for (int i = lane; i < n; i += 64) ids[i] = slot_ids[nd.sbase + (unsigned) i];
   __builtin_amdgcn_wave_barrier();
   __builtin_amdgcn_s_waitcnt(0xc07f);   // (one wave: its LDS writes are in order; wait for them before the reads)
   unsigned distinct = 0;
   for (int i = lane; i < n; i += 64) {
      const int32_t v = ids[i];
      bool first = true;
      for (int j = 0; j < i; j++) first = first && ids[j] != v;
      distinct += first ? 1u : 0u;
   }
   for (int off = 32; off > 0; off >>= 1) distinct += (unsigned) __shfl_xor((int) distinct, off, 64);
   if (lane == 0) qual[g] = distinct <= (unsigned) GROUP_FACES ? distinct : 0u;
}

// per leaf: its descriptors; per group: its faces -- counted at the first slot of the leaf / group (cnt: .a descriptors, .b group faces)
__global__ void group_count_kernel(const BNode *nodes, unsigned n_all, const unsigned *qual, S2 *cnt, unsigned long long *totals)
{
   const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= n_all) return;
   const BNode &nd = nodes[g];
   const bool parent_ok = nd.parent != NONE && qual[nd.parent] != 0u;
   if (nd.split) {
      if (qual[g] != 0u && !parent_ok) {   // a group of several leaves
         cnt[nd.sbase].b = qual[g];
         atomicAdd(totals + 0, 1ull);
         atomicAdd(totals + 1, 1ull);
      }
   } else if (parent_ok) {
      cnt[nd.sbase].a = 1u;
   } else {   // a leaf on its own
      const unsigned chunks = (nd.count + GROUP_FACES - 1) / GROUP_FACES;
      cnt[nd.sbase].a = chunks;
      cnt[nd.sbase].b = nd.count;
      atomicAdd(totals + 0, (unsigned long long) chunks);
   }
}

// one wave per group of several leaves: its distinct faces ascending, then every leaf's mask over them
__global__ __launch_bounds__(64 * GROUP_WAVES) void group_emit_kernel(const BNode *nodes, unsigned n_all, const unsigned *qual, const int32_t *slot_ids,
                                                                     const unsigned *slot_leaf, const S2 *base, PartBases bases, LeafDesc *descs,
                                                                     int32_t *gface_ids)
{
   __shared__ int32_t sh_ids[GROUP_WAVES][GROUP_SLOTS];
   __shared__ unsigned char sh_first[GROUP_WAVES][GROUP_SLOTS];
   const unsigned g = blockIdx.x * GROUP_WAVES + (threadIdx.x >> 6);
   const int lane = threadIdx.x & 63;
   if (g >= n_all) return;
   const BNode &nd = nodes[g];
   if (!nd.split || qual[g] == 0u || (nd.parent != NONE && qual[nd.parent] != 0u)) return;
   int32_t *ids = sh_ids[threadIdx.x >> 6];
   unsigned char *first = sh_first[threadIdx.x >> 6];
   const int n = (int) nd.n_slots;
   for (int i = lane; i < n; i += 64) ids[i] = slot_ids[nd.sbase + (unsigned) i];
   __builtin_amdgcn_wave_barrier();
   __builtin_amdgcn_s_waitcnt(0xc07f);
   for (int i = lane; i < n; i += 64) {
      const int32_t v = ids[i];
      bool f = true;
      for (int j = 0; j < i; j++) f = f && ids[j] != v;
      first[i] = f ? 1 : 0;
   }
   __builtin_amdgcn_wave_barrier();
   __builtin_amdgcn_s_waitcnt(0xc07f);
   const unsigned gbase = base[nd.sbase].b;   // the group's first entry among the object's group faces
   for (int i = lane; i < n; i += 64) {
      const int32_t v = ids[i];
      unsigned pos = 0;   // rank of v among the distinct faces
      for (int j = 0; j < n; j++) pos += (first[j] && ids[j] < v) ? 1u : 0u;
      if (first[i]) gface_ids[gbase + pos] = v;
      const unsigned leaf = slot_leaf[nd.sbase + (unsigned) i];
      LeafDesc *d = descs + base[leaf].a;
      atomicOr(&d->mask, 1u << pos);
      if (nd.sbase + (unsigned) i == leaf) {
         d->key = (int32_t) (leaf + (unsigned) bases.slot);
         d->gbase = (int32_t) (gbase + (unsigned) bases.gface);
      }
   }
}

// leaves on their own: GROUP_FACES faces per descriptor, in slot order
__global__ void group_alone_kernel(const BNode *nodes, unsigned n_all, const unsigned *qual, const int32_t *slot_ids, const S2 *base, PartBases bases,
                                   LeafDesc *descs, int32_t *gface_ids)
{
   const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
   if (g >= n_all) return;
   const BNode &nd = nodes[g];
   if (nd.split || (nd.parent != NONE && qual[nd.parent] != 0u)) return;
   const unsigned d0 = base[nd.sbase].a, g0 = base[nd.sbase].b, cnt = nd.count;
   for (unsigned done = 0, c = 0; done < cnt; done += GROUP_FACES, c++) {
      const unsigned n = min((unsigned) GROUP_FACES, cnt - done);
      LeafDesc d;
      d.key = (int32_t) (nd.sbase + (unsigned) bases.slot);
      d.gbase = ((int32_t) (g0 + done + (unsigned) bases.gface)) | DESC_IDENT | (done + n < cnt ? DESC_MORE : 0);
      d.mask = n == 32u ? 0xffffffffu : ((1u << n) - 1u);
      d.pad = 0;
      descs[d0 + c] = d;
   }
   for (unsigned k = 0; k < cnt; k++) gface_ids[g0 + k] = slot_ids[nd.sbase + k];
}

// ---------------------------------------------------------------- vertex normals (TriangleMesh.computeNormals, sceneObjects/TriangleMesh.java:96-126)
// The reference adds every face's unit normal to its three vertices in FACE ORDER and divides each sum by its length;
// f32 sums depend on their order, so each vertex gets the list of its faces, sorted, and adds them up in that order.

__global__ void normal_count_kernel(const int32_t *indices, unsigned n_faces, unsigned *count)
{
   const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;   // one (face, corner)
   if (c >= 3u * n_faces) return;
   atomicAdd(count + indices[c], 1u);
}
__global__ void normal_fill_kernel(const int32_t *indices, unsigned n_faces, const unsigned *offset, unsigned *cursor, unsigned *inc)
{
   const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
   if (c >= 3u * n_faces) return;
   const int32_t v = indices[c];
   inc[offset[v] + atomicAdd(cursor + v, 1u)] = c / 3u;
}
// RTStatics.computeNormal (RTStatics.java:323-336) with Vector3f.normalize
__device__ inline void face_normal(const float *vertices, const int32_t *indices, unsigned f, float n[3])
{
   const int32_t *face = indices + 3 * (size_t) f;
   const float *a = vertices + 3 * (size_t) face[0], *b = vertices + 3 * (size_t) face[1], *c = vertices + 3 * (size_t) face[2];
   const float e1[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, e2[3] = { c[0] - a[0], c[1] - a[1], c[2] - a[2] };
   const float x = e1[1] * e2[2] - e1[2] * e2[1], y = e2[0] * e1[2] - e2[2] * e1[0], z = e1[0] * e2[1] - e1[1] * e2[0];
   const float len2 = x * x + y * y + z * z;
   const float inv = (float) (1.0 / sqrt((double) len2));
   n[0] = x * inv;
   n[1] = y * inv;
   n[2] = z * inv;
}
__global__ void normal_sum_kernel(const float *vertices, const int32_t *indices, unsigned n_vertices, const unsigned *offset, unsigned *inc, float *normals)
{
   const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
   if (v >= n_vertices) return;
   const unsigned b = offset[v], e = offset[v + 1];
   for (unsigned i = b + 1; i < e; i++) {   // the vertex's faces ascending (a handful; equal ids -- a face with this corner twice -- stay twice)
      const unsigned f = inc[i];
      unsigned j = i;
      for (; j > b && inc[j - 1] > f; j--) inc[j] = inc[j - 1];
      inc[j] = f;
   }
   float d[3] = { 0.0f, 0.0f, 0.0f };
   for (unsigned i = b; i < e; i++) {
      float n[3];
      face_normal(vertices, indices, inc[i], n);
      d[0] += n[0];
      d[1] += n[1];
      d[2] += n[2];
   }
   const float sq = (0.0f - d[0]) * (0.0f - d[0]) + (0.0f - d[1]) * (0.0f - d[1]) + (0.0f - d[2]) * (0.0f - d[2]);   // RTStatics.getDistance from the origin
   const float len = (float) sqrt((double) sq);
   d[0] /= len;
   d[1] /= len;
   d[2] /= len;
   normals[3 * (size_t) v + 0] = d[0];
   normals[3 * (size_t) v + 1] = d[1];
   normals[3 * (size_t) v + 2] = d[2];
}
// the longest normal (bounds the shadow rays' origin offset, api.cpp); a NaN length sticks
__global__ void normal_max_kernel(const float *normals, unsigned n_vertices, unsigned *max_len_bits, unsigned *any_nan)
{
   const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
   if (v >= n_vertices) return;
   const float *nv = normals + 3 * (size_t) v;
   const float len = sqrtf(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
   if (len != len) atomicOr(any_nan, 1u);
   else atomicMax(max_len_bits, __float_as_uint(len));   // non-negative floats order as their bits
}
// 48 bytes per face: its three vertex normals (fetched once per final hit)
__global__ void face_normals_kernel(const float *normals, const int32_t *indices, unsigned n_faces, float *fnrm)
{
   const unsigned f = blockIdx.x * blockDim.x + threadIdx.x;
   if (f >= n_faces) return;
   float4 *dst = (float4 *) fnrm + 3 * (size_t) f;
   for (int k = 0; k < 3; k++) {
      const float *nv = normals + 3 * (size_t) indices[3 * (size_t) f + k];
      dst[k] = make_float4(nv[0], nv[1], nv[2], 0.0f);
   }
}

// group faces (layout.h): { v0.xyz, faceId } { v1.xyz, 0 } { v2.xyz, 0 } for every entry of the group-face list
__global__ void emit_gfaces_kernel(const float *vertices, const int32_t *indices, const int32_t *gface, size_t n, float *out)
{
   const size_t j = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
   if (j >= n) return;
   const int32_t face = gface[j];
   const int32_t *ix = indices + 3 * (size_t) face;
   const float *a = vertices + 3 * (size_t) ix[0], *b = vertices + 3 * (size_t) ix[1], *c = vertices + 3 * (size_t) ix[2];
   float4 *dst = (float4 *) out + 3 * j;
   dst[0] = make_float4(a[0], a[1], a[2], __int_as_float((int) face));
   dst[1] = make_float4(b[0], b[1], b[2], 0.0f);
   dst[2] = make_float4(c[0], c[1], c[2], 0.0f);
}

// ---------------------------------------------------------------- host driver

unsigned grid_for(size_t n) { return (unsigned) ((n + 255) / 256); }

// a device array that grows (contents kept)
template <typename T> struct DevVec {
   T *p = nullptr;
   size_t cap = 0;
   DevVec() = default;
   DevVec(const DevVec &) = delete;
   DevVec &operator=(const DevVec &) = delete;
   ~DevVec() { release(); }
   void release()
   {
      if (p) (void) hipFree(p);
      p = nullptr;
      cap = 0;
   }
   hipError_t ensure(size_t n, size_t keep)
   {
      if (n <= cap) return hipSuccess;
      const size_t want = std::max(n, cap + cap / 2);
      T *q = nullptr;
      hipError_t e = hipMalloc((void **) &q, want * sizeof(T));
      if (e != hipSuccess) return e;
      if (keep && p) e = hipMemcpy(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice);
      if (p) (void) hipFree(p);
      p = q;
      cap = want;
      return e;
   }
};

template <typename T> hipError_t exclusive_scan(const T *in, size_t n, T *out, DevVec<T> &bsum)
{
   const unsigned n_blocks = (unsigned) ((n + SCAN_TILE - 1) / SCAN_TILE);
   hipError_t e = bsum.ensure((size_t) n_blocks + 1, 0);
   if (e != hipSuccess) return e;
   if (n_blocks) hipLaunchKernelGGL((scan_block_sums<T>), dim3(n_blocks), dim3(SCAN_THREADS), 0, nullptr, in, (unsigned) n, bsum.p);
   hipLaunchKernelGGL((scan_of_sums<T>), dim3(1), dim3(SCAN_THREADS), 0, nullptr, bsum.p, n_blocks);
   hipLaunchKernelGGL((scan_apply<T>), dim3(n_blocks ? n_blocks : 1), dim3(SCAN_THREADS), 0, nullptr, in, (unsigned) n, bsum.p, n_blocks, out);
   return hipGetLastError();
}

#define GB_TRY(expr)                    \
   do {                                 \
      hipError_t e_ = (expr);           \
      if (e_ != hipSuccess) return e_;  \
   } while (0)

}   // namespace

void gpu_build_free(GpuBuiltObject &o)
{
   void *all[] = { o.nodes, o.slot_ids, o.descs, o.gfaces, o.fnrm };
   for (void *q : all)
      if (q) (void) hipFree(q);
   o = GpuBuiltObject();
}

namespace {

// one unsigned per item -> exclusive scan, through the S2 scan (its .b lane idles)
struct U1Scan {
   DevVec<S2> in, out, bsum;
};

__global__ void widen_kernel(const unsigned *in, unsigned n, S2 *out)
{
   const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) out[i] = S2 { in[i], 0u };
}
__global__ void narrow_kernel(const S2 *in, unsigned n, unsigned *out)
{
   const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) out[i] = in[i].a;
}

// TriangleMesh.computeNormals on device arrays; d_normals: n_vertices * 3
hipError_t device_normals(const float *d_vertices, const int32_t *d_indices, uint32_t n_vertices, uint32_t n_faces, float *d_normals)
{
   DevVec<unsigned> count, offset, inc;
   U1Scan sc;
   GB_TRY(count.ensure((size_t) n_vertices + 1, 0));
   GB_TRY(offset.ensure((size_t) n_vertices + 1, 0));
   GB_TRY(inc.ensure(std::max<size_t>(3 * (size_t) n_faces, 1), 0));
   GB_TRY(sc.in.ensure((size_t) n_vertices + 1, 0));
   GB_TRY(sc.out.ensure((size_t) n_vertices + 2, 0));
   GB_TRY(hipMemsetAsync(count.p, 0, sizeof(unsigned) * ((size_t) n_vertices + 1), nullptr));
   if (n_faces) hipLaunchKernelGGL(normal_count_kernel, dim3(grid_for(3 * (size_t) n_faces)), dim3(256), 0, nullptr, d_indices, n_faces, count.p);
   hipLaunchKernelGGL(widen_kernel, dim3(grid_for(n_vertices)), dim3(256), 0, nullptr, count.p, n_vertices, sc.in.p);
   GB_TRY(exclusive_scan<S2>(sc.in.p, n_vertices, sc.out.p, sc.bsum));
   hipLaunchKernelGGL(narrow_kernel, dim3(grid_for((size_t) n_vertices + 1)), dim3(256), 0, nullptr, sc.out.p, n_vertices + 1, offset.p);
   GB_TRY(hipMemsetAsync(count.p, 0, sizeof(unsigned) * ((size_t) n_vertices + 1), nullptr));   // now the fill cursors
   if (n_faces) hipLaunchKernelGGL(normal_fill_kernel, dim3(grid_for(3 * (size_t) n_faces)), dim3(256), 0, nullptr, d_indices, n_faces, offset.p, count.p, inc.p);
   if (n_vertices) hipLaunchKernelGGL(normal_sum_kernel, dim3(grid_for(n_vertices)), dim3(256), 0, nullptr, d_vertices, d_indices, n_vertices, offset.p, inc.p, d_normals);
   GB_TRY(hipGetLastError());
   return hipDeviceSynchronize();   // the temporaries go out of scope
}

double seconds_since(std::chrono::steady_clock::time_point &t)
{
   const auto now = std::chrono::steady_clock::now();
   const double s = std::chrono::duration<double>(now - t).count();
   t = now;
   return s;
}

}   // namespace

hipError_t gpu_compute_normals(const float *vertices, uint32_t n_vertices, const int32_t *indices, uint32_t n_faces, float *normals_out)
{
   DevVec<float> d_vertices, d_normals;
   DevVec<int32_t> d_indices;
   GB_TRY(d_vertices.ensure(std::max<size_t>(3 * (size_t) n_vertices, 1), 0));
   GB_TRY(d_normals.ensure(std::max<size_t>(3 * (size_t) n_vertices, 1), 0));
   GB_TRY(d_indices.ensure(std::max<size_t>(3 * (size_t) n_faces, 1), 0));
   if (n_vertices) GB_TRY(hipMemcpy(d_vertices.p, vertices, sizeof(float) * 3 * (size_t) n_vertices, hipMemcpyHostToDevice));
   if (n_faces) GB_TRY(hipMemcpy(d_indices.p, indices, sizeof(int32_t) * 3 * (size_t) n_faces, hipMemcpyHostToDevice));
   GB_TRY(device_normals(d_vertices.p, d_indices.p, n_vertices, n_faces, d_normals.p));
   if (n_vertices) GB_TRY(hipMemcpy(normals_out, d_normals.p, sizeof(float) * 3 * (size_t) n_vertices, hipMemcpyDeviceToHost));
   return hipSuccess;
}

hipError_t gpu_build_object(const float *vertices, uint32_t n_vertices, const float *normals, const int32_t *indices, uint32_t n_faces,
                            const float bmin[3], const float bmax[3], const BuildOptions &opts, const PartBases &bases, GpuBuiltObject &out,
                            std::string &err)
{
   out = GpuBuiltObject();
   if (!opts.use_kdtree || n_faces <= (uint32_t) opts.max_children || opts.max_depth < 1 || n_faces > 0x3fffffffu) {
      err = "gpu_build_object: the root must split";
      return hipErrorInvalidValue;
   }
   auto t_mark = std::chrono::steady_clock::now();
   DevVec<float> d_vertices, d_fb, d_normals;
   DevVec<int32_t> d_indices;
   DevVec<unsigned> ent_face[2], ent_node[2], store_face, store_node;
   DevVec<BNode> nodes;
   DevVec<S2> flags, scan_e, bsum2;
   DevVec<S4> contrib, scan_n, bsum4;
   GB_TRY(d_vertices.ensure(3 * (size_t) n_vertices, 0));
   GB_TRY(d_indices.ensure(3 * (size_t) n_faces, 0));
   GB_TRY(d_fb.ensure(6 * (size_t) n_faces, 0));
   GB_TRY(hipMemcpy(d_vertices.p, vertices, sizeof(float) * 3 * (size_t) n_vertices, hipMemcpyHostToDevice));
   GB_TRY(hipMemcpy(d_indices.p, indices, sizeof(int32_t) * 3 * (size_t) n_faces, hipMemcpyHostToDevice));
   {   // the vertex normals: the mesh's own, or TriangleMesh.computeNormals; then the longest, and three per face
      GB_TRY(d_normals.ensure(3 * (size_t) n_vertices, 0));
      if (normals) GB_TRY(hipMemcpy(d_normals.p, normals, sizeof(float) * 3 * (size_t) n_vertices, hipMemcpyHostToDevice));
      else GB_TRY(device_normals(d_vertices.p, d_indices.p, n_vertices, n_faces, d_normals.p));
      DevVec<unsigned> red;
      GB_TRY(red.ensure(2, 0));
      GB_TRY(hipMemsetAsync(red.p, 0, 2 * sizeof(unsigned), nullptr));
      hipLaunchKernelGGL(normal_max_kernel, dim3(grid_for(n_vertices)), dim3(256), 0, nullptr, d_normals.p, n_vertices, red.p, red.p + 1);
      unsigned r[2] = { 0u, 0u };
      GB_TRY(hipMemcpy(r, red.p, sizeof(r), hipMemcpyDeviceToHost));
      union {
         unsigned u;
         float f;
      } c;
      c.u = r[0];
      out.max_normal_len = r[1] ? __builtin_nanf("") : c.f;
      GB_TRY(hipMalloc((void **) &out.fnrm, sizeof(float) * 12 * (size_t) n_faces));
      hipLaunchKernelGGL(face_normals_kernel, dim3(grid_for(n_faces)), dim3(256), 0, nullptr, d_normals.p, d_indices.p, n_faces, out.fnrm);
      GB_TRY(hipGetLastError());
   }
   GB_TRY(hipDeviceSynchronize());
   out.seconds[0] = seconds_since(t_mark);
   GB_TRY(ent_face[0].ensure(n_faces, 0));
   GB_TRY(ent_node[0].ensure(n_faces, 0));
   GB_TRY(nodes.ensure(1024, 0));
   hipLaunchKernelGGL(face_bounds_kernel, dim3(grid_for(n_faces)), dim3(256), 0, nullptr, d_vertices.p, d_indices.p, n_faces, d_fb.p, ent_face[0].p,
                      ent_node[0].p);
   hipLaunchKernelGGL(root_kernel, dim3(1), dim3(1), 0, nullptr, nodes.p, n_faces, bmin[0], bmin[1], bmin[2], bmax[0], bmax[1], bmax[2]);
   GB_TRY(hipGetLastError());

   std::vector<unsigned> level_off { 0u }, level_n { 1u };
   size_t n_entries = n_faces, n_all = 1, store_used = 0;
   int cur = 0;
   for (int depth = 0;; depth++) {
      const unsigned off = level_off.back(), n_level = level_n.back();
      const int can_split = depth < opts.max_depth ? 1 : 0;
      hipLaunchKernelGGL(level_split_kernel, dim3(grid_for(n_level)), dim3(256), 0, nullptr, nodes.p, off, n_level, can_split, (unsigned) opts.max_children);
      GB_TRY(flags.ensure(n_entries, 0));
      GB_TRY(scan_e.ensure(n_entries + 1, 0));
      hipLaunchKernelGGL(level_classify_kernel, dim3(grid_for(n_entries)), dim3(256), 0, nullptr, nodes.p, off, ent_face[cur].p, ent_node[cur].p,
                         (unsigned) n_entries, d_fb.p, depth % 3, flags.p);   // RTStatics.java:50-52: X -> Y -> Z
      GB_TRY(exclusive_scan<S2>(flags.p, n_entries, scan_e.p, bsum2));
      GB_TRY(contrib.ensure(n_level, 0));
      GB_TRY(scan_n.ensure((size_t) n_level + 1, 0));
      hipLaunchKernelGGL(level_count_kernel, dim3(grid_for(n_level)), dim3(256), 0, nullptr, nodes.p, off, n_level, scan_e.p, contrib.p);
      GB_TRY(exclusive_scan<S4>(contrib.p, n_level, scan_n.p, bsum4));
      S4 tot;
      GB_TRY(hipMemcpy(&tot, scan_n.p + n_level, sizeof(S4), hipMemcpyDeviceToHost));   // children, next entries, leaf entries, splits
      if ((size_t) tot.b > 0x7ffffff0u || store_used + tot.c > 0x7ffffff0u || n_all + tot.a > 0x7ffffff0u) {
         err = "hierarchy too large: leaf slots exceed 32-bit addressing";
         gpu_build_free(out);
         return hipErrorInvalidValue;
      }
      const int nxt = cur ^ 1;
      GB_TRY(nodes.ensure(n_all + tot.a, n_all));
      GB_TRY(ent_face[nxt].ensure(tot.b, 0));
      GB_TRY(ent_node[nxt].ensure(tot.b, 0));
      GB_TRY(store_face.ensure(store_used + tot.c, store_used));
      GB_TRY(store_node.ensure(store_used + tot.c, store_used));
      const unsigned next_off = (unsigned) n_all;
      hipLaunchKernelGGL(level_children_kernel, dim3(grid_for(n_level)), dim3(256), 0, nullptr, nodes.p, off, n_level, next_off, scan_n.p,
                         (unsigned) store_used);
      hipLaunchKernelGGL(level_scatter_kernel, dim3(grid_for(n_entries)), dim3(256), 0, nullptr, nodes.p, off, next_off, ent_face[cur].p,
                         ent_node[cur].p, (unsigned) n_entries, scan_e.p, d_fb.p, ent_face[nxt].p, ent_node[nxt].p, store_face.p, store_node.p);
      GB_TRY(hipGetLastError());
      store_used += tot.c;
      if (tot.a == 0) break;
      level_off.push_back(next_off);
      level_n.push_back(tot.a);
      n_all += tot.a;
      n_entries = tot.b;
      cur = nxt;
   }

   for (int l = (int) level_off.size() - 1; l >= 0; l--)
      hipLaunchKernelGGL(subtree_sizes_kernel, dim3(grid_for(level_n[l])), dim3(256), 0, nullptr, nodes.p, level_off[l], level_n[l]);
   for (size_t l = 0; l < level_off.size(); l++)
      hipLaunchKernelGGL(dfs_numbers_kernel, dim3(grid_for(level_n[l])), dim3(256), 0, nullptr, nodes.p, level_off[l], level_n[l]);
   BNode root;
   GB_TRY(hipMemcpy(&root, nodes.p, sizeof(BNode), hipMemcpyDeviceToHost));
   if (root.n_slots != store_used || !root.split) {
      err = "gpu_build_object: internal inconsistency (leaf store does not match the hierarchy)";
      gpu_build_free(out);
      return hipErrorUnknown;
   }
   out.seconds[1] = seconds_since(t_mark);
   out.n_nodes = root.n_int;
   out.n_slots = root.n_slots;
   out.n_leaves = root.n_leaves;
   out.max_depth = (int32_t) level_off.size() - 1;
   // the temporaries of the levels are not needed any more: the grouping pass wants the room (a 28-million-face mesh: GBs)
   for (int k = 0; k < 2; k++) {
      ent_face[k].release();
      ent_node[k].release();
   }
   flags.release();
   scan_e.release();
   d_fb.release();

   // ---- leaf slots, then the leaf groups (descriptors and group faces), then the records
   const size_t n_slots = store_used;
   DevVec<unsigned> slot_leaf, qual;
   DevVec<S2> cnt, base, bsum_g;
   DevVec<unsigned long long> totals;
   DevVec<int32_t> gface_ids;
   hipError_t e = hipMalloc((void **) &out.slot_ids, sizeof(int32_t) * std::max<size_t>(n_slots, 1));
   if (e == hipSuccess) e = slot_leaf.ensure(std::max<size_t>(n_slots, 1), 0);
   if (e == hipSuccess) e = qual.ensure(n_all, 0);
   if (e == hipSuccess) e = cnt.ensure(n_slots + 1, 0);
   if (e == hipSuccess) e = base.ensure(n_slots + 2, 0);
   if (e == hipSuccess) e = totals.ensure(2, 0);
   if (e == hipSuccess) e = hipMemsetAsync(cnt.p, 0, sizeof(S2) * (n_slots + 1), nullptr);
   if (e == hipSuccess) e = hipMemsetAsync(totals.p, 0, 2 * sizeof(unsigned long long), nullptr);
   if (e != hipSuccess) {
      gpu_build_free(out);
      return e;
   }
   hipLaunchKernelGGL(emit_slots_kernel, dim3(grid_for(store_used)), dim3(256), 0, nullptr, nodes.p, store_face.p, store_node.p, (unsigned) store_used,
                      out.slot_ids, slot_leaf.p);
   const unsigned group_grid = (unsigned) ((n_all + GROUP_WAVES - 1) / GROUP_WAVES);
   hipLaunchKernelGGL(group_qualify_kernel, dim3(group_grid), dim3(64 * GROUP_WAVES), 0, nullptr, nodes.p, (unsigned) n_all, out.slot_ids, qual.p);
   hipLaunchKernelGGL(group_count_kernel, dim3(grid_for(n_all)), dim3(256), 0, nullptr, nodes.p, (unsigned) n_all, qual.p, cnt.p, totals.p);
   e = hipGetLastError();
   if (e == hipSuccess) e = exclusive_scan<S2>(cnt.p, n_slots, base.p, bsum_g);
   S2 tot_g { 0u, 0u };
   unsigned long long tot_groups[2] = { 0ull, 0ull };
   if (e == hipSuccess) e = hipMemcpy(&tot_g, base.p + n_slots, sizeof(S2), hipMemcpyDeviceToHost);
   if (e == hipSuccess) e = hipMemcpy(tot_groups, totals.p, sizeof(tot_groups), hipMemcpyDeviceToHost);
   if (e == hipSuccess) {
      out.n_descs = tot_g.a;
      out.n_gfaces = tot_g.b;
      out.n_groups = tot_groups[0];
      out.n_shared = tot_groups[1];
      e = hipMalloc((void **) &out.descs, sizeof(LeafDesc) * std::max<size_t>(out.n_descs, 1));
   }
   if (e == hipSuccess) e = hipMemsetAsync(out.descs, 0, sizeof(LeafDesc) * std::max<size_t>(out.n_descs, 1), nullptr);
   if (e == hipSuccess) e = gface_ids.ensure(std::max<size_t>(out.n_gfaces, 1), 0);
   if (e == hipSuccess) e = hipMalloc((void **) &out.gfaces, sizeof(float) * 12 * std::max<size_t>(out.n_gfaces, 1));
   if (e == hipSuccess) e = hipMalloc((void **) &out.nodes, sizeof(DevNodeRec) * (size_t) out.n_nodes);
   if (e != hipSuccess) {
      gpu_build_free(out);
      return e;
   }
   hipLaunchKernelGGL(group_emit_kernel, dim3(group_grid), dim3(64 * GROUP_WAVES), 0, nullptr, nodes.p, (unsigned) n_all, qual.p, out.slot_ids, slot_leaf.p,
                      base.p, bases, out.descs, gface_ids.p);
   hipLaunchKernelGGL(group_alone_kernel, dim3(grid_for(n_all)), dim3(256), 0, nullptr, nodes.p, (unsigned) n_all, qual.p, out.slot_ids, base.p, bases,
                      out.descs, gface_ids.p);
   if (out.n_gfaces)
      hipLaunchKernelGGL(emit_gfaces_kernel, dim3(grid_for(out.n_gfaces)), dim3(256), 0, nullptr, d_vertices.p, d_indices.p, gface_ids.p, (size_t) out.n_gfaces,
                         out.gfaces);
   e = hipGetLastError();
   if (e == hipSuccess) e = hipDeviceSynchronize();
   out.seconds[2] = seconds_since(t_mark);
   if (e == hipSuccess) {
      hipLaunchKernelGGL(emit_nodes_kernel, dim3(grid_for(n_all)), dim3(256), 0, nullptr, nodes.p, (unsigned) n_all, base.p, (unsigned) bases.node,
                         (unsigned) bases.desc, out.nodes);
      hipLaunchKernelGGL(leaf_words_kernel, dim3(grid_for(out.n_nodes)), dim3(256), 0, nullptr, out.nodes, (unsigned) out.n_nodes, out.descs,
                         (unsigned) bases.desc);
      e = hipGetLastError();
   }
   if (e == hipSuccess) e = hipDeviceSynchronize();
   out.seconds[3] = seconds_since(t_mark);
   if (e != hipSuccess) gpu_build_free(out);
   return e;
}

hipError_t gpu_emit_gfaces(const float *vertices, uint32_t n_vertices, const int32_t *indices, uint32_t n_faces, const int32_t *gface, size_t n,
                           float *dst_dev)
{
   if (!n) return hipSuccess;
   DevVec<float> d_vertices;
   DevVec<int32_t> d_indices, d_gface;
   GB_TRY(d_vertices.ensure(3 * (size_t) n_vertices, 0));
   GB_TRY(d_indices.ensure(3 * (size_t) n_faces, 0));
   GB_TRY(d_gface.ensure(n, 0));
   GB_TRY(hipMemcpy(d_vertices.p, vertices, sizeof(float) * 3 * (size_t) n_vertices, hipMemcpyHostToDevice));
   GB_TRY(hipMemcpy(d_indices.p, indices, sizeof(int32_t) * 3 * (size_t) n_faces, hipMemcpyHostToDevice));
   GB_TRY(hipMemcpy(d_gface.p, gface, sizeof(int32_t) * n, hipMemcpyHostToDevice));
   hipLaunchKernelGGL(emit_gfaces_kernel, dim3(grid_for(n)), dim3(256), 0, nullptr, d_vertices.p, d_indices.p, d_gface.p, n, dst_dev);
   hipError_t e = hipGetLastError();
   if (e == hipSuccess) e = hipDeviceSynchronize();   // the temporaries go out of scope
   return e;
}

}   // namespace mi355rt
