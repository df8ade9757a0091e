// walk_lds.hip -- the walker with its rays resident in LDS and one work list per phase.
//
// trace_kernels.hip's walker keeps one ray per lane; a wave's 64 lanes are in different phases of different rays (some
// in internal nodes, some in leaves, some between rays), and every instruction of a phase is issued for the whole wave
// with the other lanes masked off: 35-45 % of the lanes of an average instruction do work (profiles/), and the kernel is
// bound by instruction issue.  Here a workgroup owns WL_SLOTS rays whose whole state -- ray, cursor, stack, best so far --
// lives in LDS, and a step of a ray is an ENTRY of a list: node steps, leaf steps, faces to work out (resolve), and
// bookkeeping (admin: object ends, results, new rays).  Each round ("super-step", one barrier) the four waves take the
// lists 64 entries at a time, so every instruction of the node step runs for 64 rays that are all at a node, and
// likewise for leaves; a ray that waits (for enough company in the resolve or admin list) costs nothing while it waits.
// What a step computes is exactly what the lane walker computes (same tests, same order-independent fold: the header
// of trace_kernels.hip), so the answers are the same bits.
//
// Lists are triple-buffered: round i reads buffer i % 3, appends to (i + 1) % 3 and clears the counts of (i + 2) % 3.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "device_util.h"
#include "kernels.h"
#include "layout.h"
#include "walk_args.h"

namespace mi355rt {

namespace {

#ifndef WL_THREADS_N
#define WL_THREADS_N 512
#endif
constexpr int WL_THREADS = WL_THREADS_N, WL_WAVES = WL_THREADS / 64;
constexpr int WL_SLOTS = 256;   // rays resident per workgroup (slot ids fit a byte)
#ifndef WL_WAVES_PER_SIMD
#define WL_WAVES_PER_SIMD 6     // 50 KB of LDS per workgroup: three workgroups of eight waves fit a CU
#endif
#ifndef WL_RESOLVE_MIN
#define WL_RESOLVE_MIN 24
#endif
#ifndef WL_ADMIN_MIN
#define WL_ADMIN_MIN 32
#endif
#ifndef WL_NODE_STAY
#define WL_NODE_STAY 32   // a node chunk goes round again while this many of its rays are at a node again
#endif
#ifndef WL_LEAF_STAY
#define WL_LEAF_STAY 32
#endif
#ifndef WL_BUSY_MIN
#define WL_BUSY_MIN 96
#endif
#ifndef WL_STACK
#define WL_STACK STACK_SLOTS
#endif
enum { L_NODE = 0, L_LEAF = 1, L_RESOLVE = 2, L_ADMIN = 3, N_LISTS = 4, L_NONE = 4 };

// per-slot flag word
constexpr unsigned F_OWNER = 1u, F_PLAIN = 2u, F_SHADOW = 4u, F_REDO = 8u, F_OCCLUDED = 16u;
constexpr int F_K = 8, F_MIRROR = 15, F_BEST = 22;   // 7-bit fields holding value + 1 (-1 .. MAX_OBJECTS - 1)
static_assert(MAX_OBJECTS <= 64, "object indices are 7-bit fields");
__device__ __forceinline__ int fl_get(unsigned fl, int shift) { return (int) ((fl >> shift) & 127u) - 1; }
__device__ __forceinline__ unsigned fl_set(unsigned fl, int shift, int v) { return (fl & ~(127u << shift)) | ((unsigned) (v + 1) << shift); }

// the fold of KDNode.getChildIntersection, order independent (trace_kernels.hip `beats`)
__device__ __forceinline__ bool wl_beats(const SceneArgs &sc, int ob_gf, float ob_w, int ob_key, float dist, int key, int gfx)
{
   if (ob_gf < 0 || dist < ob_w) return true;
   if (dist != ob_w) return false;
   if (key != ob_key) return key > ob_key;
   if (((gfx ^ ob_gf) & DESC_BASE_MASK) == 0) return false;
   if (gfx & DESC_IDENT) return (gfx & DESC_BASE_MASK) < (ob_gf & DESC_BASE_MASK);
   const float4 *gfaces = (const float4 *) sc.gfaces;
   const int f_new = __float_as_int(gfaces[3 * (size_t) (gfx & DESC_BASE_MASK)].w), f_old = __float_as_int(gfaces[3 * (size_t) (ob_gf & DESC_BASE_MASK)].w);
   for (int j = 0; j < 2 * GROUP_FACES; j++) {
      const int f = sc.slot_ids[key + j];
      if (f == f_new) return true;
      if (f == f_old) return false;
   }
   return false;
}

template <int STATS, int MODE>
__global__ __launch_bounds__(WL_THREADS) __attribute__((amdgpu_waves_per_eu(WL_WAVES_PER_SIMD, WL_WAVES_PER_SIMD))) void walk_lds_kernel(const WalkArgs a)
{
   __shared__ float s_o[3][WL_SLOTS], s_d[3][WL_SLOTS], s_inv[3][WL_SLOTS];
   __shared__ float s_lim2[WL_SLOTS], s_obw[WL_SLOTS], s_bestw[WL_SLOTS], s_safew[WL_SLOTS];
   __shared__ int s_obgf[WL_SLOTS], s_obkey[WL_SLOTS], s_bestgf[WL_SLOTS], s_cur[WL_SLOTS], s_sp[WL_SLOTS], s_gb[WL_SLOTS], s_pend[WL_SLOTS], s_pkey[WL_SLOTS];
   __shared__ unsigned s_missed[WL_SLOTS], s_todo[WL_SLOTS], s_qi[WL_SLOTS], s_flags[WL_SLOTS];
   __shared__ int s_stack[WL_STACK * WL_SLOTS];
   __shared__ unsigned char s_list[3][N_LISTS][WL_SLOTS];
   __shared__ unsigned long long s_cnt[3];   // four 16-bit counts per buffer
   __shared__ int s_dry;

   constexpr bool CLOSEST = MODE == MODE_CLOSEST || MODE == MODE_FUSED;
   constexpr bool FUSED = MODE == MODE_FUSED;
   const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
   uint32_t *cursors = a.counters + a.cursor_index;
   const float4 *__restrict__ nodes = (const float4 *) a.sc.nodes;
   const float4 *__restrict__ gfaces = (const float4 *) a.sc.gfaces;
   const float4 *__restrict__ descs = (const float4 *) a.sc.descs;
   const float nearp = a.sc.near_plane, farp = a.sc.far_plane, eps = a.sc.epsilon;
   const V3 light = mk3(a.light[0], a.light[1], a.light[2]);
   const int n_objs = a.sc.n_objs;
   const float abs_max = a.sc.scene_abs_max;
   const int home = (int) (blockIdx.x % (unsigned) N_BANDS);
   int attempt = 0;          // per wave: the band this wave claims rays from
   bool exhausted = false;   // per wave: every band was found empty
   unsigned n_box = 0, n_tri = 0, n_hit = 0, n_ray = 0, n_spawn = 0, n_box_s = 0, n_tri_s = 0;
   unsigned d_rounds = 0, d_chunks[N_LISTS] = { 0, 0, 0, 0 }, d_lanes[N_LISTS] = { 0, 0, 0, 0 };   // diagnostics (STATS): per wave
   const unsigned long long t_begin = STATS ? wall_clock64() : 0ull;   // 100 MHz
   unsigned long long t_exhausted = 0ull;
#ifdef WL_PROFILE
   // where a round's time goes (shader clock ticks, per wave): [0] barrier wait, [1] LDS state read, [2] node / leaf data arrival,
   // [3] node / leaf compute + state write, [4] list append, [5] whole resolve chunks, [6] whole admin chunks
   unsigned long long pt[7] = { 0, 0, 0, 0, 0, 0, 0 };
   unsigned long long pc = clock64();
#define WL_TICK(i) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long now_ = clock64(); pt[i] += now_ - pc; pc = now_; } while (0)
#else
#define WL_TICK(i) do { } while (0)
#endif

   if (tid < WL_SLOTS) {
      s_cur[tid] = STACK_EMPTY;
      s_flags[tid] = 0u;
      s_pend[tid] = -1;
      s_list[0][L_ADMIN][tid] = (unsigned char) tid;
   }
   if (tid == 0) {
      s_cnt[0] = (unsigned long long) WL_SLOTS << (16 * L_ADMIN);
      s_cnt[1] = 0ull;
      s_cnt[2] = 0ull;
      s_dry = 0;
   }
   int cb = 0;

   for (;;) {
      WL_TICK(4);
      __syncthreads();
      WL_TICK(0);
      const unsigned long long cw = s_cnt[cb];
      const int dry = s_dry;
      const int nb = cb == 2 ? 0 : cb + 1, zb = nb == 2 ? 0 : nb + 1;
      if (cw == 0ull) break;
      if (tid == 0) s_cnt[zb] = 0ull;
      if (STATS) {
         d_rounds++;
         if (dry && !t_exhausted) t_exhausted = wall_clock64();
      }
      const int n0 = (int) (cw & 0xffffu), n1 = (int) ((cw >> 16) & 0xffffu), n2 = (int) ((cw >> 32) & 0xffffu), n3 = (int) ((cw >> 48) & 0xffffu);
      // faces to work out and bookkeeping wait for company while the walk keeps the waves busy
      const bool busy = n0 + n1 >= WL_BUSY_MIN;
      const bool defer_res = busy && n2 > 0 && n2 < WL_RESOLVE_MIN, defer_adm = busy && n3 > 0 && n3 < WL_ADMIN_MIN;
      const int c0 = (n0 + 63) >> 6, c1 = (n1 + 63) >> 6, c2 = defer_res ? 0 : (n2 + 63) >> 6, c3 = defer_adm ? 0 : (n3 + 63) >> 6;
      if (defer_res && wave == WL_WAVES - 1) {
         unsigned long long old = 0ull;
         if (lane == 0) old = atomicAdd(&s_cnt[nb], (unsigned long long) n2 << (16 * L_RESOLVE));
         const int base = (__builtin_amdgcn_readfirstlane((int) (old >> 32))) & 0xffff;
         if (lane < n2) s_list[nb][L_RESOLVE][base + lane] = s_list[cb][L_RESOLVE][lane];
      }
      if (defer_adm && wave == WL_WAVES - 2) {
         unsigned long long old = 0ull;
         if (lane == 0) old = atomicAdd(&s_cnt[nb], (unsigned long long) n3 << (16 * L_ADMIN));
         const int base = (__builtin_amdgcn_readfirstlane((int) (old >> 48))) & 0xffff;
         if (lane < n3) s_list[nb][L_ADMIN][base + lane] = s_list[cb][L_ADMIN][lane];
      }
      const int total = c0 + c1 + c2 + c3;
      for (int c = wave; c < total; c += WL_WAVES) {
         int type, ci, n;
         if (c < c1) {   // leaf chunks are the longest: they go first
            type = L_LEAF;
            ci = c;
            n = n1;
         } else if (c < c1 + c0) {
            type = L_NODE;
            ci = c - c1;
            n = n0;
         } else if (c < c1 + c0 + c2) {
            type = L_RESOLVE;
            ci = c - c1 - c0;
            n = n2;
         } else {
            type = L_ADMIN;
            ci = c - c1 - c0 - c2;
            n = n3;
         }
         const int idx = ci * 64 + lane;
         const bool act = idx < n;
         const int slot = (int) s_list[cb][type][act ? idx : 0];
         int next = L_NONE;
         if (STATS) {
#pragma unroll
            for (int t = 0; t < N_LISTS; t++) {
               d_chunks[t] += type == t ? 1u : 0u;
               d_lanes[t] += (type == t && act) ? 1u : 0u;
            }
         }

         if (type == L_NODE) {
            // both children of one internal node (trace_kernels.hip (C)).  A ray that steps from a node into a node stays
            // in this wave's registers for its next step, for as long as enough of the chunk's rays do the same.
            Ray r;
            r.o = mk3(s_o[0][slot], s_o[1][slot], s_o[2][slot]);
            r.inv = mk3(s_inv[0][slot], s_inv[1][slot], s_inv[2][slot]);
            r.d = r.o;   // not used by the box tests
            const float lim2 = s_lim2[slot];
            int cur = s_cur[slot], sp = s_sp[slot], gb = s_gb[slot];
            unsigned missed = s_missed[slot], todo = 0u;
            const unsigned fl = s_flags[slot];
            bool on = act;
            do {
            if (on) {
               const float4 *pa = nodes + NODE_F4 * (size_t) cur;
               WL_TICK(1);
               const float4 na = pa[0], nb4 = pa[1], nc = pa[2], q3 = pa[3], q4 = pa[4];
               WL_TICK(2);
               const int4 nd = make_int4(__float_as_int(q3.x), __float_as_int(q3.y), __float_as_int(q3.z), __float_as_int(q3.w));
               float tl = 0.0f, tr = 0.0f;
               bool lok = false, rok = false;
               if (STATS) ((FUSED && (fl & F_SHADOW)) ? n_box_s : n_box) += (nd.x != LINK_NONE ? 1u : 0u) + (nd.y != LINK_NONE ? 1u : 0u);
               if (fl & F_PLAIN) {
                  const f2 ax = f2 { na.x, na.y } - r.o.x, ay = f2 { na.z, na.w } - r.o.y, az = f2 { nb4.x, nb4.y } - r.o.z;
                  const f2 bx = f2 { nb4.z, nb4.w } - r.o.x, by = f2 { nc.x, nc.y } - r.o.y, bz = f2 { nc.z, nc.w } - r.o.z;
                  const f2 t1x = ax * r.inv.x, t1y = ay * r.inv.y, t1z = az * r.inv.z;
                  const f2 t2x = bx * r.inv.x, t2y = by * r.inv.y, t2z = bz * r.inv.z;
                  const f2 ddx = { fmaxf(fmaxf(ax.x, -bx.x), 0.0f), fmaxf(fmaxf(ax.y, -bx.y), 0.0f) };
                  const f2 ddy = { fmaxf(fmaxf(ay.x, -by.x), 0.0f), fmaxf(fmaxf(ay.y, -by.y), 0.0f) };
                  const f2 ddz = { fmaxf(fmaxf(az.x, -bz.x), 0.0f), fmaxf(fmaxf(az.y, -bz.y), 0.0f) };
                  const f2 d2 = ddx * ddx + ddy * ddy + ddz * ddz;
                  tl = fmaxf(fmaxf(fminf(t1x.x, t2x.x), fminf(t1y.x, t2y.x)), fminf(t1z.x, t2z.x));
                  tr = fmaxf(fmaxf(fminf(t1x.y, t2x.y), fminf(t1y.y, t2y.y)), fminf(t1z.y, t2z.y));
                  const float hl = fminf(fminf(fmaxf(t1x.x, t2x.x), fmaxf(t1y.x, t2y.x)), fmaxf(t1z.x, t2z.x));
                  const float hr = fminf(fminf(fmaxf(t1x.y, t2x.y), fmaxf(t1y.y, t2y.y)), fmaxf(t1z.y, t2z.y));
                  lok = nd.x != LINK_NONE && tl <= hl && tl < farp && hl > nearp && !(d2.x > lim2);
                  rok = nd.y != LINK_NONE && tr <= hr && tr < farp && hr > nearp && !(d2.y > lim2);
                  if (MODE == MODE_SHADOW && !(fl & F_REDO)) {
                     const f2 fx = { fmaxf(light.x - na.x, nb4.z - light.x), fmaxf(light.x - na.y, nb4.w - light.x) };
                     const f2 fy = { fmaxf(light.y - na.z, nc.x - light.y), fmaxf(light.y - na.w, nc.y - light.y) };
                     const f2 fz = { fmaxf(light.z - nb4.x, nc.z - light.z), fmaxf(light.z - nb4.y, nc.w - light.z) };
                     const f2 far2 = (fx * fx + fy * fy + fz * fz) * 1.0001f + a.light_prune_c;
                     lok = lok && !(d2.x > far2.x);
                     rok = rok && !(d2.y > far2.y);
                  }
               } else {
                  if (nd.x != LINK_NONE)
                     lok = slab_test(r, nearp, farp, na.x, na.z, nb4.x, nb4.z, nc.x, nc.z, tl) &&
                           !(origin_box_dist2(r, na.x, na.z, nb4.x, nb4.z, nc.x, nc.z) > lim2);
                  if (nd.y != LINK_NONE)
                     rok = slab_test(r, nearp, farp, na.y, na.w, nb4.y, nb4.w, nc.y, nc.w, tr) &&
                           !(origin_box_dist2(r, na.y, na.w, nb4.y, nb4.w, nc.y, nc.w) > lim2);
                  if (MODE == MODE_SHADOW && !(fl & F_REDO) && a.sc.boxes_finite) {
                     lok = lok && !(origin_box_dist2(r, na.x, na.z, nb4.x, nb4.z, nc.x, nc.z) > light_far2(light, na.x, na.z, nb4.x, nb4.z, nc.x, nc.z) * 1.0001f + a.light_prune_c);
                     rok = rok && !(origin_box_dist2(r, na.y, na.w, nb4.y, nb4.w, nc.y, nc.w) > light_far2(light, na.y, na.w, nb4.y, nb4.w, nc.y, nc.w) * 1.0001f + a.light_prune_c);
                  }
               }
               const int le = nd.z > 0 ? ~nd.x : nd.x, re = nd.w > 0 ? ~nd.y : nd.y;
               const int lg = __float_as_int(q4.x), rg = __float_as_int(q4.z);
               const unsigned ltodo = __float_as_uint(q4.y) & ~((((lg ^ gb) & DESC_BASE_MASK) == 0) ? missed : 0u);
               const unsigned rtodo = __float_as_uint(q4.w) & ~((((rg ^ gb) & DESC_BASE_MASK) == 0) ? missed : 0u);
               if (nd.z > 0 && ltodo == 0u && !(lg & DESC_MORE)) lok = false;
               if (nd.w > 0 && rtodo == 0u && !(rg & DESC_MORE)) rok = false;
               int took = 0;
               if (lok && rok) {
                  const bool swap = tr < tl;
                  cur = swap ? re : le;
                  took = swap ? 2 : 1;
                  if (WL_STACK >= STACK_SLOTS || sp < WL_STACK) {   // (a shorter stack is a timing experiment only: it drops subtrees)
                     s_stack[sp * WL_SLOTS + slot] = swap ? le : re;
                     ++sp;
                  }
               } else if (lok) {
                  cur = le;
                  took = 1;
               } else if (rok) {
                  cur = re;
                  took = 2;
               } else if (sp > 0) {
                  --sp;
                  cur = s_stack[sp * WL_SLOTS + slot];
               } else {
                  cur = STACK_EMPTY;
               }
               todo = 0u;
               if (took != 0 && cur < 0) {
                  const int g = took == 1 ? lg : rg;
                  if (((g ^ gb) & DESC_BASE_MASK) != 0) missed = 0u;
                  gb = g;
                  todo = took == 1 ? ltodo : rtodo;
               }
               on = (unsigned) cur < (unsigned) STACK_EMPTY;
               WL_TICK(3);
            }
            } while ((int) __popcll(__ballot(on)) >= WL_NODE_STAY);
            if (act) {
               s_cur[slot] = cur;
               s_sp[slot] = sp;
               s_gb[slot] = gb;
               s_missed[slot] = missed;
               s_todo[slot] = todo;
               next = cur == STACK_EMPTY ? L_ADMIN : (cur < 0 ? L_LEAF : L_NODE);
            }
         } else if (type == L_LEAF) {
            // a leaf (trace_kernels.hip (D)): its descriptor, or two of its faces; a face that passes waits in the resolve
            // list.  A ray whose next step is a leaf step too stays here while enough of the chunk's rays do.
            Ray r;
            r.o = mk3(s_o[0][slot], s_o[1][slot], s_o[2][slot]);
            r.d = mk3(s_d[0][slot], s_d[1][slot], s_d[2][slot]);
            r.inv = r.d;   // not used by the triangle tests
            int cur = s_cur[slot], sp = s_sp[slot], gb = s_gb[slot];
            unsigned missed = s_missed[slot], todo = s_todo[slot];
            int pend = -1;
            bool on = act;
            do {
            if (on) {
               const bool want_desc = todo == 0u;
               const unsigned rest0 = todo & (todo - 1u);
               const size_t f0 = !want_desc ? (size_t) (gb & DESC_BASE_MASK) + (size_t) (__ffs((int) todo) - 1) : 0;
               const size_t f1 = rest0 != 0u ? (size_t) (gb & DESC_BASE_MASK) + (size_t) (__ffs((int) rest0) - 1) : f0;
               const float4 *pa = want_desc ? descs + (size_t) ~cur : gfaces + 3 * f0;
               const float4 *pb = gfaces + 3 * f1;
               WL_TICK(1);
               const float4 q0 = pa[0], q1 = pa[1], q2 = pa[2], q3 = pb[0], q4 = pb[1], q5 = pb[2];
               WL_TICK(2);
               if (want_desc) {
                  const int g = __float_as_int(q0.y);
                  const unsigned mask = __float_as_uint(q0.z);
                  if (((g ^ gb) & DESC_BASE_MASK) != 0) missed = 0u;
                  gb = g;
                  todo = mask & ~missed;
               } else {
                  const unsigned bit0 = todo & (0u - todo), bit1 = rest0 & (0u - rest0);
                  unsigned done = bit0 | bit1;
                  if (STATS) ((FUSED && (s_flags[slot] & F_SHADOW)) ? n_tri_s : n_tri) += bit1 ? 2u : 1u;
                  const bool m0 = tri_meets(q0, q1, q2, r, eps);
                  const bool m1 = bit1 != 0u && tri_meets(q3, q4, q5, r, eps);
                  missed |= (m0 ? 0u : bit0) | (m1 ? 0u : bit1);
                  if (m0 || m1) {
                     pend = ((gb & DESC_BASE_MASK) + __ffs((int) (m0 ? bit0 : bit1)) - 1) | (gb & DESC_IDENT);
                     s_pkey[slot] = ~cur;
                     s_pend[slot] = pend;
                     if (m0 && m1) done = bit0;   // the second one is met again once the first has been worked out
                  }
                  todo &= ~done;
               }
               if (todo == 0u) {
                  if (gb & DESC_MORE) {
                     cur -= 1;
                  } else if (sp > 0) {
                     --sp;
                     cur = s_stack[sp * WL_SLOTS + slot];
                  } else {
                     cur = STACK_EMPTY;
                  }
               }
               on = cur < 0 && pend < 0;
               WL_TICK(3);
            }
            } while ((int) __popcll(__ballot(on)) >= WL_LEAF_STAY);
            if (act) {
               s_cur[slot] = cur;
               s_sp[slot] = sp;
               s_gb[slot] = gb;
               s_missed[slot] = missed;
               s_todo[slot] = todo;
               next = pend >= 0 ? L_RESOLVE : (cur == STACK_EMPTY ? L_ADMIN : (cur < 0 ? L_LEAF : L_NODE));
            }
         } else if (type == L_RESOLVE) {
            if (act) {
               // a face that passed the triangle tests: its point and distance, folded into the object's best so far
               Ray r;
               r.o = mk3(s_o[0][slot], s_o[1][slot], s_o[2][slot]);
               r.d = mk3(s_d[0][slot], s_d[1][slot], s_d[2][slot]);
               r.inv = r.d;
               const int pend = s_pend[slot];
               const unsigned fl = s_flags[slot];
               const int k = fl_get(fl, F_K);
               int cur = s_cur[slot];
               const float4 *t = gfaces + 3 * (size_t) (pend & DESC_BASE_MASK);
               const float4 A = t[0], B = t[1], C = t[2];
               const int key = __float_as_int(descs[(size_t) s_pkey[slot]].x);
               const float sd = a.sc.objs[k].shadow_distance;
               float u, v, dist;
               V3 P;
               (void) tri_test(A, B, C, r, eps, u, v, P, dist);
               if ((dist > sd) && wl_beats(a.sc, s_obgf[slot], s_obw[slot], s_obkey[slot], dist, key, pend)) {
                  s_obw[slot] = dist;
                  s_obgf[slot] = pend;
                  s_obkey[slot] = key;
                  s_lim2[slot] = fminf(s_lim2[slot], prune_limit2(r, dist, abs_max));
                  // any accepted face occludes (any-hit forms), or one safely before half way to the light (device_util.h, shortcut 1): the walk ends here
                  const bool ends = (FUSED && (fl & F_SHADOW)) || MODE == MODE_SHADOW_ANY || (MODE == MODE_SHADOW && dist < s_safew[slot]);
                  if (ends) {
                     cur = STACK_EMPTY;
                     s_cur[slot] = cur;
                     s_sp[slot] = 0;
                     s_todo[slot] = 0u;
                  }
               }
               s_pend[slot] = -1;
               next = cur == STACK_EMPTY ? L_ADMIN : (cur < 0 ? L_LEAF : L_NODE);
            }
            WL_TICK(5);
         } else {
            // bookkeeping: a ray whose walk of object k has ended folds the object's result and enters the next object
            // or is finished (its record is written; MODE_FUSED: it goes on as its hit's shadow ray); free slots take
            // new rays from the queue
            unsigned fl = act ? s_flags[slot] : 0u;
            bool owner = act && (fl & F_OWNER) != 0u;
            int k = fl_get(fl, F_K), mirror = fl_get(fl, F_MIRROR), best_obj = fl_get(fl, F_BEST);
            Ray r;
            r.o = mk3(s_o[0][slot], s_o[1][slot], s_o[2][slot]);
            r.d = mk3(s_d[0][slot], s_d[1][slot], s_d[2][slot]);
            r.inv = mk3(s_inv[0][slot], s_inv[1][slot], s_inv[2][slot]);
            float best_w = s_bestw[slot];
            int best_gf = s_bestgf[slot];
            unsigned qi = s_qi[slot];
            bool occluded = (fl & F_OCCLUDED) != 0u;
            bool shadow = (fl & F_SHADOW) != 0u, redo = (fl & F_REDO) != 0u;
            bool need_enter = false, new_ray = false;
            // the next object of the ray, or none
            auto advance = [&]() -> bool {
               int nk = redo ? k : k + 1;
               if (CLOSEST && nk == mirror) ++nk;
               const bool done = (nk >= n_objs) || (MODE != MODE_CLOSEST && occluded);
               if (!done) k = nk;
               return done;
            };
            if (owner) {
               const float ob_w = s_obw[slot];
               const int ob_gf = s_obgf[slot];
               bool redo_k = false;
               if (k >= 0 && ob_gf >= 0) {
                  if (CLOSEST && !(FUSED && shadow)) {
                     if ((ob_w > eps) && (best_obj < 0 || ob_w < best_w)) {   // Camera.java:208
                        best_w = ob_w;
                        best_gf = ob_gf;
                        best_obj = k;
                     }
                  } else if (MODE == MODE_SHADOW_ANY || FUSED) {
                     occluded = true;
                  } else if (ob_w > eps) {   // RTStatics.java:628-633
                     if (ob_w < s_safew[slot]) {
                        occluded = true;
                     } else {
                        const float4 *t = gfaces + 3 * (size_t) (ob_gf & DESC_BASE_MASK);
                        float u, v, dist;
                        V3 P;
                        (void) tri_test(t[0], t[1], t[2], r, eps, u, v, P, dist);
                        if (ob_w < distance3(P, light) + eps) {
                           if (redo) occluded = true;
                           else redo_k = true;
                        }
                     }
                  }
               }
               redo = MODE == MODE_SHADOW && redo_k;
               if (!advance()) {
                  need_enter = true;
               } else if (CLOSEST && !(FUSED && shadow)) {
                  // the winner's point, barycentrics and normal, recomputed (same arithmetic, same bits), and the record
                  float4 h0 = make_float4(0, 0, 0, 0), h1 = make_float4(0, 0, 0, __int_as_float(-1));
                  if (best_obj >= 0) {
                     const float4 *t = gfaces + 3 * (size_t) (best_gf & DESC_BASE_MASK);
                     const float4 A = t[0], B = t[1], C = t[2];
                     float u = 0, v = 0, dist = 0;
                     V3 P = mk3(0, 0, 0);
                     (void) tri_test(A, B, C, r, eps, u, v, P, dist);
                     const int face = __float_as_int(A.w);
                     const float4 *fn = (const float4 *) a.sc.fnrm + 3 * ((size_t) a.sc.objs[best_obj].face_base + (size_t) face);
                     const float4 n0 = fn[0], n1 = fn[1], n2 = fn[2];
                     const float wb = 1.0f - u - v;   // Triangle.java:126-129, not renormalised
                     h0 = make_float4(P.x, P.y, P.z, dist);
                     h1 = make_float4(wb * n0.x + u * n1.x + v * n2.x, wb * n0.y + u * n1.y + v * n2.y, wb * n0.z + u * n1.z + v * n2.z,
                                      __int_as_float(face));
                     if (STATS) n_hit++;
                  }
                  float4 *hq = (float4 *) a.hits + 2 * (size_t) qi;
                  hq[0] = h0;
                  hq[1] = h1;
                  a.hit_obj[qi] = best_obj;
                  if (FUSED && best_obj >= 0) {
                     // the slot goes on with this hit's shadow ray (RTStatics.java:619-620, the Ray ctor, RTStatics.offsetPosition)
                     const V3 P = mk3(h0.x, h0.y, h0.z), N = mk3(h1.x, h1.y, h1.z);
                     make_ray(r, add3(P, scale3(N, eps)), normalize3(normalize3(sub3(light, P))));
                     new_ray = true;
                     shadow = true;
                     k = -1;
                     mirror = -1;
                     best_obj = -1;
                     best_gf = -1;
                     best_w = 0.0f;
                     occluded = false;
                     if (STATS) n_spawn++;
                     need_enter = !advance();
                  } else {
                     owner = false;
                  }
               } else {
                  a.mat0[4 * (size_t) qi + 3] = occluded ? 1.0f : 0.0f;
                  owner = false;
               }
            }

            // free slots take new rays (trace_kernels.hip (3): band cursors, shadow lists dealt out transposed)
            const bool want = act && !owner && !dry && !exhausted;
            const unsigned long long idle = __ballot(want);
            if (idle != 0ull) {
               const unsigned want_n = (unsigned) __popcll(idle);
               constexpr unsigned SPREAD = CLOSEST ? 1u : 64u;
               unsigned base = 0, rows = 0, count = 0, lo = 0;
               bool got = false;
               while (attempt < N_BANDS) {
                  const unsigned rg = (unsigned) (home + attempt) % (unsigned) N_BANDS;
                  lo = rg * a.band_cap;
                  count = a.counters[a.count_index + rg];
                  rows = (count + SPREAD - 1u) / SPREAD;
                  unsigned b = 0;
                  if (lane == 0) b = atomicAdd(cursors + rg, want_n);
                  b = (unsigned) __builtin_amdgcn_readfirstlane((int) b);
                  if (b < rows * SPREAD) {
                     base = b;
                     got = true;
                     break;
                  }
                  ++attempt;
               }
               if (!got) {
                  exhausted = true;
                  if (lane == 0) s_dry = 1;
               } else if (want) {
                  const unsigned logical = base + (unsigned) rank_below(idle);
                  const unsigned rel = (logical % SPREAD) * rows + logical / SPREAD;
                  const unsigned my = lo + rel;
                  const unsigned limit = logical < rows * SPREAD ? lo + count : 0u;
                  if (my < limit) {
                     qi = my;
                     V3 o, d;
                     if (CLOSEST) {
                        const float4 *q = (const float4 *) a.rays + 2 * (size_t) my;
                        const float4 r0 = q[0], r1 = q[1];
                        o = mk3(r0.x, r0.y, r0.z);
                        d = mk3(r1.x, r1.y, r1.z);
                        mirror = (int) (short) (__float_as_int(r1.w) & 0xffff);
                     } else {
                        qi = a.shadow[my];   // the path
                        const float4 *hq = (const float4 *) a.hits0 + 2 * (size_t) qi;
                        const float4 h0 = hq[0], h1 = hq[1];
                        const V3 P = mk3(h0.x, h0.y, h0.z), N = mk3(h1.x, h1.y, h1.z);
                        d = normalize3(normalize3(sub3(light, P)));   // RTStatics.java:619-620, then the Ray ctor
                        o = add3(P, scale3(N, eps));                  // RTStatics.offsetPosition
                        mirror = -1;
                     }
                     make_ray(r, o, d);
                     new_ray = true;
                     owner = true;
                     shadow = false;
                     redo = false;
                     k = -1;
                     best_obj = -1;
                     best_gf = -1;
                     best_w = 0.0f;
                     occluded = false;
                     if (STATS) n_ray++;
                     need_enter = !advance();
                  }
               }
            }

            if (act) {
               int cur = STACK_EMPTY;
               if (need_enter) {
                  const DevObject *ob = a.sc.objs + k;
                  s_obgf[slot] = -1;
                  s_obw[slot] = 0.0f;
                  s_obkey[slot] = 0;
                  s_sp[slot] = 0;
                  s_lim2[slot] = (CLOSEST && best_obj >= 0) ? prune_limit2(r, best_w, abs_max) : __builtin_inff();
                  if (MODE == MODE_SHADOW) s_safew[slot] = (!redo && ob->shadow_distance >= eps) ? shadow_safe_w(r, light, abs_max) : 0.0f;
                  float te;
                  if (STATS) ((FUSED && shadow) ? n_box_s : n_box)++;
                  if (slab_test(r, nearp, farp, ob->bmin[0], ob->bmin[1], ob->bmin[2], ob->bmax[0], ob->bmax[1], ob->bmax[2], te))
                     cur = ob->root_entry;   // KDTree.getChildIntersection (bounds/KDTree.java:59-65)
                  s_todo[slot] = 0u;
               }
               if (new_ray) {
                  s_o[0][slot] = r.o.x;
                  s_o[1][slot] = r.o.y;
                  s_o[2][slot] = r.o.z;
                  s_d[0][slot] = r.d.x;
                  s_d[1][slot] = r.d.y;
                  s_d[2][slot] = r.d.z;
                  s_inv[0][slot] = r.inv.x;
                  s_inv[1][slot] = r.inv.y;
                  s_inv[2][slot] = r.inv.z;
                  s_gb[slot] = -1;   // another ray: nothing is known to miss
                  s_missed[slot] = 0u;
                  s_qi[slot] = qi;
               }
               s_cur[slot] = cur;
               s_bestw[slot] = best_w;
               s_bestgf[slot] = best_gf;
               unsigned nf = (owner ? F_OWNER : 0u) | (shadow ? F_SHADOW : 0u) | (redo ? F_REDO : 0u) | (occluded ? F_OCCLUDED : 0u);
               if (owner && plain_ray(r) && a.sc.boxes_finite) nf |= F_PLAIN;
               nf = fl_set(fl_set(fl_set(nf, F_K, k), F_MIRROR, mirror), F_BEST, best_obj);
               s_flags[slot] = nf;
               // a free slot waits in this list for the next claim; once the queue is dry it has nothing left to do
               next = owner ? (cur == STACK_EMPTY ? L_ADMIN : (cur < 0 ? L_LEAF : L_NODE)) : ((dry || exhausted) ? L_NONE : L_ADMIN);
            }
            WL_TICK(6);
         }

         // every entry goes to the list of its next step: one LDS atomic per chunk (the four counts share a word)
         {
            const unsigned long long m0 = __ballot(next == L_NODE), m1 = __ballot(next == L_LEAF), m2 = __ballot(next == L_RESOLVE), m3 = __ballot(next == L_ADMIN);
            const unsigned long long add = (unsigned long long) __popcll(m0) | ((unsigned long long) __popcll(m1) << 16) |
                                           ((unsigned long long) __popcll(m2) << 32) | ((unsigned long long) __popcll(m3) << 48);
            unsigned long long old = 0ull;
            if (lane == 0 && add != 0ull) old = atomicAdd(&s_cnt[nb], add);
            const unsigned old_lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) old);
            const unsigned old_hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (old >> 32));
            if (next != L_NONE) {
               const unsigned long long mm = next == L_NODE ? m0 : (next == L_LEAF ? m1 : (next == L_RESOLVE ? m2 : m3));
               const unsigned half = next < 2 ? old_lo : old_hi;
               const unsigned base = (next & 1) ? (half >> 16) : (half & 0xffffu);
               s_list[nb][next][base + (unsigned) rank_below(mm)] = (unsigned char) slot;
            }
         }
         WL_TICK(4);
      }
      cb = nb;
   }

   if (STATS) {
      add_stat(a.counters, ST_BOX, n_box + n_box_s);
      add_stat(a.counters, ST_TRI, n_tri + n_tri_s);
      if (!CLOSEST) add_stat(a.counters, ST_SHADOW, n_ray);
      if (FUSED) add_stat(a.counters, ST_SHADOW, n_spawn);
      add_stat(a.counters, ST_HITS, a.stats == 2 ? n_hit : 0u);
      const int kind = !CLOSEST ? 2 : (a.stats == 2 ? 0 : 1);
      add_stat(a.counters, ST_KIND_RAYS + kind, n_ray);
      add_stat(a.counters, ST_KIND_BOX + kind, n_box);
      add_stat(a.counters, ST_KIND_TRI + kind, n_tri);
      if (FUSED) {
         add_stat(a.counters, ST_KIND_RAYS + 2, n_spawn);
         add_stat(a.counters, ST_KIND_BOX + 2, n_box_s);
         add_stat(a.counters, ST_KIND_TRI + 2, n_tri_s);
      }
      // diagnostics: entries and chunks of each list (lane use of a phase = entries / (64 x chunks)), rounds, timeline
      add_stat(a.counters, ST_NODE_LANE, d_lanes[L_NODE]);
      add_stat(a.counters, ST_NODE_WAVE, lane == 0 ? d_chunks[L_NODE] : 0u);
      add_stat(a.counters, ST_LEAF_LANE, d_lanes[L_LEAF]);
      add_stat(a.counters, ST_LEAF_WAVE, lane == 0 ? d_chunks[L_LEAF] : 0u);
      add_stat(a.counters, ST_OUTER_WAVE, lane == 0 ? d_rounds : 0u);
      add_stat(a.counters, ST_REFILL, d_lanes[L_ADMIN]);
      add_stat(a.counters, 12, lane == 0 ? d_chunks[L_ADMIN] : 0u);
      add_stat(a.counters, 63, d_lanes[L_RESOLVE]);
      add_stat(a.counters, 62, lane == 0 ? d_chunks[L_RESOLVE] : 0u);
      if (lane == 0) {
         const unsigned long long t_end = wall_clock64();
         unsigned long long *st = (unsigned long long *) (a.counters + CNT_STATS);
         atomicMax(st + 13, t_end - t_begin);
         atomicAdd(st + 14, t_end - t_begin);
         atomicAdd(st + 15, t_exhausted ? t_end - t_exhausted : 0ull);
         unsigned long long *tl = st + ST_KIND_TIMELINE + 6 * kind;
         atomicMax(tl + 0, ~t_begin);
         if (t_exhausted) {
            atomicMax(tl + 1, ~t_exhausted);
            atomicMax(tl + 2, t_exhausted);
         }
         atomicMax(tl + 3, t_end);
#ifdef WL_PROFILE
         for (int i = 0; i < 7; i++) atomicAdd(st + ST_TRIP_HIST + i, pt[i]);
#endif
      }
   }
}

template <int MODE> void launch_mode(const WalkArgs &a, unsigned grid, int stats, hipStream_t stream)
{
   if (stats) hipLaunchKernelGGL((walk_lds_kernel<1, MODE>), dim3(grid), dim3(WL_THREADS), 0, stream, a);
   else hipLaunchKernelGGL((walk_lds_kernel<0, MODE>), dim3(grid), dim3(WL_THREADS), 0, stream, a);
}

}   // namespace

hipError_t launch_walk_lds(const WalkArgs &a, int mode, int n_cus, unsigned long long max_items, int stats, hipStream_t stream)
{
   const char *e = getenv("MI355RT_WL_BLOCKS_PER_CU");
   const int per_cu = e && *e ? atoi(e) : 3;   // LDS: 50 KB per workgroup
   static_assert(WL_WAVES_PER_SIMD >= 1, "");
   const unsigned long long want = (unsigned long long) n_cus * (unsigned long long) per_cu;
   const unsigned long long need = (max_items + WL_SLOTS - 1) / WL_SLOTS;
   const unsigned grid = (unsigned) (need < want ? (need ? need : 1ull) : want);
   switch (mode) {
   case MODE_CLOSEST: launch_mode<MODE_CLOSEST>(a, grid, stats, stream); break;
   case MODE_SHADOW: launch_mode<MODE_SHADOW>(a, grid, stats, stream); break;
   case MODE_SHADOW_ANY: launch_mode<MODE_SHADOW_ANY>(a, grid, stats, stream); break;
   default: launch_mode<MODE_FUSED>(a, grid, stats, stream); break;
   }
   return hipGetLastError();
}

}   // namespace mi355rt
