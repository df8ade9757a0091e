// device_util.h -- device-side vector arithmetic (the reference's javax.vecmath / RTStatics forms), the slab and
// triangle tests, statistics and queue-append helpers shared by the kernels of libmi355rt.  Internal linkage: every
// .hip file that includes it gets its own inlined copies.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"

namespace mi355rt {

namespace {

struct V3 {
   float x, y, z;
};

__device__ __forceinline__ V3 mk3(float x, float y, float z)
{
   V3 r;
   r.x = x;
   r.y = y;
   r.z = z;
   return r;
}
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 add3(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 scale3(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
// Vector3f.dot: (x*x' + y*y') + z*z'
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// Vector3f.cross(a, b)
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return mk3(a.y * b.z - a.z * b.y, b.x * a.z - b.z * a.x, a.x * b.y - a.y * b.x); }
// Vector3f.normalize
__device__ __forceinline__ V3 normalize3(V3 a)
{
   const float s = a.x * a.x + a.y * a.y + a.z * a.z;
   const float n = (float) (1.0 / sqrt((double) s));
   return mk3(a.x * n, a.y * n, a.z * n);
}
// RTStatics.getDistance: (float) Math.sqrt((double) f32 sum of squares); for an f32 argument the
// correctly rounded f32 sqrt is the same value.
__device__ __forceinline__ float distance3(V3 a, V3 b)
{
   const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
   return __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
}
// RTStatics.getReflectionDirection(n, dir)
__device__ __forceinline__ V3 reflect3(V3 n, V3 dir)
{
   V3 N = scale3(n, 2.0f);
   N = scale3(N, dot3(dir, n));
   return normalize3(sub3(dir, N));
}

struct Ray {
   V3 o, d, inv;
};

__device__ __forceinline__ void make_ray(Ray &r, V3 o, V3 d)
{
   r.o = o;
   r.d = d;
   r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
}

typedef float f2 __attribute__((ext_vector_type(2)));   // operands of the packed-f32 instructions

// every operand the slab test will see from this ray is finite (so none of its products can be a NaN)
__device__ __forceinline__ bool plain_ray(const Ray &r)
{
   const float s = (fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z)) + (fabsf(r.inv.x) + fabsf(r.inv.y) + fabsf(r.inv.z));
   return s < __builtin_inff();   // false for any inf or NaN among them (a finite sum that overflows only costs the fast path)
}

// RTStatics.aabbIntersection (RTStatics.java:102-158) with 1/direction hoisted out (same value on
// every call for one ray).  Branch-free: after a "return false" nothing else is observable, so
// evaluating the rest is equivalent.  Every comparison is the reference's own, so NaNs fall
// through exactly as they do there.
__device__ __forceinline__ bool slab_test(const Ray &r, float nearp, float farp, float lox, float loy, float loz, float hix,
                                          float hiy, float hiz, float &t_enter)
{
   float txmin, txmax, tymin, tymax, tzmin, tzmax;
   if (r.inv.x >= 0) {
      txmin = (lox - r.o.x) * r.inv.x;
      txmax = (hix - r.o.x) * r.inv.x;
   } else {
      txmin = (hix - r.o.x) * r.inv.x;
      txmax = (lox - r.o.x) * r.inv.x;
   }
   if (r.inv.y >= 0) {
      tymin = (loy - r.o.y) * r.inv.y;
      tymax = (hiy - r.o.y) * r.inv.y;
   } else {
      tymin = (hiy - r.o.y) * r.inv.y;
      tymax = (loy - r.o.y) * r.inv.y;
   }
   bool ok = !(txmin > tymax || tymin > txmax);
   if (tymin > txmin) txmin = tymin;
   if (tymax < txmax) txmax = tymax;
   if (r.inv.z >= 0) {
      tzmin = (loz - r.o.z) * r.inv.z;
      tzmax = (hiz - r.o.z) * r.inv.z;
   } else {
      tzmin = (hiz - r.o.z) * r.inv.z;
      tzmax = (loz - r.o.z) * r.inv.z;
   }
   ok = ok && !(txmin > tzmax || tzmin > txmax);
   if (tzmin > txmin) txmin = tzmin;
   if (tzmax < txmax) txmax = tzmax;
   t_enter = txmin;
   return ok && (txmin < farp) && (txmax > nearp);
}

// Squared distance from the ray origin to a box; NaN terms are ignored (=> smaller => less pruning).
__device__ __forceinline__ float origin_box_dist2(const Ray &r, float lox, float loy, float loz, float hix, float hiy, float hiz)
{
   const float dx = fmaxf(fmaxf(lox - r.o.x, r.o.x - hix), 0.0f);
   const float dy = fmaxf(fmaxf(loy - r.o.y, r.o.y - hiy), 0.0f);
   const float dz = fmaxf(fmaxf(loz - r.o.z, r.o.z - hiz), 0.0f);
   return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ float prune_limit2(const Ray &r, float w, float scene_abs_max)
{
   const float lim = w * 1.0001f + 1e-5f * (scene_abs_max + fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z));
   return lim * lim;
}

// MODE_SHADOW, the reference-exact shadow query.  RTStatics.shadowIntersects takes an object's CLOSEST accepted hit c
// (distance w_c = |o - P_c| from the ray origin to the interpolated point P_c, which lies in its face, hence in every
// box around that face -- but not necessarily on the ray: for a face almost parallel to the ray the barycentrics are
// rounding noise) and calls the object an occluder iff w_c > EPSILON && w_c < |P_c - light| + EPSILON
// (RTStatics.java:626-633).  Two exact shortcuts follow from the triangle inequality alone, with D = |o - light|:
//   (1) stop early: |P_c - light| >= D - w_c, so any closest hit with w_c < (D + EPSILON) / 2 passes.  Hence once ANY
//       accepted face a of the object has w_a safely below D / 2, the closest one has too (w_c <= w_a): the object
//       occludes, whatever else the walk would find (needs shadowDistance >= EPSILON, so that w_c > EPSILON as well).
//   (2) skip hopeless boxes: every face in a box has w >= dist(o, box) and |P - light| <= the distance from the light to
//       the box's farthest corner; when dist(o, box) exceeds that (plus EPSILON and margins) no face in it can pass.
//       Skipping them can only remove failing candidates, so: no candidate left -> the closest (if any) failed -> no
//       occluder; the closest remaining candidate fails -> the true closest is it or a skipped one -> no occluder.
//       Only "the closest remaining candidate PASSES" without (1) having fired is undecided (a skipped face might have
//       been closer): that ray walks the object again with (1) and (2) off.  With hit points on the ray this needs a
//       hit inside the margin band around D / 2; it is there for the noise cases above.
// shadow_safe_w: the distance below which (1) applies, <= 0 when it never does.  m bounds the rounding of every
// distance involved (coordinates up to the scene's size), as in prune_limit2.
__device__ __forceinline__ float shadow_safe_w(const Ray &r, V3 light, float scene_abs_max)
{
   const float m = 1e-5f * (scene_abs_max + fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z) + fabsf(light.x) + fabsf(light.y) + fabsf(light.z));
   const float t = distance3(r.o, light) * 0.5f * 0.999f - 2.0f * m;
   return t == t ? t : 0.0f;
}
// squared distance from the light to the farthest corner of a box
__device__ __forceinline__ float light_far2(V3 light, float lox, float loy, float loz, float hix, float hiy, float hiz)
{
   const float fx = fmaxf(light.x - lox, hix - light.x), fy = fmaxf(light.y - loy, hiy - light.y), fz = fmaxf(light.z - loz, hiz - light.z);
   return fx * fx + fy * fy + fz * fz;
}

// Triangle.intersectsTriangle (sceneObjects/Triangle.java:85-133) on one leaf slot.
// Returns false where the reference returns null; otherwise the barycentrics, the interpolated
// point and its distance from the origin (returnValue[6]).
__device__ __forceinline__ bool tri_test(const float4 A, const float4 B, const float4 C, const Ray &r, float eps, float &u, float &v,
                                         V3 &P, float &dist)
{
   const V3 p0 = mk3(A.x, A.y, A.z);
   const V3 e1 = sub3(mk3(B.x, B.y, B.z), p0);
   const V3 e2 = sub3(mk3(C.x, C.y, C.z), p0);
   const V3 p = cross3(r.d, e2);
   const float divisor = dot3(p, e1);
   if ((divisor < eps) && (divisor > -eps)) return false;
   const V3 T = sub3(r.o, p0);
   const V3 q = cross3(T, e1);
   u = dot3(p, T) / divisor;
   if ((u < 0) || (u > 1)) return false;
   v = dot3(q, r.d) / divisor;
   if ((v < 0) || (v + u > 1)) return false;
   const float wb = 1.0f - u - v;
   P = mk3(wb * A.x + u * B.x + v * C.x, wb * A.y + u * B.y + v * C.y, wb * A.z + u * B.z + v * C.z);
   dist = distance3(r.o, P);
   return true;
}

// The tests of Triangle.intersectsTriangle alone (Triangle.java:85-124): does the ray meet this face at all?  The
// closest-hit walker asks only this in its leaf loop and leaves the point and distance of the few faces that pass to
// a later, batched tri_test (same arithmetic, so the same answer).
__device__ __forceinline__ bool tri_meets(const float4 A, const float4 B, const float4 C, const Ray &r, float eps)
{
   const V3 p0 = mk3(A.x, A.y, A.z);
   const V3 e1 = sub3(mk3(B.x, B.y, B.z), p0);
   const V3 e2 = sub3(mk3(C.x, C.y, C.z), p0);
   const V3 p = cross3(r.d, e2);
   const float divisor = dot3(p, e1);
   if ((divisor < eps) && (divisor > -eps)) return false;
   const V3 T = sub3(r.o, p0);
   const V3 q = cross3(T, e1);
   const float u = dot3(p, T) / divisor;
   if ((u < 0) || (u > 1)) return false;
   const float v = dot3(q, r.d) / divisor;
   if ((v < 0) || (v + u > 1)) return false;
   return true;
}

// tri_meets that also hands back the barycentrics it worked out, and the rest of tri_test from them: a face that passed
// in the leaf step keeps (u, v) and the batched pass computes only the point and its distance (same expressions on the
// same bits as tri_test, so the same answer).
__device__ __forceinline__ bool tri_meets_uv(const float4 A, const float4 B, const float4 C, const Ray &r, float eps, float &u, float &v)
{
   const V3 p0 = mk3(A.x, A.y, A.z);
   const V3 e1 = sub3(mk3(B.x, B.y, B.z), p0);
   const V3 e2 = sub3(mk3(C.x, C.y, C.z), p0);
   const V3 p = cross3(r.d, e2);
   const float divisor = dot3(p, e1);
   if ((divisor < eps) && (divisor > -eps)) return false;
   const V3 T = sub3(r.o, p0);
   const V3 q = cross3(T, e1);
   u = dot3(p, T) / divisor;
   if ((u < 0) || (u > 1)) return false;
   v = dot3(q, r.d) / divisor;
   if ((v < 0) || (v + u > 1)) return false;
   return true;
}
__device__ __forceinline__ void tri_point(const float4 A, const float4 B, const float4 C, const Ray &r, float u, float v, V3 &P, float &dist)
{
   const float wb = 1.0f - u - v;
   P = mk3(wb * A.x + u * B.x + v * C.x, wb * A.y + u * B.y + v * C.y, wb * A.z + u * B.z + v * C.z);
   dist = distance3(r.o, P);
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
   for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
   return v;
}

__device__ __forceinline__ void add_stat(uint32_t *counters, int which, unsigned v)
{
   const unsigned long long s = wave_sum((unsigned long long) v);
   if ((threadIdx.x & 63) == 0 && s) atomicAdd((unsigned long long *) (counters + CNT_STATS) + which, s);
}
enum { ST_PRIMARY = 0, ST_SHADOW = 1, ST_SECONDARY = 2, ST_BOX = 3, ST_TRI = 4, ST_HITS = 5,
       // walker diagnostics (collect_stats builds only): lane-trips and wave-trips of each phase
       ST_NODE_LANE = 6, ST_NODE_WAVE = 7, ST_LEAF_LANE = 8, ST_LEAF_WAVE = 9, ST_OUTER_WAVE = 10, ST_REFILL = 11,
       // per walker launch kind (0 primary, 1 secondary, 2 shadow): rays, box tests, triangle tests
       ST_KIND_RAYS = 16, ST_KIND_BOX = 19, ST_KIND_TRI = 22,
       // per kind, 6 slots each: ~min t_begin, ~min t_queue_dry, max t_queue_dry, max t_end (100 MHz ticks), most trips of one ray, waves that got a ray
       ST_KIND_TIMELINE = 25,
       // rays of the LAST walker launch by trips taken: < 16, 32, 64, 128, 256, 512, 1024, more (8 slots per kind: primary 43.., shadow 51..)
       ST_TRIP_HIST = 43,
       ST_TAIL_TRIPS = 59 };   // {max, sum} of wave trips after the queue ran dry: primary 59,60; shadow 61,62

// number of set bits of mask below this lane
__device__ __forceinline__ int rank_below(unsigned long long mask)
{
   return (int) __builtin_amdgcn_mbcnt_hi((unsigned) (mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) mask, 0u));
}

// true on exactly one of the currently active lanes
__device__ __forceinline__ bool first_active_lane()
{
   const unsigned long long m = __ballot(1);
   return (int) (threadIdx.x & 63) == (__ffsll((unsigned long long) m) - 1);
}

// appends one record per lane with `push` set; returns the lane's index (wave-aggregated atomic)
__device__ __forceinline__ unsigned queue_push(unsigned *count, bool push)
{
   const unsigned long long mask = __ballot(push);
   if (mask == 0ull) return 0u;
   const int lane = threadIdx.x & 63;
   const int leader = __ffsll((unsigned long long) mask) - 1;
   unsigned base = 0;
   if (lane == leader) base = atomicAdd(count, (unsigned) __popcll(mask));
   base = (unsigned) __shfl((int) base, leader, 64);
   return base + (unsigned) __popcll(mask & ((1ull << lane) - 1ull));
}

// The same for a whole workgroup whose waves all append to ONE queue: one atomic per workgroup.  (An atomic
// that returns a value costs a round trip to the memory side, and the queue counters are only eight words: with
// one atomic per wave generate_kernel spent 0.6 of its 0.73 ms waiting for them.)  Every thread of the workgroup
// must call it; s_cnt holds WAVES + 1 words.
template <int WAVES> __device__ __forceinline__ unsigned block_queue_push(unsigned *count, bool push, unsigned *s_cnt)
{
   const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
   const unsigned long long mask = __ballot(push);
   if (lane == 0) s_cnt[w] = (unsigned) __popcll(mask);
   __syncthreads();
   unsigned before = 0, total = 0;
#pragma unroll
   for (int i = 0; i < WAVES; i++) {
      const unsigned c = s_cnt[i];
      before += i < w ? c : 0u;
      total += c;
   }
   if (threadIdx.x == 0) s_cnt[WAVES] = total ? atomicAdd(count, total) : 0u;
   __syncthreads();
   const unsigned base = s_cnt[WAVES];
   __syncthreads();   // s_cnt may be reused by the next call
   return base + before + (unsigned) __popcll(mask & ((1ull << lane) - 1ull));
}

}   // namespace

}   // namespace mi355rt
