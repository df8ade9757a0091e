// trace_kernels.hip -- gfx950 kernels for the per-pixel trace loop.
//
// Compiled with -ffp-contract=off: the reference is Java, which never fuses a*b+c, and hit ids are
// compared bit-exactly.  f32 divide and sqrt are the correctly rounded forms (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt); Vector3f.normalize's (float)(1.0/Math.sqrt((double)s))
// is done in f64 as written.
//
// Reference semantics carried over (paths under /root/reference/src/stephen/ranger/ar/):
//   slab test        RTStatics.aabbIntersection            RTStatics.java:102-158
//   triangle test    Triangle.intersectsTriangle           sceneObjects/Triangle.java:85-133
//   leaf / node fold KDNode.getChildIntersection           bounds/KDNode.java:92-123
//   object fold      Camera.getClosestIntersection         Camera.java:193-215
//   shadow test      RTStatics.shadowIntersects            RTStatics.java:617-639
//   shading          PhongLightingModel / BRDFMaterial / ReflectionMaterial
//   pixel loop       RenderThread.run, Camera.setPixel     RenderThread.java:30-69, Camera.java:96-108
//
// What is NOT carried over is the visit order.  The reference visits every child whose box passes
// the slab test, in left/right order, with no early-out.  Here children are visited nearest-first
// and a subtree is skipped when the distance from the ray ORIGIN to its box exceeds the best hit
// distance so far (plus a margin).  That is exact because the reference's hit distance is the
// Euclidean distance from the origin to an interpolated point inside the triangle's bounding box
// (Triangle.java:126-130), and every box on the path to a leaf contains that box.  The reference's
// order-dependent tie rule (equal distance: the later leaf in left-to-right order wins, within a
// leaf the earlier face wins) is restated order-independently on (distance, first-slot-of-leaf).
#include <hip/hip_runtime.h>

#include "kernels.h"
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
   float abs_margin;
};

struct Consts {
   float eps, nearp, farp;
};

struct Counters {
   unsigned box, tri;
};

struct ObjHit {
   int face;       // < 0: none
   unsigned key;   // first slot of the leaf that produced it
   float w, px, py, pz, u, v;
};

// RTStatics.aabbIntersection with 1/direction hoisted out (same value on every call for one ray).
// Written branch-free: after a "return false" nothing else is observable, so evaluating the rest is
// equivalent.  Every comparison is the reference's own (NaNs fall through exactly as they do there).
__device__ __forceinline__ bool slab_test(const Ray &r, const Consts &k, float lox, float loy, float loz, float hix, float hiy,
                                          float hiz, float &t_enter)
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
   return ok && (txmin < k.farp) && (txmax > k.nearp);
}

// Squared distance from the ray origin to a box; NaN terms are ignored (=> smaller => less pruning).
__device__ __forceinline__ float origin_box_dist2(const Ray &r, float lox, float loy, float loz, float hix, float hiy, float hiz)
{
   const float dx = fmaxf(fmaxf(lox - r.o.x, r.o.x - hix), 0.0f);
   const float dy = fmaxf(fmaxf(loy - r.o.y, r.o.y - hiy), 0.0f);
   const float dz = fmaxf(fmaxf(loz - r.o.z, r.o.z - hiz), 0.0f);
   return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ float prune_limit2(const Ray &r, float w)
{
   const float lim = w * 1.0001f + r.abs_margin;
   return lim * lim;
}

// All faces of one leaf (KDNode.java:110-118) folded into the object's best hit.
template <bool STATS>
__device__ __forceinline__ void leaf_faces(const float4 *__restrict__ tris, int link, int cnt, const Ray &r, const Consts &k,
                                           float shadow_distance, ObjHit &best, float &lim2, Counters &ctr)
{
   for (int s = 0; s < cnt; ++s) {
      const float4 *t = tris + 3 * (size_t) (link + s);
      const float4 A = t[0], B = t[1], C = t[2];
      if (STATS) ctr.tri++;
      const V3 p0 = mk3(A.x, A.y, A.z);
      const V3 e1 = sub3(mk3(B.x, B.y, B.z), p0);
      const V3 e2 = sub3(mk3(C.x, C.y, C.z), p0);
      const V3 p = cross3(r.d, e2);
      const float divisor = dot3(p, e1);
      if ((divisor < k.eps) && (divisor > -k.eps)) continue;
      const V3 T = sub3(r.o, p0);
      const V3 q = cross3(T, e1);
      const float u = dot3(p, T) / divisor;
      if ((u < 0) || (u > 1)) continue;
      const float v = dot3(q, r.d) / divisor;
      if ((v < 0) || (v + u > 1)) continue;
      const float wb = 1.0f - u - v;
      const float hx = wb * A.x + u * B.x + v * C.x;
      const float hy = wb * A.y + u * B.y + v * C.y;
      const float hz = wb * A.z + u * B.z + v * C.z;
      const float dist = distance3(r.o, mk3(hx, hy, hz));
      if (!(dist > shadow_distance)) continue;
      if (best.face < 0 || dist < best.w || (dist == best.w && (unsigned) link > best.key)) {
         best.face = __float_as_int(A.w);
         best.key = (unsigned) link;
         best.w = dist;
         best.px = hx;
         best.py = hy;
         best.pz = hz;
         best.u = u;
         best.v = v;
         const float l2 = prune_limit2(r, dist);
         lim2 = fminf(lim2, l2);
      }
   }
}

// Closest accepted hit of one object = KDTree.getChildIntersection (bounds/KDTree.java:59-65).
// `lim2_in` is an external squared prune limit (infinity when there is none).
template <bool STATS>
__device__ __forceinline__ void traverse_object(const DevObject &ob, const Ray &r, const Consts &k, bool want, float lim2_in,
                                                ObjHit &best, int *__restrict__ stack, Counters &ctr)
{
   best.face = -1;
   best.key = 0;
   best.w = 0.0f;
   best.px = best.py = best.pz = best.u = best.v = 0.0f;
   float lim2 = lim2_in;
   int sp = 0;
   const float4 *__restrict__ nodes = (const float4 *) ob.nodes;
   const float4 *__restrict__ tris = (const float4 *) ob.tris;

   if (want) {
      float te;
      if (STATS) ctr.box++;
      if (slab_test(r, k, ob.bmin[0], ob.bmin[1], ob.bmin[2], ob.bmax[0], ob.bmax[1], ob.bmax[2], te)) {
         if (ob.root_is_leaf) {
            leaf_faces<STATS>(tris, ob.root_link, ob.root_cnt, r, k, ob.shadow_distance, best, lim2, ctr);
         } else {
            stack[0] = 0;
            sp = 1;
         }
      }
   }

   while (sp > 0) {
      --sp;
      const int node = stack[sp * BLOCK_THREADS];
      const float4 *rec = nodes + 4 * (size_t) node;
      const float4 a = rec[0], b = rec[1], c = rec[2];
      const int4 d = *(const int4 *) (rec + 3);
      float tl = 0.0f, tr = 0.0f;
      bool lok = false, rok = false;
      if (d.x != LINK_NONE) {
         if (STATS) ctr.box++;
         lok = slab_test(r, k, a.x, a.y, a.z, a.w, b.x, b.y, tl) && !(origin_box_dist2(r, a.x, a.y, a.z, a.w, b.x, b.y) > lim2);
      }
      if (d.y != LINK_NONE) {
         if (STATS) ctr.box++;
         rok = slab_test(r, k, b.z, b.w, c.x, c.y, c.z, c.w, tr) && !(origin_box_dist2(r, b.z, b.w, c.x, c.y, c.z, c.w) > lim2);
      }
      // nearest child first; `swap` only chooses an order, never whether a child is visited
      const bool swap = lok && rok && (tr < tl);
      const int n_link = swap ? d.y : d.x, n_cnt = swap ? d.w : d.z;
      const int f_link = swap ? d.x : d.y, f_cnt = swap ? d.z : d.w;
      const bool n_ok = swap ? rok : lok, f_ok = swap ? lok : rok;
      // leaf children are tested on the spot (they are small); internal children go on the stack
      if (n_ok && n_cnt > 0) leaf_faces<STATS>(tris, n_link, n_cnt, r, k, ob.shadow_distance, best, lim2, ctr);
      if (f_ok && f_cnt > 0) leaf_faces<STATS>(tris, f_link, f_cnt, r, k, ob.shadow_distance, best, lim2, ctr);
      if (f_ok && f_cnt == 0) {
         stack[sp * BLOCK_THREADS] = f_link;
         ++sp;
      }
      if (n_ok && n_cnt == 0) {
         stack[sp * BLOCK_THREADS] = n_link;
         ++sp;
      }
   }
}

// interpolated normal of a hit: w*n0 + u*n1 + v*n2 (Triangle.java:128-129), not renormalised
__device__ __forceinline__ V3 hit_normal(const DevObject *objs, int obj, const ObjHit &h)
{
   const float4 *fn = (const float4 *) objs[obj].fnrm + 3 * (size_t) h.face;
   const float4 n0 = fn[0], n1 = fn[1], n2 = fn[2];
   const float wb = 1.0f - h.u - h.v;
   return mk3(wb * n0.x + h.u * n1.x + h.v * n2.x, wb * n0.y + h.u * n1.y + h.v * n2.y, wb * n0.z + h.u * n1.z + h.v * n2.z);
}

__device__ __forceinline__ void make_ray(Ray &r, V3 o, V3 d, float scene_abs_max)
{
   r.o = o;
   r.d = d;
   r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
   r.abs_margin = 1e-5f * (scene_abs_max + fabsf(o.x) + fabsf(o.y) + fabsf(o.z));
}

// ---------------------------------------------------------------- colour / BRDF (device)

// java.lang.Math.min / max (float)
__device__ __forceinline__ float jminf(float a, float b)
{
   if (a != a) return a;
   if (a == 0.0f && b == 0.0f && (__float_as_int(b) < 0)) return b;
   return (a <= b) ? a : b;
}
__device__ __forceinline__ float jmaxf(float a, float b)
{
   if (a != a) return a;
   if (a == 0.0f && b == 0.0f && (__float_as_int(a) < 0)) return b;
   return (a >= b) ? a : b;
}

// RTStatics.convertRGBtoHSV RTStatics.java:442-460
__device__ __forceinline__ void rgb_to_hsv(const float rgb[3], float hsv[3])
{
   const float R = rgb[0], G = rgb[1], B = rgb[2];
   const float x = jminf(R, jminf(G, B));
   const float v = jmaxf(R, jmaxf(G, B));
   if (v == x) {
      hsv[0] = -1.0f;
      hsv[1] = 0.0f;
      hsv[2] = v;
      return;
   }
   const float f = R == x ? G - B : G == x ? B - R : R - G;
   const int i = R == x ? 3 : G == x ? 5 : 1;
   hsv[0] = (float) i - f / (v - x);
   hsv[1] = (v - x) / v;
   hsv[2] = v;
}

// RTStatics.convertHSVtoRGB RTStatics.java:462-512
__device__ __forceinline__ void hsv_to_rgb(const float hsv[3], float rgb[3])
{
   const float h = hsv[0], s = hsv[1], v = hsv[2];
   if (h == -1.0f) {
      rgb[0] = rgb[1] = rgb[2] = v;
      return;
   }
   const double fl = floor((double) h);
   int i;
   if (fl != fl) i = 0;
   else if (fl >= 2147483647.0) i = 2147483647;
   else if (fl <= -2147483648.0) i = (-2147483647 - 1);
   else i = (int) fl;
   float f = h - (float) i;
   if (i % 2 == 0) f = 1.0f - f;
   const float m = v * (1.0f - s);
   const float n = v * (1.0f - s * f);
   float r = 0.0f, g = 0.0f, b = 0.0f;
   switch (i) {
   case 6:
   case 0: r = v; g = n; b = m; break;
   case 1: r = n; g = v; b = m; break;
   case 2: r = m; g = v; b = n; break;
   case 3: r = m; g = n; b = v; break;
   case 4: r = n; g = m; b = v; break;
   case 5: r = v; g = m; b = n; break;
   default: break;
   }
   rgb[0] = jminf(1.0f, jmaxf(0.0f, r));
   rgb[1] = jminf(1.0f, jmaxf(0.0f, g));
   rgb[2] = jminf(1.0f, jmaxf(0.0f, b));
}

#define MI_PI_D 3.14159265358979323846
#define MI_PI_F ((float) MI_PI_D)
#define MI_2PI_F (2.0f * (float) MI_PI_D)

// PBRTMath.normalizePhi PBRTMath.java:99-101
__device__ __forceinline__ float normalize_phi(float phi) { return ((double) phi < MI_PI_D / 2.0f) ? phi + 2.0f * MI_PI_F : phi; }

// PBRTMath.getThetaPhi PBRTMath.java:103-112
__device__ __forceinline__ void theta_phi(V3 n, V3 s, V3 t, V3 dir, float &theta, float &phi)
{
   const float z = dot3(dir, n);
   const float y = dot3(dir, s);
   const float x = dot3(dir, t);
   theta = (float) acos((double) z);
   phi = (float) atan2((double) y, (double) x);
}

// PBRTMath.getRemappedDirection (both overloads) PBRTMath.java:59-97
__device__ __forceinline__ void remap_vectors(V3 normal, V3 tangent, V3 in_dir, V3 out_dir, float res[3])
{
   const V3 n = normalize3(normal);
   const V3 t = normalize3(tangent);
   const V3 s = normalize3(cross3(n, t));
   float ti, pi_, to, po;
   theta_phi(n, s, t, in_dir, ti, pi_);
   theta_phi(n, s, t, out_dir, to, po);
   const float cosi = (float) cos((double) ti);
   const float coso = (float) cos((double) to);
   const float sini = (float) sin((double) ti);
   const float sino = (float) sin((double) to);
   const float phii = normalize_phi(pi_);
   const float phio = normalize_phi(po);
   float dphi = phii - phio;
   if (dphi < 0.0f) dphi += MI_2PI_F;
   if (dphi > MI_2PI_F) dphi -= MI_2PI_F;
   if (dphi > MI_PI_F) dphi = MI_2PI_F - dphi;
   res[0] = sini * sino;
   res[1] = dphi / MI_PI_F;
   res[2] = cosi * coso;
}

// BRDFMaterial.getBRDFLuminocity materials/BRDFMaterial.java:53-92
__device__ __noinline__ float brdf_luminocity(const float *__restrict__ dirs, const float *__restrict__ weights, int n_dirs,
                                              int brdf_samples, V3 ray_dir, V3 point, V3 normal, V3 light)
{
   float luminocity = 0.0f, weight = 0.0f;
   int ctr = 0;
   const V3 neg_ray = normalize3(scale3(ray_dir, -1.0f));
   const V3 temp_dir = normalize3(sub3(light, point));
   // PBRTMath.getNormalTangent PBRTMath.java:118-130
   V3 rv = mk3(0.0f, 1.0f, 0.0f);
   if (fabsf(dot3(normal, rv)) > 0.9f) rv = mk3(-1.0f, 0.0f, 0.0f);
   const V3 tangent = cross3(normal, rv);
   for (int i = 0; i < brdf_samples; i++) {
      float rm[3];
      remap_vectors(normal, tangent, neg_ray, reflect3(normal, temp_dir), rm);
      float last_max_dist2 = 0.001f;
      while ((ctr < 4) && (last_max_dist2 < 1.5f)) {
         for (int j = 0; j < n_dirs; j++) {
            const float d0 = rm[0] - dirs[3 * j], d1 = rm[1] - dirs[3 * j + 1], d2 = rm[2] - dirs[3 * j + 2];
            const float dist2 = fabsf(d0 * d0 + d1 * d1 + d2 * d2);   // PBRTMath.getDistance2
            if (dist2 < last_max_dist2) {
               const float temp = (float) exp(-100.0 * (double) dist2);
               luminocity += weights[j] * temp;
               weight += temp;
               ctr++;
            }
         }
         last_max_dist2 *= 2.0f;
      }
      if (ctr >= 4) break;   // later samples would neither loop nor change anything
   }
   if (ctr == 0) return 0.0f;
   return luminocity / weight * 6.0f;
}

// arithmetic of PhongLightingModel.getPixelColor lighting/PhongLightingModel.java:32-51
__device__ __forceinline__ void phong_combine(const float color_in[3], const float kd[3], const float ks[3], float a,
                                              const float is_[3], const float ia[3], V3 light, V3 point, V3 normal, V3 ray_dir,
                                              bool shadowed, float out[3])
{
   const V3 L = normalize3(sub3(light, point));
   const V3 V = sub3(mk3(0.0f, 0.0f, 0.0f), ray_dir);
   const V3 R = reflect3(normal, L);
   const double LdotN = (double) dot3(L, normal);
   const double RdotVexpA = pow((double) dot3(V, R), (double) a);
   const double spec = shadowed ? 0.0f : 1.0f;
   const double shade = shadowed ? 0.6f : 1.0f;
   for (int c = 0; c < 3; c++) {
      const double term = shade * ((double) kd[c] * LdotN * (double) is_[c] + spec * (double) ks[c] * RdotVexpA * (double) is_[c] +
                                   (double) (0.4f * ia[c]));
      out[c] = (float) ((double) color_in[c] * term);
   }
}

// java.util.Random restated (see DESIGN.md "sample jitter")
__device__ __forceinline__ unsigned long long rng_init(unsigned long long seed, unsigned long long pixel)
{
   unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (pixel + 1ULL);
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
   z = z ^ (z >> 31);
   return (z ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1);
}
__device__ __forceinline__ float rng_next_float(unsigned long long &state)
{
   state = (state * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
   return (float) (int) (state >> 24) / (float) (1 << 24);
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
   for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
   return v;
}

enum { PH_PRIMARY = 0, PH_SECONDARY = 1, PH_SHADOW = 2 };

}   // namespace

// ---------------------------------------------------------------- frame kernel
//
// One lane owns one pixel and walks its samples in order (RTStatics.computeColorAverage sums in
// sample order).  Each trip round the outer loop traces ONE ray for every live lane, whatever kind
// (primary, mirror bounce, shadow), so all ray kinds share one traversal loop and a lane on its
// shadow ray runs beside a lane on its next primary ray.
template <bool STATS>
__global__ __launch_bounds__(BLOCK_THREADS) void render_kernel(const RenderParams p)
{
   __shared__ int s_stack[STACK_SLOTS * BLOCK_THREADS];
   const int tid = threadIdx.x;
   const int lane = tid & 63;
   int *stack = s_stack + tid;

   const long long micro = (long long) blockIdx.x * (BLOCK_THREADS / 64) + (tid >> 6);
   bool active = micro < (long long) p.n_micro;
   int px = 0, py = 0;
   if (active) {
      const int k = (int) (micro / p.micro_per_tile);
      const int within = (int) (micro % p.micro_per_tile);
      const int tile = p.tile_rank + k * p.tile_world;
      const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
      const int mpr = p.tile_size / 8;
      px = tx * p.tile_size + (within % mpr) * 8 + (lane & 7);
      py = ty * p.tile_size + (within / mpr) * 8 + (lane >> 3);
      active = (px < p.width) && (py < p.height);
   }
   const long long pix = (long long) py * p.width + px;

   const Consts k = { p.epsilon, p.near_plane, p.far_plane };
   const V3 cam_o = mk3(p.origin[0], p.origin[1], p.origin[2]);
   const V3 light = mk3(p.light_origin[0], p.light_origin[1], p.light_origin[2]);
   // RenderThread.java:39-40
   const float x0 = (float) px * p.x_inc + p.x_min;
   const float y0 = (float) py * p.y_inc + p.y_min;

   unsigned long long rng = rng_init(p.rng_seed, (unsigned long long) pix);
   Counters ctr = { 0u, 0u };
   unsigned n_primary = 0, n_shadow = 0, n_secondary = 0, n_hits = 0;

   int j = 0, phase = PH_PRIMARY, bounces = 0;
   float ar = 0.0f, ag = 0.0f, ab = 0.0f;
   V3 P0 = mk3(0, 0, 0), N0 = P0, D0 = P0;   // primary hit (for Phong)
   V3 Pc = P0, Nc = P0, Dc = P0;             // hit whose material is being evaluated
   int obj0 = 0, objc = 0;
   float mat[3] = { 0.0f, 0.0f, 0.0f };

   for (;;) {
      if (!__any(active)) break;

      // ---- next ray of this lane
      V3 o = cam_o, d = mk3(0.0f, 0.0f, -1.0f);
      int mirror = -1;
      const bool shadow_mode = (phase == PH_SHADOW);
      if (active) {
         if (phase == PH_PRIMARY) {
            float vx = x0, vy = y0;
            if (j != 0) {   // RenderThread.java:43-44
               vx = x0 + (rng_next_float(rng) * 2.0f - 1.0f) * p.x_inc / 2.0f;
               vy = y0 + (rng_next_float(rng) * 2.0f - 1.0f) * p.y_inc / 2.0f;
            }
            const float vz = -p.near_plane_cam;
            V3 t = mk3(p.rot[0] * vx + p.rot[1] * vy + p.rot[2] * vz, p.rot[3] * vx + p.rot[4] * vy + p.rot[5] * vz,
                       p.rot[6] * vx + p.rot[7] * vy + p.rot[8] * vz);
            d = normalize3(normalize3(t));   // RenderThread.java:47, then Ray ctor (Ray.java:12)
            o = cam_o;
            n_primary++;
         } else if (phase == PH_SECONDARY) {
            // ReflectionMaterial.java:16-17, Camera.java:196-200
            o = add3(Pc, scale3(Nc, k.eps));
            d = normalize3(reflect3(Nc, Dc));
            mirror = objc;
            n_secondary++;
         } else {
            // RTStatics.shadowIntersects RTStatics.java:618-622
            d = normalize3(normalize3(sub3(light, P0)));
            o = add3(P0, scale3(N0, k.eps));
            n_shadow++;
         }
      }
      Ray r;
      make_ray(r, o, d, p.scene_abs_max);

      // ---- objects, in scene order
      ObjHit best;
      best.face = -1;
      best.key = 0;
      best.w = 0.0f;
      best.px = best.py = best.pz = best.u = best.v = 0.0f;
      int best_obj = -1;
      bool occluded = false;
      for (int i = 0; i < p.n_objs; i++) {
         const DevObject &ob = p.objs[i];
         const bool want = active && (i != mirror) && !(shadow_mode && occluded);
         const float lim2_in = (!shadow_mode && best_obj >= 0) ? prune_limit2(r, best.w) : __builtin_inff();
         ObjHit h;
         traverse_object<STATS>(ob, r, k, want, lim2_in, h, stack, ctr);
         if (want && h.face >= 0) {
            if (shadow_mode) {
               if (h.w > k.eps) {   // RTStatics.java:628-633
                  const float light_distance = distance3(mk3(h.px, h.py, h.pz), light);
                  if (h.w < light_distance + k.eps) occluded = true;
               }
            } else if ((h.w > k.eps) && (best_obj < 0 || h.w < best.w)) {   // Camera.java:208
               best = h;
               best_obj = i;
            }
         }
      }

      // ---- consume the result
      if (active) {
         bool to_material = false, to_light = false, to_accum = false;
         float cr = 0.0f, cg = 0.0f, cb = 0.0f;
         if (phase == PH_PRIMARY) {
            if (best_obj < 0) {
               cr = p.light_ambient[0];
               cg = p.light_ambient[1];
               cb = p.light_ambient[2];
               to_accum = true;
               if (j == 0) {
                  if (p.hit_id) p.hit_id[pix] = -1;
                  if (p.hit_obj) p.hit_obj[pix] = -1;
                  if (p.hit_w) p.hit_w[pix] = 0.0f;
               }
            } else {
               n_hits++;
               P0 = mk3(best.px, best.py, best.pz);
               N0 = hit_normal(p.objs, best_obj, best);
               D0 = d;
               obj0 = best_obj;
               Pc = P0;
               Nc = N0;
               Dc = D0;
               objc = obj0;
               bounces = 0;
               to_material = true;
               if (j == 0) {
                  if (p.hit_id) p.hit_id[pix] = best.face;
                  if (p.hit_obj) p.hit_obj[pix] = best_obj;
                  if (p.hit_w) p.hit_w[pix] = best.w;
               }
            }
         } else if (phase == PH_SECONDARY) {
            if (best_obj < 0) {   // ReflectionMaterial.java:19-20
               mat[0] = p.light_ambient[0];
               mat[1] = p.light_ambient[1];
               mat[2] = p.light_ambient[2];
               to_light = true;
            } else {
               Pc = mk3(best.px, best.py, best.pz);
               Nc = hit_normal(p.objs, best_obj, best);
               Dc = d;
               objc = best_obj;
               bounces++;
               to_material = true;
            }
         } else {
            const DevMaterial &m0 = p.objs[obj0].mat;
            float out[3];
            phong_combine(mat, m0.diffuse, m0.specular, m0.shininess, p.light_emission, p.light_ambient, light, P0, N0, D0, occluded,
                          out);
            cr = out[0];
            cg = out[1];
            cb = out[2];
            to_accum = true;
         }

         if (to_material) {
            const DevMaterial &m = p.objs[objc].mat;
            if (m.kind == 2u /* reflection */ && bounces < p.max_mirror_bounces) {
               phase = PH_SECONDARY;
            } else if (m.kind == 1u /* BRDF */) {
               // BRDFMaterial.getMaterialColor materials/BRDFMaterial.java:158-167
               const float lum = brdf_luminocity(p.brdf_dirs, p.brdf_weights + (size_t) m.brdf_id * (size_t) p.n_dirs, p.n_dirs,
                                                 p.brdf_samples, Dc, Pc, Nc, light);
               float hsv[3];
               rgb_to_hsv(m.diffuse, hsv);
               hsv[2] = lum;
               hsv_to_rgb(hsv, mat);
               to_light = true;
            } else {
               mat[0] = m.diffuse[0];
               mat[1] = m.diffuse[1];
               mat[2] = m.diffuse[2];
               to_light = true;
            }
         }
         if (to_light) {
            if (p.lighting_model == 1) {
               phase = PH_SHADOW;
            } else {
               cr = mat[0];
               cg = mat[1];
               cb = mat[2];
               to_accum = true;
            }
         }
         if (to_accum) {
            ar += cr;
            ag += cg;
            ab += cb;
            j++;
            phase = PH_PRIMARY;
            if (j >= p.multi_samples) {
               // RTStatics.computeColorAverage + Camera.setPixel
               float fr = ar / (float) p.multi_samples, fg = ag / (float) p.multi_samples, fb = ab / (float) p.multi_samples;
               if (fr != fr || fg != fg || fb != fb) fr = fg = fb = 0.0f;
               if (p.rgb) {
                  p.rgb[3 * pix + 0] = fr;
                  p.rgb[3 * pix + 1] = fg;
                  p.rgb[3 * pix + 2] = fb;
               }
               if (p.argb8) {
                  const float br = jminf(1.0f, jmaxf(0.0f, fr)), bg = jminf(1.0f, jmaxf(0.0f, fg)), bb = jminf(1.0f, jmaxf(0.0f, fb));
                  const unsigned ir = (unsigned) (int) ((double) (br * 255.0f) + 0.5);
                  const unsigned ig = (unsigned) (int) ((double) (bg * 255.0f) + 0.5);
                  const unsigned ib = (unsigned) (int) ((double) (bb * 255.0f) + 0.5);
                  p.argb8[pix] = 0xFF000000u | ((ir & 255u) << 16) | ((ig & 255u) << 8) | (ib & 255u);
               }
               active = false;
            }
         }
      }
   }

   if (STATS && p.stats) {
      const unsigned long long s0 = wave_sum(n_primary), s1 = wave_sum(n_shadow), s2 = wave_sum(n_secondary);
      const unsigned long long s3 = wave_sum(ctr.box), s4 = wave_sum(ctr.tri), s5 = wave_sum(n_hits);
      if (lane == 0) {
         atomicAdd(&p.stats[0], s0);
         atomicAdd(&p.stats[1], s1);
         atomicAdd(&p.stats[2], s2);
         atomicAdd(&p.stats[3], s3);
         atomicAdd(&p.stats[4], s4);
         atomicAdd(&p.stats[5], s5);
      }
   }
}

// ---------------------------------------------------------------- ray-batch kernel
// Camera.getClosestIntersection(null, origin, direction, null, 0) per ray.
__global__ __launch_bounds__(BLOCK_THREADS) void trace_rays_kernel(const TraceParams p)
{
   __shared__ int s_stack[STACK_SLOTS * BLOCK_THREADS];
   int *stack = s_stack + threadIdx.x;
   const unsigned long long i = (unsigned long long) blockIdx.x * BLOCK_THREADS + threadIdx.x;
   const bool active = i < p.n;
   const Consts k = { p.epsilon, p.near_plane, p.far_plane };
   V3 o = mk3(0, 0, 0), d = mk3(0, 0, -1);
   if (active) {
      o = mk3(p.origins[3 * i], p.origins[3 * i + 1], p.origins[3 * i + 2]);
      d = normalize3(mk3(p.dirs[3 * i], p.dirs[3 * i + 1], p.dirs[3 * i + 2]));   // Ray ctor
   }
   Ray r;
   make_ray(r, o, d, p.scene_abs_max);
   Counters ctr = { 0u, 0u };
   ObjHit best;
   best.face = -1;
   best.key = 0;
   best.w = 0.0f;
   best.px = best.py = best.pz = best.u = best.v = 0.0f;
   int best_obj = -1;
   for (int j = 0; j < p.n_objs; j++) {
      const float lim2_in = (best_obj >= 0) ? prune_limit2(r, best.w) : __builtin_inff();
      ObjHit h;
      traverse_object<false>(p.objs[j], r, k, active, lim2_in, h, stack, ctr);
      if (active && h.face >= 0 && (h.w > k.eps) && (best_obj < 0 || h.w < best.w)) {
         best = h;
         best_obj = j;
      }
   }
   if (!active) return;
   V3 n = mk3(0, 0, 0);
   if (best_obj >= 0 && p.hit_normal) n = hit_normal(p.objs, best_obj, best);
   if (p.hit_id) p.hit_id[i] = best_obj >= 0 ? best.face : -1;
   if (p.hit_obj) p.hit_obj[i] = best_obj;
   if (p.hit_w) p.hit_w[i] = best_obj >= 0 ? best.w : 0.0f;
   if (p.hit_point) {
      p.hit_point[3 * i] = best.px;
      p.hit_point[3 * i + 1] = best.py;
      p.hit_point[3 * i + 2] = best.pz;
   }
   if (p.hit_normal) {
      p.hit_normal[3 * i] = n.x;
      p.hit_normal[3 * i + 1] = n.y;
      p.hit_normal[3 * i + 2] = n.z;
   }
}

// device arithmetic probe used by the GPU tests: the three primitives parity rests on
__global__ void arith_probe_kernel(unsigned long long n, const float *a, const float *b, float *out_div, float *out_sqrt,
                                   float *out_rnorm, float *out_pow)
{
   const unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n) return;
   out_div[i] = a[i] / b[i];
   out_sqrt[i] = __builtin_sqrtf(a[i]);
   out_rnorm[i] = (float) (1.0 / sqrt((double) a[i]));
   out_pow[i] = (float) pow((double) b[i], 20.0);
}

// ---------------------------------------------------------------- launchers

hipError_t launch_render(const RenderParams &p, bool stats, hipStream_t stream)
{
   if (p.n_micro <= 0) return hipSuccess;
   const unsigned waves_per_block = BLOCK_THREADS / 64;
   const unsigned grid = (unsigned) ((p.n_micro + waves_per_block - 1) / waves_per_block);
   if (stats) hipLaunchKernelGGL(render_kernel<true>, dim3(grid), dim3(BLOCK_THREADS), 0, stream, p);
   else hipLaunchKernelGGL(render_kernel<false>, dim3(grid), dim3(BLOCK_THREADS), 0, stream, p);
   return hipGetLastError();
}

hipError_t launch_trace_rays(const TraceParams &p, hipStream_t stream)
{
   if (p.n == 0) return hipSuccess;
   const unsigned long long grid = (p.n + BLOCK_THREADS - 1) / BLOCK_THREADS;
   if (grid > 0x7fffffffULL) return hipErrorInvalidValue;
   hipLaunchKernelGGL(trace_rays_kernel, dim3((unsigned) grid), dim3(BLOCK_THREADS), 0, stream, p);
   return hipGetLastError();
}

hipError_t launch_arith_probe(unsigned long long n, const float *a, const float *b, float *out_div, float *out_sqrt,
                              float *out_rnorm, float *out_pow, hipStream_t stream)
{
   if (n == 0) return hipSuccess;
   const unsigned grid = (unsigned) ((n + 255) / 256);
   hipLaunchKernelGGL(arith_probe_kernel, dim3(grid), dim3(256), 0, stream, n, a, b, out_div, out_sqrt, out_rnorm, out_pow);
   return hipGetLastError();
}

}   // namespace mi355rt
