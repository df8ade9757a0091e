// trace_kernels.hip -- gfx950 kernels for the per-pixel trace loop, as a wavefront pipeline:
//
//   generate   one lane per (pixel, sample): primary ray (RenderThread.java:39-47); rays that miss
//              every object's root box are resolved on the spot, the rest are compacted into Q0
//   trace      persistent wave64s pull 64-ray batches from a queue and find the closest hit
//              (Camera.getClosestIntersection over KDTree/KDNode.getChildIntersection)
//   shade      material colour of each hit (ColorInformation / BRDFMaterial / ReflectionMaterial);
//              mirrors push a bounce ray into the secondary queue, lit hits push a shadow ray
//   shadow     same traversal with RTStatics.shadowIntersects' rule, one flag per path
//   phong      PhongLightingModel.getPixelColor's combine -> the sample's colour
//   resolve    RTStatics.computeColorAverage in sample order + Camera.setPixel
//
// Compiled with -ffp-contract=off: the reference is Java, which never fuses a*b+c, and hit ids are
// compared bit-exactly.  f32 divide and sqrt are the correctly rounded forms (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt); Vector3f.normalize's (float)(1.0/Math.sqrt((double)s))
// is done in f64 as written.
//
// What is NOT carried over from the reference is the visit order.  The reference visits every
// child whose box passes the slab test, left then right, with no early-out.  Here children are
// visited nearest-first and a subtree is skipped when the distance from the ray ORIGIN to its box
// exceeds the best hit distance so far (plus a margin).  That is exact because the reference's hit
// distance is the Euclidean distance from the origin to an interpolated point inside the
// triangle's bounding box (Triangle.java:126-130), and every box on the path to a leaf contains
// that box.  The reference's order-dependent tie rule (equal distance: the later leaf in
// left-to-right order wins, within a leaf the earlier face wins; KDNode.java:105,114) is restated
// order-independently on (distance, first-slot-of-leaf).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>

#include "analytic.h"
#include "device_util.h"
#include "kernels.h"
#include "layout.h"
#include "switches.hpp"
#include "walk_args.h"

namespace mi355rt {

namespace {

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

// One pass over the table of remapped directions: visit(j, PBRTMath.getDistance2 of entry j) in table order.  Two
// entries per step in packed f32 (the same operations per entry, taken in the same order).
template <typename F> __device__ __forceinline__ void brdf_scan(const float *__restrict__ dirs, int n_dirs, const float rm[3], F &&visit)
{
   int j = 0;
   for (; j + 1 < n_dirs; j += 2) {
      const f2 d0 = rm[0] - f2 { dirs[3 * j], dirs[3 * j + 3] }, d1 = rm[1] - f2 { dirs[3 * j + 1], dirs[3 * j + 4] };
      const f2 d2 = rm[2] - f2 { dirs[3 * j + 2], dirs[3 * j + 5] };
      const f2 sum = d0 * d0 + d1 * d1 + d2 * d2;
      visit(j, fabsf(sum.x));
      visit(j + 1, fabsf(sum.y));
   }
   for (; j < n_dirs; j++) {
      const float d0 = rm[0] - dirs[3 * j], d1 = rm[1] - dirs[3 * j + 1], d2 = rm[2] - dirs[3 * j + 2];
      visit(j, fabsf(d0 * d0 + d1 * d1 + d2 * d2));
   }
}

// BRDFMaterial.getBRDFLuminocity materials/BRDFMaterial.java:53-92.
// The reference rescans the whole table with a doubled radius (0.001, 0.002, ... below 1.5: eleven scans) until four
// entries -- counted again in every later pass -- have been inside it, adding weight * exp(-100 d) per entry and pass in
// (pass, table) order.  On the measured tables most of those scans find nothing, and the ones that do find a handful.
// The same sum in the same order from three scans: (1) the nearest entry's distance: passes whose radius does not reach
// it find nothing and change nothing; R = the first radius that does; (2) how many entries lie inside R, 2R, 4R, 8R: from R
// on every pass counts at least the nearest entry again, so the loop ends within four passes, and the counts say after
// which one (P); (3) the entries inside the last radius, in table order, into a short per-lane list -- the passes are then
// replayed over that list.  A lane whose list would not fit runs the reference's scans for its passes instead.
constexpr int BRDF_LIST = 16;
__device__ float brdf_luminocity(const float *__restrict__ dirs, const float *__restrict__ weights, int n_dirs, int brdf_samples,
                                 V3 ray_dir, V3 point, V3 normal, V3 light)
{
   __shared__ int s_bj[BRDF_LIST * BLOCK_THREADS];
   __shared__ float s_bd[BRDF_LIST * BLOCK_THREADS];
   int *bj = s_bj + threadIdx.x;
   float *bd = s_bd + threadIdx.x;
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
      float nearest = __builtin_inff();   // (a NaN distance is inside no radius and is never the nearest)
      brdf_scan(dirs, n_dirs, rm, [&](int, float d) {
         if (d < nearest) nearest = d;
      });
      float R = 0.001f;   // the reference's radii, by its own doubling
      while (!(nearest < R) && (R < 1.5f)) R *= 2.0f;
      if (!(R < 1.5f)) continue;   // no pass of this sample finds an entry
      const float R1 = R * 2.0f, R2 = R1 * 2.0f, R3 = R2 * 2.0f;
      int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
      brdf_scan(dirs, n_dirs, rm, [&](int, float d) {
         c0 += d < R ? 1 : 0;
         c1 += d < R1 ? 1 : 0;
         c2 += d < R2 ? 1 : 0;
         c3 += d < R3 ? 1 : 0;
      });
      // the passes this sample runs: radius R, 2R, ... while ctr < 4 and the radius is below 1.5
      int n_pass = 0, after = ctr, inside = 0;
      {
         float rad = R;
         const int c[4] = { c0, c1, c2, c3 };
#pragma unroll
         for (int k = 0; k < 4; k++) {
            if ((after < 4) && (rad < 1.5f)) {
               after += c[k];
               inside = c[k];
               n_pass = k + 1;
               rad *= 2.0f;
            }
         }
      }
      const float r_last = n_pass == 1 ? R : (n_pass == 2 ? R1 : (n_pass == 3 ? R2 : R3));
      if (inside <= BRDF_LIST) {
         int n = 0;
         brdf_scan(dirs, n_dirs, rm, [&](int j, float d) {
            if (d < r_last) {
               bj[n * BLOCK_THREADS] = j;
               bd[n * BLOCK_THREADS] = d;
               n++;
            }
         });
         float rad = R;
         for (int k = 0; k < n_pass; k++) {
            for (int e = 0; e < n; e++) {
               const float d = bd[e * BLOCK_THREADS];
               if (d < rad) {
                  const float temp = (float) exp(-100.0 * (double) d);
                  luminocity += weights[bj[e * BLOCK_THREADS]] * temp;
                  weight += temp;
                  ctr++;
               }
            }
            rad *= 2.0f;
         }
      } else {
         float rad = R;
         for (int k = 0; k < n_pass; k++) {
            brdf_scan(dirs, n_dirs, rm, [&](int j, float d) {
               if (d < rad) {
                  const float temp = (float) exp(-100.0 * (double) d);
                  luminocity += weights[j] * temp;
                  weight += temp;
                  ctr++;
               }
            });
            rad *= 2.0f;
         }
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

// java.util.Random restated (DESIGN.md "sample jitter"): one stream per pixel
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
// state after n further draws: the LCG's affine map composed n times (square-and-multiply)
__device__ __forceinline__ unsigned long long rng_skip(unsigned long long state, unsigned n)
{
   unsigned long long acc_mul = 1ULL, acc_add = 0ULL, cur_mul = 0x5DEECE66DULL, cur_add = 0xBULL;
   while (n) {
      if (n & 1u) {
         acc_mul *= cur_mul;
         acc_add = acc_add * cur_mul + cur_add;
      }
      cur_add = (cur_mul + 1ULL) * cur_add;
      cur_mul *= cur_mul;
      n >>= 1;
   }
   return (acc_mul * state + acc_add) & ((1ULL << 48) - 1);
}

__device__ __forceinline__ void store_ray(float *rays, unsigned idx, V3 o, V3 d, unsigned slot, int aux)
{
   float4 *q = (float4 *) rays + 2 * (size_t) idx;
   q[0] = make_float4(o.x, o.y, o.z, __uint_as_float(slot));
   q[1] = make_float4(d.x, d.y, d.z, __int_as_float(aux));
}

// pixel of (micro-tile, lane); false when it falls off the image.  opix = where the pixel's outputs go: its place in the
// image (y * width + x), or, with settings.tile_compact, in this rank's run of owned tiles (owned tile k, row-major
// inside the tile): what the one exchange step of a multi-GPU frame sends as it is.
// n / d and n % d for the frame's fixed divisors (layout.h FastDiv); 0 <= n < 2^31
__device__ __forceinline__ unsigned fast_div(unsigned n, const FrameArgs::FastDiv &f) { return f.m ? __umulhi(n, f.m) >> f.s : n >> f.s; }
__device__ __forceinline__ void fast_divmod(unsigned n, const FrameArgs::FastDiv &f, int &q, int &r)
{
   const unsigned qq = fast_div(n, f);
   q = (int) qq;
   r = (int) (n - qq * f.d);
}

__device__ __forceinline__ bool micro_pixel(const FrameArgs &p, int micro, int lane, int &px, int &py, long long &opix)
{
   int k, within, tx, ty, wx, wy;
   fast_divmod((unsigned) micro, p.div_micro_per_tile, k, within);
   const int tile = p.tile_rank + k * p.tile_world;
   fast_divmod((unsigned) tile, p.div_tiles_x, ty, tx);
   fast_divmod((unsigned) within, p.div_mpr, wy, wx);
   const int lx = wx * 8 + (lane & 7), ly = wy * 8 + (lane >> 3);
   px = tx * p.tile_size + lx;
   py = ty * p.tile_size + ly;
   opix = p.tile_compact ? ((long long) k * p.tile_size + ly) * p.tile_size + lx : (long long) py * p.width + px;
   return (px < p.width) && (py < p.height);
}

__device__ __forceinline__ void write_miss_outputs(const FrameArgs &p, long long pix)
{
   if (p.hit_id) p.hit_id[pix] = -1;
   if (p.hit_obj) p.hit_obj[pix] = -1;
   if (p.hit_w) p.hit_w[pix] = 0.0f;
}

}   // namespace

// ---------------------------------------------------------------- generate
// One wave = one 8x8 pixel group at one sample index j; sample slot = (group * ms + j) * 64 + lane.
// A workgroup is GEN_WAVES consecutive waves of ONE band (workgroup b serves band b % N_BANDS, so workgroups that
// run at the same time append to different sub-queues) and appends its rays with one atomic.
constexpr int GEN_WAVES = 16;
__global__ __launch_bounds__(GEN_WAVES * 64) void generate_kernel(const FrameArgs p)
{
   __shared__ unsigned s_cnt[GEN_WAVES + 1];
   const int lane = threadIdx.x & 63;
   const int ms = p.multi_samples;
   const int n_micro = p.micro_end - p.micro_begin;
   const int per_band = (n_micro + N_BANDS - 1) / N_BANDS;
   const int band = (int) (blockIdx.x % (unsigned) N_BANDS);
   const unsigned wave_in_band = (blockIdx.x / (unsigned) N_BANDS) * GEN_WAVES + (threadIdx.x >> 6);   // < 2^31: launch_generate
   int unit, j;
   fast_divmod(wave_in_band, p.div_ms, unit, j);
   const int local_micro = band * per_band + unit;
   const bool wave_ok = unit < per_band && local_micro < n_micro;
   const unsigned slot = ((unsigned) local_micro * (unsigned) ms + (unsigned) j) * 64u + (unsigned) lane;
   int px, py;
   long long opix;
   const bool active = micro_pixel(p, p.micro_begin + (wave_ok ? local_micro : 0), lane, px, py, opix) && wave_ok;
   const long long pix = (long long) py * p.width + px;

   // RenderThread.java:39-47
   const float x0 = (float) px * p.x_inc + p.x_min;
   const float y0 = (float) py * p.y_inc + p.y_min;
   float vx = x0, vy = y0;
   if (j != 0) {
      // (j is the same for the whole wave: with the count in a scalar register the skip's affine map is composed on the
      // scalar unit, and a lane only applies it)
      const unsigned n_skip = (unsigned) __builtin_amdgcn_readfirstlane((int) (2u * (unsigned) (j - 1)));
      unsigned long long rng = rng_skip(rng_init(p.rng_seed, (unsigned long long) pix), n_skip);
      vx = x0 + (rng_next_float(rng) * 2.0f - 1.0f) * p.x_inc / 2.0f;
      vy = y0 + (rng_next_float(rng) * 2.0f - 1.0f) * p.y_inc / 2.0f;
   }
   const float vz = -p.near_plane_cam;
   const V3 t = mk3(p.rot[0] * vx + p.rot[1] * vy + p.rot[2] * vz, p.rot[3] * vx + p.rot[4] * vy + p.rot[5] * vz,
                    p.rot[6] * vx + p.rot[7] * vy + p.rot[8] * vz);
   const V3 d = normalize3(normalize3(t));   // RenderThread.java:47, then the Ray ctor (Ray.java:12)
   const V3 o = mk3(p.origin[0], p.origin[1], p.origin[2]);
   Ray r;
   make_ray(r, o, d);

   // Camera.java:205: object.intersects(ray) for each object; a ray that enters no root box is a miss
   bool any_root = false;
   for (int i = 0; i < p.sc.n_objs; i++) {
      const DevObject &ob = p.sc.objs[i];
      float te;
      any_root = any_root || slab_test(r, p.sc.near_plane, p.sc.far_plane, ob.bmin[0], ob.bmin[1], ob.bmin[2], ob.bmax[0],
                                       ob.bmax[1], ob.bmax[2], te);
   }
   if (p.sc.n_analytic > 0) any_root = true;   // spheres have their own bounding test (BoundingSphere.java:29-102): analytic_kernels.hip
   const bool push = active && any_root;
   // A sample whose ray enters no root box has the ambient colour (RenderThread.java: the lighting model's answer for no
   // hit).  It is not stored: one bit per sample slot says which slots became rays, and resolve supplies the colour of
   // the rest (on a 4K frame around a statue that is nine samples in ten: 260 MB less to write, and to read back, per pass).
   if (active && !any_root && j == 0) write_miss_outputs(p, opix);
   {
      const unsigned long long qm = __ballot(push);
      if (lane == 0 && wave_ok) p.queued[(size_t) local_micro * (size_t) ms + (size_t) j] = qm;
   }
   // the pass's 8x8 groups are cut into 8 bands (image order); each band has its own sub-queue
   const unsigned idx = (unsigned) band * p.band_cap + block_queue_push<GEN_WAVES>(p.counters + CNT_Q0 + band, push, s_cnt);
   if (push) store_ray(p.rays0, idx, o, d, slot, -1);
   if (p.stats) {
      add_stat(p.counters, ST_PRIMARY, active ? 1u : 0u);
      add_stat(p.counters, ST_BOX, active ? (unsigned) p.sc.n_objs : 0u);
   }
}

// min / max as the instructions themselves.  Through fminf / fmaxf the compiler first quiets every operand it cannot prove
// free of signalling NaNs (a v_max_f32 x, x, x each: twelve of them in the packed node step, whose operands are halves of
// packed products); these are for values that are known not to be NaN at all (the packed step's argument), where
// v_min / v_max / v_min3 / v_max3 give what fminf / fmaxf give.
#ifndef WALK_RAW_MINMAX
#define WALK_RAW_MINMAX 1
#endif
__device__ __forceinline__ float raw_min(float a, float b)
{
#if WALK_RAW_MINMAX
   float r;
   asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
   return r;
#else
   return fminf(a, b);
#endif
}
__device__ __forceinline__ float raw_max(float a, float b)
{
#if WALK_RAW_MINMAX
   float r;
   asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
   return r;
#else
   return fmaxf(a, b);
#endif
}
__device__ __forceinline__ float raw_min3(float a, float b, float c)
{
#if WALK_RAW_MINMAX
   float r;
   asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
   return r;
#else
   return fminf(fminf(a, b), c);
#endif
}
__device__ __forceinline__ float raw_max3(float a, float b, float c)
{
#if WALK_RAW_MINMAX
   float r;
   asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
   return r;
#else
   return fmaxf(fmaxf(a, b), c);
#endif
}

// ---------------------------------------------------------------- the walker
// One kernel walks every kind of ray:
//   MODE_CLOSEST     Camera.getClosestIntersection (Camera.java:193-215) for the rays of a queue;
//                    aux of a ray = the mirror object to skip (-1: none) in its low 16 bits
//   MODE_SHADOW      RTStatics.shadowIntersects (RTStatics.java:617-639) for the paths of the
//                    shadow list: per object the closest accepted hit, then the before-the-light test
//   MODE_SHADOW_ANY  the same where the host proved that any accepted hit occludes (api.cpp)
//   MODE_FUSED       MODE_CLOSEST, and each lane whose ray hit something goes straight on to walk that hit's
//                    shadow ray (as MODE_SHADOW_ANY would) before it takes another ray: no shadow queue pass,
//                    no second persistent launch with a tail of its own (long passes only: api.cpp)
//
// Persistent wave64s.  A LANE that has finished its ray takes the next one from the queue without
// waiting for the other 63 (rays differ several-fold in length; batch-at-a-time left 5 of 6 lanes
// idle).  A lane is therefore a small state machine: {ray, current object k, cursor `cur`, stack
// depth, best of object k, best of the scene}.  All objects share one node array and one slot array
// (links are absolute), so lanes of one wave may be inside different objects.  In scenes of small leaves
// the machine has two sides -- `cur` walks the nodes while `lf` tests a leaf's faces, both in the same
// trip ("two sides" below); in scenes whose leaves hold tens of faces there is one cursor and the WAVE
// tests a leaf's chunk ("burst").
//
// The queue is cut into 8 contiguous ranges with one cursor each; a wave starts on range
// blockIdx % 8 (workgroups b and b + 8 share an XCD, hence an L2) and moves on to the next range
// when its own is drained.  Queue order follows image order, so each L2 mostly sees one band of the
// image.  Placement only affects speed: every ray is claimed exactly once whatever runs where.
// lane roles
enum { ROLE_IDLE = 0, ROLE_OWNER = 1, ROLE_HELPER = 2 };
#ifndef WALK_WAVES_PER_SIMD
#define WALK_WAVES_PER_SIMD 6   // closest-hit and fused walks, two-sided form: 80 VGPRs without scratch (3.61 ms at 5 waves, 3.51 at 6)
#endif
#ifndef WALK_SHADOW_WAVES_PER_SIMD
#define WALK_SHADOW_WAVES_PER_SIMD 5   // the shadow forms carry more state (held to 80 VGPRs the exact form spills 32 bytes per lane)
#endif
#ifndef WALK_BIG_WAVES_PER_SIMD
#define WALK_BIG_WAVES_PER_SIMD 5   // (6 waves at 80 VGPRs spill the burst's two passes in flight: Lucy-class 4K frame 54.1 ms at 5, 64.5 at 6)
#endif
#ifndef WALK_SHADOW_BLOCK
#define WALK_SHADOW_BLOCK (1 << 13)   // power of two, a multiple of WALK_SHADOW_SPREAD (4K x 16 spp in one pass, shadow walk: Lucy-class 25.0 ms unblocked, 21.5 at 2^17, 20.3 at 2^13, 19.7 at 2^9; Thai-class 6.5 / 6.1 / 5.7 / 6.2; Dragon-class 0.92 / 0.92 / 0.92 / 1.33)
#endif
#ifndef WALK_SHADOW_SPREAD
#define WALK_SHADOW_SPREAD 64   // power of two
#endif
#ifndef WALK_REFILL_MIN
#define WALK_REFILL_MIN 28   // (re-tuned on the round's last kernels with WALK_RESOLVE_MIN: 24 / 12 before; -0.8 % on the bench frame, -1.2 % with two frames in flight)
#endif
#ifndef WALK_RESOLVE_MIN
#define WALK_RESOLVE_MIN 20
#endif
#ifndef WALK_ADMIN_MIN
#define WALK_ADMIN_MIN 6
#endif
#ifndef WALK_BURST_MIN
#define WALK_BURST_MIN 8
#endif
#ifndef WALK_BURST_NODE_FEW
#define WALK_BURST_NODE_FEW 8
#endif
#ifndef WALK_LEAF_BATCH
#define WALK_LEAF_BATCH 2
#endif
#ifndef WALK_TAIL_STEAL_MIN
#define WALK_TAIL_STEAL_MIN 24   // (with the two-sided trip a steal round costs more than the few thieves of a nearly busy wave gain: bench frame 3.47 ms at 1, 3.38 at 8, 3.33 at 16, 3.25 at 24, 3.32 at 32, 3.29 with no stealing at all)
#endif
#ifndef WALK_NODE_REPS
#define WALK_NODE_REPS 1
#endif
#ifndef WALK_FLAT_CONTROL
#define WALK_FLAT_CONTROL 1
#endif
#ifndef WALK_SIDE_MIN
#define WALK_SIDE_MIN 12   // (measured flat from 1 to 28 on the bench frame)
#endif
constexpr int LEAF_BATCH = WALK_LEAF_BATCH;        // faces of a leaf tested per trip
static_assert(LEAF_BATCH == 2, "a leaf trip fetches two faces");

// static LDS of one walker workgroup: the stack, the steal list, and the waiting face's leaf and barycentrics
constexpr int WALK_LDS_BYTES = (STACK_SLOTS + 1 + 1 + 2) * BLOCK_THREADS * 4;   // 26 KiB; six workgroups per CU: 156 of 160 KiB
constexpr int walk_waves_per_simd(int stats, int mode, int big)
{
   return big ? WALK_BIG_WAVES_PER_SIMD : (stats > 1 || mode == MODE_SHADOW || mode == MODE_SHADOW_ANY) ? WALK_SHADOW_WAVES_PER_SIMD : WALK_WAVES_PER_SIMD;
}

template <int STATS, int MODE, int BIG>   // STATS: 0 none, 1 counts + timeline (light), 2 + walker diagnostics (heavy, changes occupancy); BIG: leaves of tens of faces (see "burst" below)
__global__ __launch_bounds__(BLOCK_THREADS) __attribute__((amdgpu_waves_per_eu(walk_waves_per_simd(STATS, MODE, BIG), walk_waves_per_simd(STATS, MODE, BIG)))) void walk_kernel(const WalkArgs a)
{
   __shared__ int s_stack[STACK_SLOTS * BLOCK_THREADS];
   __shared__ int s_list[BLOCK_THREADS];                              // per wave: compacted victim lanes
   __shared__ int s_pkey[BLOCK_THREADS];                              // per lane: the leaf descriptor its waiting face (`pend`) was met in
   __shared__ float s_puv[2 * BLOCK_THREADS];                         // two-sided form: that face's barycentrics, as the leaf step worked them out
   // the tuning rests on this many workgroups fitting a CU's LDS: a word added above must not silently cost a wave per SIMD
   static_assert(WALK_LDS_BYTES * (walk_waves_per_simd(STATS, MODE, BIG) * 4 / (BLOCK_THREADS / 64)) <= 160 * 1024,
                 "the walker's workgroups per CU no longer fit 160 KB of LDS");
   const int lane = threadIdx.x & 63;
   const int wave_base = threadIdx.x & ~63;
   int *__restrict__ stack = s_stack + threadIdx.x;
   volatile int *list = s_list + wave_base;
   uint32_t *cursors = a.counters + a.cursor_index;
   const float4 *__restrict__ nodes = (const float4 *) a.sc.nodes;
   const float4 *__restrict__ gfaces = (const float4 *) a.sc.gfaces;
   const float4 *__restrict__ descs = (const float4 *) a.sc.descs;
   const float nearp = a.sc.near_plane, farp = a.sc.far_plane, eps = a.sc.epsilon;
   const V3 light = mk3(a.light[0], a.light[1], a.light[2]);
   const int n_objs = a.sc.n_objs;
   const float abs_max = a.sc.scene_abs_max;

   // A queue much shorter than the pass's samples promised (a statue that fills a tenth of a 4K frame) wants fewer waves:
   // every wave in flight at the end is a ray started late (launch sizes: blocks_per_cu_for).
   if (a.trim_unit != 0u) {
      unsigned total = 0;
      for (int b = 0; b < N_BANDS; b++) total += a.counters[a.count_index + b];
      const unsigned per_cu = (total + a.trim_unit / 2u) / a.trim_unit;
      const unsigned useful = (per_cu > a.trim_lo ? per_cu : a.trim_lo) * a.trim_step;
      if (blockIdx.x >= useful) return;
   }
   // wave-uniform
   const int home = (int) (blockIdx.x % (unsigned) N_BANDS);
   int attempt = 0;
   bool exhausted = false;
   // per lane
   // the small per-lane fields share one register (the kernel sits at the 80-VGPR step for 6 waves/SIMD)
   struct {
      unsigned role : 2, owner : 6, pending : 6;
      int k : 7, mirror : 7;   // -1 .. MAX_OBJECTS - 1
      unsigned plain : 1;      // finite origin and 1/direction (and finite node boxes): no NaN can arise in the slab test
      unsigned unsaved : 1;    // closest modes, ROLE_IDLE: the finished ray's result is still in registers (see "retire" below)
      unsigned shadow : 1;     // MODE_FUSED: this lane walks (or helps walk) a shadow ray
      unsigned redo : 1;       // MODE_SHADOW: the object is being walked again without the two shortcuts (device_util.h)
   } st;
   constexpr bool CLOSEST = MODE == MODE_CLOSEST || MODE == MODE_FUSED;   // rays come from the queue; faces wait in `pend`
   constexpr bool FUSED = MODE == MODE_FUSED;
   constexpr bool DEFER = CLOSEST || BIG != 0;   // faces that pass the triangle tests wait in `pend` for the resolve block
   constexpr bool DUAL = BIG == 0;   // the two-sided trip ("two sides" below); the burst form keeps one cursor
   // Thresholds, measured on the bench frame and its shares: they were launch arguments while they were being
   // tuned; as constants they cost no scalar registers (the kernel is short of them: it spills scalars into vector lanes).
   constexpr int REFILL_MIN = CLOSEST ? WALK_REFILL_MIN : 64;   // refill once this many lanes are idle (or all of them); shadow waves refill whole
   constexpr int STEAL_MIN = CLOSEST ? 32 : 64;      // while the queue still has rays: steal once this many lanes are idle
   constexpr int TAIL_STEAL_MIN = WALK_TAIL_STEAL_MIN;                // once the queue is dry: steal with this many idle lanes
   constexpr int AGE_MIN = 8;                        // shadow modes: idle lanes help a ray that has been walked for this many trips at once
   constexpr int SIDE_MIN = WALK_SIDE_MIN;           // two-sided trip: a side with fewer lanes than this sits the trip out, unless the other does
   constexpr int RESOLVE_MIN = CLOSEST ? WALK_RESOLVE_MIN : 1;   // faces that passed the triangle tests wait until this many lanes hold one
   constexpr int BURST_MIN = WALK_BURST_MIN, BURST_NODE_FEW = WALK_BURST_NODE_FEW;   // BIG: see "burst"
   constexpr int ADMIN_MIN = WALK_ADMIN_MIN;         // lanes whose walk has ended wait until this many do before the bookkeeping block runs
   static_assert(MAX_OBJECTS <= 64, "k and mirror are 7-bit fields");
   st.role = ROLE_IDLE;
   st.owner = lane;      // the lane whose ray this lane is walking (itself unless ROLE_HELPER)
   st.pending = 0;       // ROLE_OWNER: helpers still walking parts of my current object
   unsigned qi = 0;
   Ray r;
   make_ray(r, mk3(0, 0, 0), mk3(0, 0, -1));
   st.mirror = -1;
   st.k = -1;
   st.plain = 0;
   st.unsaved = 0;
   st.shadow = 0;
   st.redo = 0;
   float safe_w = 0.0f;   // MODE_SHADOW: an accepted face nearer than this ends the walk of its object (shadow_safe_w)
   int cur = STACK_EMPTY;
   // Two sides (DUAL): the reference visits every leaf whose boxes the ray meets, so nothing obliges a lane to finish a leaf
   // before it walks on.  `cur` is the node side (a node; a leaf waiting for the leaf side to come free; STACK_EMPTY with an
   // empty stack), `lf` the leaf side (~descriptor of the leaf being tested, 0: none).  A trip runs a node step AND two
   // faces of the leaf for such a lane; what the node side does ahead of a face that turns out to be hit is pruned when
   // the answer is in, exactly as a stolen subtree's is.
   int lf = 0;
   int wg = 0;          // a leaf that waits in `cur` straight from its parent's record has its descriptor words with it:
   unsigned wmask = 0;  // group word and face mask (wmask == 0: it came off the stack, its descriptor is to be read)
   int sp = 0, bot = 0;   // live stack entries are [bot, sp): thieves take from the bottom
   int age = 0;           // shadow modes: trips spent on the current ray (helpers inherit their victim's)
   int pend = -1;         // MODE_CLOSEST: a face of the current object that passed the triangle tests and whose point and
                          // distance are still to be computed: its group-face entry | DESC_IDENT of its group (the leaf it
                          // was met in waits in s_pkey); see "resolve" below
   // leaves (layout.h "LEAVES"): cur < 0 is the leaf descriptor ~cur.  todo == 0: the descriptor is still to be read; else
   // the faces of the leaf (bits over its group's faces) still to be tested.  gb: the descriptor's gbase word (group base
   // and flags); missed: the faces of that group this lane's ray has been tested against and missed -- they miss in
   // every leaf, so a leaf's work is its mask & ~missed.
   int gb = -1;
   unsigned missed = 0, todo = 0;
   bool resolve_req = false;   // wave-uniform: some lane found a second such face
   float lim2 = 0.0f;
   float ob_w = 0.0f, best_w = 0.0f;
   int ob_gf = -1, ob_key = 0, best_gf = -1, best_obj = -1;   // *_gf: group-face entry | DESC_IDENT
   bool occluded = false;
   // The fold of KDNode.getChildIntersection restated order-independently (header comment): a candidate beats the object's
   // best so far when it is nearer; at equal distance when its leaf comes later in the reference's order (key = the leaf's
   // first slot); in one leaf, when its slot comes earlier (KDNode.java:110-118: the first face wins).  Slots of one leaf:
   // a leaf that is its own group lists its faces in slot order; else the leaf's slot ids are searched (two DIFFERENT
   // faces of one small leaf at exactly one distance: rays through a shared vertex or edge).
   auto beats = [&](float dist, int key, int gfx) -> bool {
      if (ob_gf < 0 || dist < ob_w) return true;
      if (dist != ob_w) return false;
      if (key != ob_key) return key > ob_key;
      if (((gfx ^ ob_gf) & DESC_BASE_MASK) == 0) return false;
      if (gfx & DESC_IDENT) return (gfx & DESC_BASE_MASK) < (ob_gf & DESC_BASE_MASK);
      const int f_new = __float_as_int(gfaces[3 * (size_t) (gfx & DESC_BASE_MASK)].w), f_old = __float_as_int(gfaces[3 * (size_t) (ob_gf & DESC_BASE_MASK)].w);
      for (int j = 0; j < 2 * GROUP_FACES; j++) {
         const int f = a.sc.slot_ids[key + j];
         if (f == f_new) return true;
         if (f == f_old) return false;
      }
      return false;
   };
   auto over = [&]() -> bool { return cur == STACK_EMPTY && (!DUAL || lf == 0); };   // this lane's walk of its object (or subtree) has ended
   unsigned n_box = 0, n_tri = 0, n_hit = 0, n_ray = 0, n_spawn = 0;
   unsigned n_box_s = 0, n_tri_s = 0;   // MODE_FUSED: tests made for shadow rays
   unsigned d_ray_trips = 0, d_ray_trips_max = 0;
   unsigned d_hist[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
   unsigned d_node_lane = 0, d_node_wave = 0, d_leaf_lane = 0, d_leaf_wave = 0, d_outer = 0, d_refill = 0, d_steal = 0;
   unsigned d_repeat = 0;   // diagnostics: faces a leaf held that the ray had already missed in its group (tests saved)
   const unsigned long long t_begin = STATS ? wall_clock64() : 0ull;   // 100 MHz
   unsigned long long t_exhausted = 0ull;
   unsigned d_tail_trips = 0;   // wave trips after the queue ran dry

   for (;;) {
      if (STATS > 1 && lane == 0) d_outer++;
      if (STATS && exhausted) d_tail_trips++;
      // ---- resolve: most faces fail the triangle tests; computing the point and distance of the one or two per wave
      // and trip that pass, right there, ran some forty instructions for a lane or two twice a trip.  The leaf loop
      // only notes the slot; here, once enough lanes hold one (or a lane needs its answer: its walk of the object is
      // over, or it found another face), all of them are worked out together and folded into the lane's best.  The
      // fold does not depend on when it happens (see the tie rule), and slots of one leaf still arrive in slot order.
      if (DEFER) {
         const unsigned long long pm = __ballot(pend >= 0);
         if (pm != 0ull && (resolve_req || exhausted || (int) __popcll(pm) >= RESOLVE_MIN || __ballot(pend >= 0 && over()) != 0ull)) {
            const float sd = st.k >= 0 ? a.sc.objs[st.k].shadow_distance : 0.0f;
            if (pend >= 0) {
               const float4 *t = gfaces + 3 * (size_t) (pend & DESC_BASE_MASK);
               const float4 A = t[0], B = t[1], C = t[2];
               float u, v, dist;
               V3 P;
               if (DUAL) tri_point(A, B, C, r, s_puv[threadIdx.x], s_puv[BLOCK_THREADS + threadIdx.x], P, dist);
               else (void) tri_test(A, B, C, r, eps, u, v, P, dist);
               const int key = __float_as_int(descs[(size_t) s_pkey[threadIdx.x]].x);
               if ((dist > sd) && beats(dist, key, pend)) {
                  ob_w = dist;
                  ob_gf = pend;
                  ob_key = key;
                  lim2 = fminf(lim2, prune_limit2(r, dist, abs_max));
                  // any accepted face occludes (any-hit forms), or one safely before half way to the light
                  // (device_util.h, shortcut 1): the walk ends here
                  if ((FUSED && st.shadow) || MODE == MODE_SHADOW_ANY || (MODE == MODE_SHADOW && dist < safe_w)) {
                     sp = bot;
                     cur = STACK_EMPTY;
                     lf = 0;
                     todo = 0;
                  }
               }
            }
            pend = -1;
            resolve_req = false;
         }
      }
      // The bookkeeping below serves lanes whose walk has just ended.  One ends somewhere in the wave nearly every trip, and
      // the block is a hundred-odd instructions for the whole wave: ended lanes wait until admin_min of them do (or the
      // idle ones suffice for a refill, or nothing else is left to walk, or the queue is dry and every lane counts).
      const unsigned long long ended = __ballot(over());
      const int n_serve = __popcll(__ballot(over() && st.role != ROLE_IDLE));
      if (ended != 0ull && (exhausted || n_serve >= ADMIN_MIN || ended == ~0ull || (n_serve == 0 && (int) __popcll(ended) >= REFILL_MIN))) {
         // ---- (1) helpers that finished their subtree hand (w, slot, leaf) to the ray's owner, which
         // folds it in with the same order-independent rule the leaves use
         unsigned long long fin = __ballot(st.role == ROLE_HELPER && over());
         while (fin != 0ull) {
            const int h = __ffsll((unsigned long long) fin) - 1;
            fin &= fin - 1ull;
            // h is wave-uniform: v_readlane (a few cycles) instead of a ds_bpermute round trip per value
            const int o = __builtin_amdgcn_readlane((int) st.owner, h);
            const float hw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ob_w), h));
            const int hs = __builtin_amdgcn_readlane(ob_gf, h), hk = __builtin_amdgcn_readlane(ob_key, h);
            if (lane == o) {
               --st.pending;
               if (hs >= 0 && beats(hw, hk, hs)) {
                  ob_w = hw;
                  ob_gf = hs;
                  ob_key = hk;
                  lim2 = fminf(lim2, prune_limit2(r, hw, abs_max));
               }
            }
         }
         if (st.role == ROLE_HELPER && over()) {
            st.role = ROLE_IDLE;
            st.owner = lane;
         }

         // ---- (2) an owner whose walk of object k has ended (and whose helpers are all back): fold the
         // object's result, enter the next object or retire the ray
         if (st.role == ROLE_OWNER && over() && st.pending == 0) {
            bool redo_k = false;
            if (st.k >= 0 && ob_gf >= 0) {
               if (CLOSEST && !(FUSED && st.shadow)) {
                  if ((ob_w > eps) && (best_obj < 0 || ob_w < best_w)) {   // Camera.java:208
                     best_w = ob_w;
                     best_gf = ob_gf;
                     best_obj = st.k;
                  }
               } else if (MODE == MODE_SHADOW_ANY || FUSED) {
                  occluded = true;
               } else if (ob_w > eps) {   // RTStatics.java:628-633
                  if (ob_w < safe_w) {
                     occluded = true;   // shortcut 1 (device_util.h): the closest face is at most this far, and passes
                  } else {
                     const float4 *t = gfaces + 3 * (size_t) (ob_gf & DESC_BASE_MASK);
                     float u, v, dist;
                     V3 P;
                     (void) tri_test(t[0], t[1], t[2], r, eps, u, v, P, dist);
                     if (ob_w < distance3(P, light) + eps) {
                        // after the full walk (redo) this IS the closest face; otherwise a skipped box might have held a
                        // closer one that fails the test: walk this object again without the shortcuts
                        if (st.redo) occluded = true;
                        else redo_k = true;
                     }
                  }
               }
            }
            if (MODE == MODE_SHADOW) st.redo = redo_k ? 1 : 0;
            int nk = redo_k ? st.k : st.k + 1;
            if (CLOSEST && nk == st.mirror) ++nk;
            const bool done = (nk >= n_objs) || (MODE != MODE_CLOSEST && occluded);
            if (!done) {
               st.k = nk;
               const DevObject *ob = a.sc.objs + st.k;
               ob_gf = -1;
               ob_w = 0.0f;
               ob_key = 0;
               sp = bot = 0;
               lim2 = (CLOSEST && best_obj >= 0) ? prune_limit2(r, best_w, abs_max) : __builtin_inff();
               if (MODE == MODE_SHADOW) safe_w = (!st.redo && ob->shadow_distance >= eps) ? shadow_safe_w(r, light, abs_max) : 0.0f;
               float te;
               if (STATS) (FUSED && st.shadow ? n_box_s : n_box)++;
               if (slab_test(r, nearp, farp, ob->bmin[0], ob->bmin[1], ob->bmin[2], ob->bmax[0], ob->bmax[1], ob->bmax[2], te)) {
                  cur = ob->root_entry;   // KDTree.getChildIntersection (bounds/KDTree.java:59-65)
                  wmask = 0u;
                  todo = 0;
               }
            } else {
               if (CLOSEST && !(FUSED && st.shadow)) st.unsaved = 1;
               else a.mat0[4 * (size_t) qi + 3] = occluded ? 1.0f : 0.0f;
               st.role = ROLE_IDLE;
            }
         }

         // ---- retire: the winner's point, barycentrics and normal are recomputed once (same arithmetic => same bits)
         // and the hit record is written.  A ray finishes somewhere in the wave every other trip, and doing this for one
         // lane at a time cost a fifth of the kernel's instructions.  A finished lane keeps its result in registers
         // instead (nothing touches an idle lane's ray or best-so-far) until the wave is about to hand idle lanes new
         // work -- a refill or a steal -- and then all of them are written in one go.
         if (CLOSEST) {
            const unsigned long long um = __ballot(st.unsaved != 0);
            if (um != 0ull) {
               const unsigned long long idle_now = __ballot(st.role == ROLE_IDLE);
               const int n_idle = __popcll(idle_now);
               const bool refill_now = !exhausted && (n_idle >= REFILL_MIN || idle_now == ~0ull);
               const bool steal_now = exhausted || n_idle >= STEAL_MIN;
               if ((refill_now || steal_now) && st.unsaved) {
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
                     h1 = make_float4(wb * n0.x + u * n1.x + v * n2.x, wb * n0.y + u * n1.y + v * n2.y,
                                      wb * n0.z + u * n1.z + v * n2.z, __int_as_float(face));
                     if (STATS) n_hit++;
                  }
                  float4 *hq = (float4 *) a.hits + 2 * (size_t) qi;
                  hq[0] = h0;
                  hq[1] = h1;
                  a.hit_obj[qi] = best_obj;
                  st.unsaved = 0;
                  if (FUSED && best_obj >= 0) {
                     // the lane goes on with this hit's shadow ray (RTStatics.java:619-620, then the Ray ctor;
                     // RTStatics.offsetPosition): the same arithmetic on the same bits as a shadow launch reading the record
                     const V3 P = mk3(h0.x, h0.y, h0.z), N = mk3(h1.x, h1.y, h1.z);
                     make_ray(r, add3(P, scale3(N, eps)), normalize3(normalize3(sub3(light, P))));
                     st.plain = plain_ray(r) && a.sc.boxes_finite;
                     st.role = ROLE_OWNER;
                     st.owner = lane;
                     st.pending = 0;
                     st.k = -1;
                     st.mirror = -1;
                     st.shadow = 1;
                     sp = bot = 0;
                     best_obj = -1;
                     best_gf = -1;
                     best_w = 0.0f;
                     occluded = false;
                     age = 0;
                     gb = -1;   // another ray: nothing is known to miss
                     missed = 0;
                     if (STATS) n_spawn++;
                  }
               }
            }
         }

         // ---- (3) idle lanes take new rays from the queue
         const unsigned long long idle = __ballot(st.role == ROLE_IDLE);
         if (idle != 0ull && !exhausted && ((int) __popcll(idle) >= REFILL_MIN || idle == ~0ull)) {
            const unsigned want_n = (unsigned) __popcll(idle);
            if (STATS > 1 && lane == 0) d_refill++;
            // A band's queue of n rays is dealt out as a rows x spread matrix read down the columns: logical index L
            // (what the cursor counts) -> ray (L % spread) * rows + L / spread.  Neighbouring rays (a grazing patch of
            // the surface means long walks for all of them) then land in different waves instead of making one wave
            // the launch's tail.  Shadow rays only: primary rays lose more from the lost coherence than they gain (measured).
            constexpr unsigned SPREAD = CLOSEST ? 1u : (unsigned) WALK_SHADOW_SPREAD;
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
               if (STATS) t_exhausted = wall_clock64();
            } else if (st.role == ROLE_IDLE) {
               const unsigned logical = base + (unsigned) rank_below(idle);
               // (the matrix is per block of WALK_SHADOW_BLOCK rays, so that what the chip works on at one time stays a patch
               // of the image however long the queue is: with a 4K x 16 spp frame as ONE pass and the whole band as one matrix
               // the Lucy-class shadow launch read 185 GB, L2 hit 0.39, against 8 x 13 GB as eight passes)
               unsigned my;
               bool mine;
               if (CLOSEST) {
                  // (SPREAD is 1 here: the cursor counts rays.  The dealing, if any, is a launch argument -- walk_args.h: per
                  // block the largest whole matrix of 2^sl columns is read down its columns, the block's last few rays in order)
                  const unsigned sl = a.primary_spread & 0xffu, bl = (a.primary_spread >> 8) & 0xffu;
                  const unsigned blk0 = (logical >> bl) << bl, off = logical - blk0;
                  const unsigned bsize = blk0 < count ? min(1u << bl, count - blk0) : 0u;
                  const unsigned rows_f = bsize >> sl, full = rows_f << sl;
                  const unsigned dealt = (off & ((1u << sl) - 1u)) * rows_f + (off >> sl);
                  my = lo + blk0 + (a.primary_spread != 0u && off < full ? dealt : off);
                  mine = logical < count;
               } else {
                  constexpr unsigned BLK = (unsigned) WALK_SHADOW_BLOCK;
                  const unsigned blk0 = logical & ~(BLK - 1u), off = logical & (BLK - 1u);
                  const unsigned bsize = blk0 < count ? min(BLK, count - blk0) : 0u;
                  const unsigned rows_b = (bsize + SPREAD - 1u) / SPREAD;
                  const unsigned idx = (off % SPREAD) * rows_b + off / SPREAD;
                  my = lo + blk0 + idx;
                  mine = logical < rows * SPREAD && idx < bsize;
               }
               if (mine) {
                  qi = my;
                  V3 o, d;
                  if (CLOSEST) {
                     const float4 *q = (const float4 *) a.rays + 2 * (size_t) my;
                     const float4 r0 = q[0], r1 = q[1];
                     o = mk3(r0.x, r0.y, r0.z);
                     d = mk3(r1.x, r1.y, r1.z);
                     st.mirror = (int) (short) (__float_as_int(r1.w) & 0xffff);
                  } else {
                     qi = a.shadow[my];   // the path
                     const float4 *hq = (const float4 *) a.hits0 + 2 * (size_t) qi;
                     const float4 h0 = hq[0], h1 = hq[1];
                     const V3 P = mk3(h0.x, h0.y, h0.z), N = mk3(h1.x, h1.y, h1.z);
                     d = normalize3(normalize3(sub3(light, P)));   // RTStatics.java:619-620, then the Ray ctor
                     o = add3(P, scale3(N, eps));                  // RTStatics.offsetPosition
                     st.mirror = -1;
                  }
                  make_ray(r, o, d);
                  st.plain = plain_ray(r) && a.sc.boxes_finite;
                  st.role = ROLE_OWNER;
                  st.owner = lane;
                  st.pending = 0;
                  st.k = -1;
                  st.shadow = 0;
                  st.redo = 0;
                  sp = bot = 0;
                  best_obj = -1;
                  best_gf = -1;
                  best_w = 0.0f;
                  occluded = false;
                  age = 0;
                  gb = -1;
                  missed = 0;
                  if (STATS) n_ray++;
                  if (STATS > 1) {
                     d_ray_trips_max = max(d_ray_trips_max, d_ray_trips);
                     if (n_ray > 1) {
                        const int bin = d_ray_trips < 16 ? 0 : min(7, 28 - __clz(d_ray_trips));
#pragma unroll
                        for (int q = 0; q < 8; q++) d_hist[q] += (q == bin) ? 1u : 0u;
                     }
                     d_ray_trips = 0;
                  }
               }
            }
         }

         // ---- (4) lanes still idle steal work inside the wave: the BOTTOM stack entry (the largest
         // pending subtree) of a busy lane, walked with a copy of that lane's ray.  Which lane walks a
         // subtree cannot change the result: the fold is order independent (see the header comment).
         const unsigned long long thieves = __ballot(st.role == ROLE_IDLE && !st.unsaved);   // an unsaved lane still holds its ray
         // Before the queue is dry a steal needs steal_min idle lanes (they would rather wait for a refill) -- except,
         // for shadow rays, from a ray that has been walked for age_min trips already: those few long rays are what
         // the launch ends up waiting for, so they get help as soon as a lane is free.
         const bool steal_any = (exhausted && (int) __popcll(thieves) >= TAIL_STEAL_MIN) || (int) __popcll(thieves) >= STEAL_MIN;
         const unsigned long long victims = __ballot(st.role != ROLE_IDLE && sp > bot && (steal_any || (!CLOSEST && age >= AGE_MIN)));   // measured: no gain for queue rays, fused or not
         if (thieves != 0ull && victims != 0ull) {
            const int n_pairs = min(__popcll(thieves), __popcll(victims));
            const int v_rank = rank_below(victims), t_rank = rank_below(thieves);
            const bool is_victim = ((victims >> lane) & 1ull) != 0ull && v_rank < n_pairs;
            const bool is_thief = st.role == ROLE_IDLE && !st.unsaved && t_rank < n_pairs;
            if (is_victim) list[v_rank] = lane;
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the list is written before it is read (one wave, in order)
            const int v = is_thief ? list[t_rank] : lane;
            const int v_bot = __shfl(bot, v, 64);
            const int entry = s_stack[min(v_bot, STACK_SLOTS - 1) * BLOCK_THREADS + wave_base + v];
            const int v_owner = __shfl(st.owner, v, 64), v_k = __shfl(st.k, v, 64);
            const float v_lim2 = __shfl(lim2, v, 64);
            const float ox = __shfl(r.o.x, v, 64), oy = __shfl(r.o.y, v, 64), oz = __shfl(r.o.z, v, 64);
            const float dx = __shfl(r.d.x, v, 64), dy = __shfl(r.d.y, v, 64), dz = __shfl(r.d.z, v, 64);
            const float ix = __shfl(r.inv.x, v, 64), iy = __shfl(r.inv.y, v, 64), iz = __shfl(r.inv.z, v, 64);
            const int v_age = CLOSEST ? 0 : __shfl(age, v, 64);
            const int v_shadow = FUSED ? __shfl((int) st.shadow, v, 64) : 0;
            const int v_redo = MODE == MODE_SHADOW ? __shfl((int) st.redo, v, 64) : 0;
            const float v_safe_w = MODE == MODE_SHADOW ? __shfl(safe_w, v, 64) : 0.0f;
            if (is_victim) ++bot;
            if (is_thief) {
               st.role = ROLE_HELPER;
               st.owner = v_owner;
               st.k = v_k;
               st.shadow = (unsigned) v_shadow;
               st.redo = (unsigned) v_redo;
               safe_w = v_safe_w;
               lim2 = v_lim2;
               r.o = mk3(ox, oy, oz);
               r.d = mk3(dx, dy, dz);
               r.inv = mk3(ix, iy, iz);
               st.plain = plain_ray(r) && a.sc.boxes_finite;
               age = v_age;
               cur = entry;
               lf = 0;
               wmask = 0u;
               todo = 0;
               sp = bot = 0;
               ob_gf = -1;
               ob_w = 0.0f;
               ob_key = 0;
               gb = -1;
               missed = 0;
               if (STATS > 1) d_steal++;
            }
            unsigned long long tm = __ballot(is_thief);
            while (tm != 0ull) {
               const int t = __ffsll((unsigned long long) tm) - 1;
               tm &= tm - 1ull;
               const int o = __builtin_amdgcn_readlane((int) st.owner, t);
               if (lane == o) ++st.pending;
            }
         }
         // prune limits found on the owner's side reach its helpers here
         lim2 = fminf(lim2, __shfl(lim2, st.owner, 64));

         if (__ballot(st.role != ROLE_IDLE) == 0ull) {
            if (exhausted) break;
            continue;
         }
      }

      if (DUAL) {
         // ---- two sides: a node step for the lanes at a node, then two faces for the lanes that hold a leaf (a lane
         // can be both).  Fetches are one after the other -- node records, then faces -- so that they share registers.
         if (STATS > 1 && st.role == ROLE_OWNER) d_ray_trips++;
         if (!CLOSEST) ++age;
         // (WALK_NODE_REPS node steps per trip: a ray's chain of trips is mostly its chain of node records -- a primary ray that
         // hits the surface tests some two hundred boxes of the reference's hierarchy, two per step)
         bool ran_node = false;
#pragma unroll
         for (int rep = 0; rep < WALK_NODE_REPS; rep++) {
            if (lf == 0 && cur < 0) {   // the leaf side is free: the waiting leaf moves over
               lf = cur;
               todo = 0u;   // (one that came off the stack: its descriptor is read first)
               if (wmask != 0u) {
                  if (((wg ^ gb) & DESC_BASE_MASK) != 0) missed = 0u;   // another group
                  gb = wg;
                  todo = wmask & ~missed;
                  if (todo == 0u && !(wg & DESC_MORE)) lf = 0;   // meanwhile the ray has missed all of its faces next door
                  wmask = 0u;
               }
               if (sp > bot) {
                  --sp;
                  cur = stack[sp * BLOCK_THREADS];
               } else {
                  cur = STACK_EMPTY;
               }
            }
            const bool in_node = (unsigned) cur < (unsigned) STACK_EMPTY;
            const int n_node = __popcll(__ballot(in_node)), n_leaf0 = __popcll(__ballot(lf < 0));
            const bool run_node = n_node > 0 && (exhausted || n_node >= SIDE_MIN || n_node >= n_leaf0);
            ran_node = ran_node || run_node;
            if (run_node) {
               const float4 *pa = nodes + NODE_F4 * (size_t) (in_node ? cur : 0);
               const float4 na = pa[0], nb = pa[1], nc = pa[2], q3 = pa[3], q4 = pa[4];
               const int4 nd = make_int4(__float_as_int(q3.x), __float_as_int(q3.y), __float_as_int(q3.z), __float_as_int(q3.w));
               if (in_node) {
                  if (STATS > 1) {
                     d_node_lane++;
                     if (first_active_lane()) d_node_wave++;
                  }
                  float tl = 0.0f, tr = 0.0f;
                  bool lok = false, rok = false;
                  if (STATS) (FUSED && st.shadow ? n_box_s : n_box) += (nd.x != LINK_NONE ? 1u : 0u) + (nd.y != LINK_NONE ? 1u : 0u);
                  if (st.plain) {   // (the packed form of the step below: same arithmetic, same argument)
                     const f2 ax = f2 { na.x, na.y } - r.o.x, ay = f2 { na.z, na.w } - r.o.y, az = f2 { nb.x, nb.y } - r.o.z;
                     const f2 bx = f2 { nb.z, nb.w } - r.o.x, by = f2 { nc.x, nc.y } - r.o.y, bz = f2 { nc.z, nc.w } - r.o.z;
                     const f2 t1x = ax * r.inv.x, t1y = ay * r.inv.y, t1z = az * r.inv.z;
                     const f2 t2x = bx * r.inv.x, t2y = by * r.inv.y, t2z = bz * r.inv.z;
                     const f2 ddx = { fmaxf(fmaxf(ax.x, -bx.x), 0.0f), fmaxf(fmaxf(ax.y, -bx.y), 0.0f) };
                     const f2 ddy = { fmaxf(fmaxf(ay.x, -by.x), 0.0f), fmaxf(fmaxf(ay.y, -by.y), 0.0f) };
                     const f2 ddz = { fmaxf(fmaxf(az.x, -bz.x), 0.0f), fmaxf(fmaxf(az.y, -bz.y), 0.0f) };
                     const f2 d2 = ddx * ddx + ddy * ddy + ddz * ddz;
                     tl = raw_max3(raw_min(t1x.x, t2x.x), raw_min(t1y.x, t2y.x), raw_min(t1z.x, t2z.x));
                     tr = raw_max3(raw_min(t1x.y, t2x.y), raw_min(t1y.y, t2y.y), raw_min(t1z.y, t2z.y));
                     const float hl = raw_min3(raw_max(t1x.x, t2x.x), raw_max(t1y.x, t2y.x), raw_max(t1z.x, t2z.x));
                     const float hr = raw_min3(raw_max(t1x.y, t2x.y), raw_max(t1y.y, t2y.y), raw_max(t1z.y, t2z.y));
                     lok = nd.x != LINK_NONE && tl <= hl && tl < farp && hl > nearp && !(d2.x > lim2);
                     rok = nd.y != LINK_NONE && tr <= hr && tr < farp && hr > nearp && !(d2.y > lim2);
                     if (MODE == MODE_SHADOW && !st.redo) {
                        const f2 fx = { fmaxf(light.x - na.x, nb.z - light.x), fmaxf(light.x - na.y, nb.w - light.x) };
                        const f2 fy = { fmaxf(light.y - na.z, nc.x - light.y), fmaxf(light.y - na.w, nc.y - light.y) };
                        const f2 fz = { fmaxf(light.z - nb.x, nc.z - light.z), fmaxf(light.z - nb.y, nc.w - light.z) };
                        const f2 far2 = (fx * fx + fy * fy + fz * fz) * 1.0001f + a.light_prune_c;
                        lok = lok && !(d2.x > far2.x);
                        rok = rok && !(d2.y > far2.y);
                     }
                  } else {
                     if (nd.x != LINK_NONE)
                        lok = slab_test(r, nearp, farp, na.x, na.z, nb.x, nb.z, nc.x, nc.z, tl) &&
                              !(origin_box_dist2(r, na.x, na.z, nb.x, nb.z, nc.x, nc.z) > lim2);
                     if (nd.y != LINK_NONE)
                        rok = slab_test(r, nearp, farp, na.y, na.w, nb.y, nb.w, nc.y, nc.w, tr) &&
                              !(origin_box_dist2(r, na.y, na.w, nb.y, nb.w, nc.y, nc.w) > lim2);
                     if (MODE == MODE_SHADOW && !st.redo && a.sc.boxes_finite) {
                        lok = lok && !(origin_box_dist2(r, na.x, na.z, nb.x, nb.z, nc.x, nc.z) > light_far2(light, na.x, na.z, nb.x, nb.z, nc.x, nc.z) * 1.0001f + a.light_prune_c);
                        rok = rok && !(origin_box_dist2(r, na.y, na.w, nb.y, nb.w, nc.y, nc.w) > light_far2(light, na.y, na.w, nb.y, nb.w, nc.y, nc.w) * 1.0001f + a.light_prune_c);
                     }
                  }
                  const int le = nd.x ^ nd.z;   // (the record's "is a leaf" masks: a leaf is ~descriptor)
                  const int re = nd.y ^ nd.w;
                  // what is left to test in a leaf child is known here (the leaf side's group and `missed` only ever grow
                  // while the group stays: a face in `missed` HAS missed this ray, whenever that is asked)
                  const int lg = __float_as_int(q4.x), rg = __float_as_int(q4.z);
                  const unsigned ltodo = __float_as_uint(q4.y) & ~((((lg ^ gb) & DESC_BASE_MASK) == 0) ? missed : 0u);
                  const unsigned rtodo = __float_as_uint(q4.w) & ~((((rg ^ gb) & DESC_BASE_MASK) == 0) ? missed : 0u);
                  if (nd.z != 0 && ltodo == 0u && !(lg & DESC_MORE)) {
                     if (STATS > 1 && lok) d_repeat += (unsigned) __popc(__float_as_uint(q4.y));
                     lok = false;
                  }
                  if (nd.w != 0 && rtodo == 0u && !(rg & DESC_MORE)) {
                     if (STATS > 1 && rok) d_repeat += (unsigned) __popc(__float_as_uint(q4.w));
                     rok = false;
                  }
                  // A leaf child goes to the leaf side if that is free (with its words: no descriptor read); the node side
                  // goes on with the other child or the stack.  A leaf child that finds the leaf side busy waits in `cur`.
#if WALK_FLAT_CONTROL
                  // (the same decisions as the branch form below, taken on the children in visiting order -- A first, B second --
                  // so that every choice is one select deep)
                  const bool lfree = lf == 0;
                  const bool both = lok && rok, any = lok || rok;
                  const bool first_right = rok && (!lok || tr < tl);   // nearest child first; the order never decides WHETHER a child is visited
                  const int eA = first_right ? re : le, eB = first_right ? le : re;
                  const int gA = first_right ? rg : lg, gB = first_right ? lg : rg;
                  const unsigned mA = first_right ? __float_as_uint(q4.w) : __float_as_uint(q4.y), mB = first_right ? __float_as_uint(q4.y) : __float_as_uint(q4.w);
                  const unsigned tA = first_right ? rtodo : ltodo, tB = first_right ? ltodo : rtodo;
                  const bool hand_a = any && lfree && eA < 0;              // A goes to the leaf side ...
                  const bool hand_b = both && lfree && !hand_a && eB < 0;  // ... or B does, and the node side goes on with A
                  const bool need_pop = !any || (hand_a && !both);
                  if (both && !hand_a && !hand_b) {
                     stack[sp * BLOCK_THREADS] = eB;
                     ++sp;
                  }
                  int next = hand_a ? eB : eA;
                  if (need_pop) {
                     next = STACK_EMPTY;
                     if (sp > bot) {
                        --sp;
                        next = stack[sp * BLOCK_THREADS];
                     }
                  }
                  cur = next;
                  wmask = 0u;
                  if (!need_pop && cur < 0) {   // a leaf child that waits in `cur` keeps its words
                     wg = hand_a ? gB : gA;
                     wmask = hand_a ? mB : mA;
                  }
                  if (hand_a || hand_b) {
                     const int g = hand_a ? gA : gB;
                     lf = hand_a ? eA : eB;
                     if (((g ^ gb) & DESC_BASE_MASK) != 0) missed = 0u;   // another group
                     if (STATS > 1) d_repeat += (unsigned) __popc((hand_a ? mA : mB) & missed);
                     gb = g;
                     todo = hand_a ? tA : tB;   // 0 only for a leaf that goes on in further descriptors: the leaf step moves on
                  }
#else
                  const bool lfree = lf == 0;
                  int hand = 0;   // 1 / 2: the left / right child goes to the leaf side
                  int kept = 0;   // 1 / 2: the left / right child is what `cur` holds now
                  if (lok && rok) {
                     const bool swap = tr < tl;   // nearest child first; the order never decides WHETHER a child is visited
                     const int n1 = swap ? re : le, n2 = swap ? le : re;
                     if (lfree && n1 < 0) {
                        hand = swap ? 2 : 1;
                        cur = n2;
                        kept = swap ? 1 : 2;
                     } else if (lfree && n2 < 0) {
                        hand = swap ? 1 : 2;
                        cur = n1;
                        kept = swap ? 2 : 1;
                     } else {
                        cur = n1;
                        kept = swap ? 2 : 1;
                        stack[sp * BLOCK_THREADS] = n2;
                        ++sp;
                     }
                  } else if (lok || rok) {
                     const int n1 = lok ? le : re;
                     if (lfree && n1 < 0) {
                        hand = lok ? 1 : 2;
                        if (sp > bot) {
                           --sp;
                           cur = stack[sp * BLOCK_THREADS];
                        } else {
                           cur = STACK_EMPTY;
                        }
                     } else {
                        cur = n1;
                        kept = lok ? 1 : 2;
                     }
                  } else if (sp > bot) {
                     --sp;
                     cur = stack[sp * BLOCK_THREADS];
                  } else {
                     cur = STACK_EMPTY;
                  }
                  wmask = 0u;
                  if (kept != 0 && cur < 0) {
                     wg = kept == 1 ? lg : rg;
                     wmask = kept == 1 ? __float_as_uint(q4.y) : __float_as_uint(q4.w);
                  }
                  if (hand != 0) {
                     const int g = hand == 1 ? lg : rg;
                     lf = hand == 1 ? le : re;
                     if (((g ^ gb) & DESC_BASE_MASK) != 0) missed = 0u;   // another group
                     if (STATS > 1) d_repeat += (unsigned) __popc((hand == 1 ? __float_as_uint(q4.y) : __float_as_uint(q4.w)) & missed);
                     gb = g;
                     todo = hand == 1 ? ltodo : rtodo;   // 0 only for a leaf that goes on in further descriptors: the leaf step moves on
                  }
#endif
               }
            }
         }
         // the leaf side, with the leaves the node step has just handed over
         bool want_resolve = false;
         const bool in_leaf = lf < 0;
         const int n_leaf = __popcll(__ballot(in_leaf));
         const bool run_leaf = n_leaf > 0 && (exhausted || n_leaf >= SIDE_MIN || !ran_node);
         if (run_leaf) {
            const bool want_desc = in_leaf && todo == 0u;
            const unsigned rest0 = todo & (todo - 1u);
            const size_t f0 = in_leaf && !want_desc ? (size_t) (gb & DESC_BASE_MASK) + (size_t) (__ffs((int) todo) - 1) : 0;
            const size_t f1 = in_leaf && rest0 != 0u ? (size_t) (gb & DESC_BASE_MASK) + (size_t) (__ffs((int) rest0) - 1) : f0;
            const float4 *pa = want_desc ? descs + (size_t) ~lf : gfaces + 3 * f0;
            const float4 *pb = gfaces + 3 * f1;
            float4 q[6];
#pragma unroll
            for (int j = 0; j < 3; j++) q[j] = pa[j];   // (a descriptor is one float4: the array is padded for the other two)
#pragma unroll
            for (int j = 0; j < 3; j++) q[3 + j] = pb[j];
            if (in_leaf) {
               if (STATS > 1) {
                  d_leaf_lane++;
                  if (first_active_lane()) d_leaf_wave++;
               }
               if (todo == 0u) {
                  const int g = __float_as_int(q[0].y);
                  const unsigned mask = __float_as_uint(q[0].z);
                  if (((g ^ gb) & DESC_BASE_MASK) != 0) missed = 0u;   // another group (gb == -1: none yet)
                  gb = g;
                  todo = mask & ~missed;
                  if (STATS > 1) d_repeat += (unsigned) __popc(mask & missed);
               } else {
                  const unsigned bit0 = todo & (0u - todo), rest = todo & (todo - 1u), bit1 = rest & (0u - rest);
                  const int gbase = gb & DESC_BASE_MASK;
                  unsigned done = bit0 | bit1;
                  if (CLOSEST) {
                     if (STATS) (FUSED && st.shadow ? n_tri_s : n_tri) += bit1 ? 2u : 1u;
                     float u0 = 0.0f, v0 = 0.0f, u1 = 0.0f, v1 = 0.0f;
                     const bool m0 = tri_meets_uv(q[0], q[1], q[2], r, eps, u0, v0);
                     const bool m1 = bit1 != 0u && tri_meets_uv(q[3], q[4], q[5], r, eps, u1, v1);
                     missed |= (m0 ? 0u : bit0) | (m1 ? 0u : bit1);
                     if (m0 || m1) {
                        if (pend < 0) {
                           pend = (gbase + __ffs((int) (m0 ? bit0 : bit1)) - 1) | (gb & DESC_IDENT);
                           s_pkey[threadIdx.x] = ~lf;
                           s_puv[threadIdx.x] = m0 ? u0 : u1;   // the batched pass goes on from here
                           s_puv[BLOCK_THREADS + threadIdx.x] = m0 ? v0 : v1;
                           if (m0 && m1) done = bit0;   // the second one is met again next trip, when the first has been worked out
                        } else {
                           done &= ~((m0 ? bit0 : 0u) | (m1 ? bit1 : 0u));   // one waits already: it is worked out first
                        }
                     }
                     want_resolve = done != (bit0 | bit1) || (FUSED && st.shadow && pend >= 0);
                  } else {
#pragma unroll
                     for (int j = 0; j < LEAF_BATCH; j++) {
                        const unsigned bit = j ? bit1 : bit0;
                        if (bit != 0u) {
                           if (STATS) n_tri++;
                           float u, v, dist;
                           V3 P;
                           const bool met = tri_test(q[3 * j], q[3 * j + 1], q[3 * j + 2], r, eps, u, v, P, dist);
                           if (!met) missed |= bit;
                           const int gfx = (gbase + __ffs((int) bit) - 1) | (gb & DESC_IDENT);
                           const int key = met ? __float_as_int(descs[(size_t) ~lf].x) : 0;
                           if (met && (dist > a.sc.objs[st.k].shadow_distance) && beats(dist, key, gfx)) {
                              ob_w = dist;
                              ob_gf = gfx;
                              ob_key = key;
                              lim2 = fminf(lim2, prune_limit2(r, dist, abs_max));
                              if (MODE == MODE_SHADOW_ANY || (MODE == MODE_SHADOW && dist < safe_w)) {   // device_util.h, shortcut 1
                                 done = todo;          // nothing else in this leaf ...
                                 gb &= ~DESC_MORE;     // ... or the rest of it ...
                                 sp = bot;             // ... or anywhere else
                                 cur = STACK_EMPTY;
                                 break;
                              }
                           }
                        }
                     }
                  }
                  todo &= ~done;
               }
               if (todo == 0u) lf = (gb & DESC_MORE) ? lf - 1 : 0;   // this descriptor is finished: the leaf's next one, or the side is free
            }
         }
         if (DEFER) resolve_req = __ballot(want_resolve) != 0ull;
         continue;
      }

      // ---- the burst form (BIG): one cursor per lane -- an internal node (C) or a leaf whose chunks the wave tests (D').
      // One fetch per lane per trip, issued up front: a node lane reads its 80-byte record, a leaf lane its 16-byte
      // descriptor (a leaf that came off the stack; one entered from its parent has the words already); a leaf's faces
      // are fetched by the lanes that test them, in the burst below.
      if (STATS > 1 && st.role == ROLE_OWNER) d_ray_trips++;
      if (!CLOSEST) ++age;
      const bool is_node = (unsigned) cur < (unsigned) STACK_EMPTY;
      const bool is_leaf = cur < 0;
      const int n_node = __popcll(__ballot(is_node));
      constexpr int NQ = 5;
      float4 q[NQ];
      {
         const bool want_desc = is_leaf && todo == 0u;
         const float4 *pa = is_node ? nodes + NODE_F4 * (size_t) cur : (want_desc ? descs + (size_t) ~cur : gfaces);
         const float4 *pb = is_node ? pa + 3 : gfaces;
#pragma unroll
         for (int j = 0; j < 3; j++) q[j] = pa[j];   // (a descriptor is one float4: the array is padded for the other two)
         q[3] = pb[0];
         q[4] = pb[1];   // a node record's fifth float4: its leaf children's descriptor words
      }
      const float4 na = q[0], nb = q[1], nc = q[2];
      const int4 nd = make_int4(__float_as_int(q[3].x), __float_as_int(q[3].y), __float_as_int(q[3].z), __float_as_int(q[3].w));

      if (is_node) {
         // (C) both children of one internal node
         if (STATS > 1) {
            d_node_lane++;
            if (first_active_lane()) d_node_wave++;
         }
         float tl = 0.0f, tr = 0.0f;
         bool lok = false, rok = false;
         if (STATS) (FUSED && st.shadow ? n_box_s : n_box) += (nd.x != LINK_NONE ? 1u : 0u) + (nd.y != LINK_NONE ? 1u : 0u);
         if (st.plain) {
            // Both children at once in packed f32 (the record interleaves them).  With every operand finite no NaN can
            // arise, and RTStatics.aabbIntersection (RTStatics.java:102-158) reduces to its textbook form: per axis
            // the reference's sign-selected (tmin, tmax) IS (min, max) of the two products (rounding is monotone and
            // lo <= hi), its two early-outs and merges are "the three intervals intersect", and what it compares
            // with FAR / NEAR are max3 of the tmins and min3 of the tmaxs -- the same values, the same answer.
            const f2 ax = f2 { na.x, na.y } - r.o.x, ay = f2 { na.z, na.w } - r.o.y, az = f2 { nb.x, nb.y } - r.o.z;
            const f2 bx = f2 { nb.z, nb.w } - r.o.x, by = f2 { nc.x, nc.y } - r.o.y, bz = f2 { nc.z, nc.w } - r.o.z;
            const f2 t1x = ax * r.inv.x, t1y = ay * r.inv.y, t1z = az * r.inv.z;
            const f2 t2x = bx * r.inv.x, t2y = by * r.inv.y, t2z = bz * r.inv.z;
            // squared distance from the origin to each box (the prune test), from the same differences
            const f2 ddx = { fmaxf(fmaxf(ax.x, -bx.x), 0.0f), fmaxf(fmaxf(ax.y, -bx.y), 0.0f) };
            const f2 ddy = { fmaxf(fmaxf(ay.x, -by.x), 0.0f), fmaxf(fmaxf(ay.y, -by.y), 0.0f) };
            const f2 ddz = { fmaxf(fmaxf(az.x, -bz.x), 0.0f), fmaxf(fmaxf(az.y, -bz.y), 0.0f) };
            const f2 d2 = ddx * ddx + ddy * ddy + ddz * ddz;
            tl = raw_max3(raw_min(t1x.x, t2x.x), raw_min(t1y.x, t2y.x), raw_min(t1z.x, t2z.x));
            tr = raw_max3(raw_min(t1x.y, t2x.y), raw_min(t1y.y, t2y.y), raw_min(t1z.y, t2z.y));
            const float hl = raw_min3(raw_max(t1x.x, t2x.x), raw_max(t1y.x, t2y.x), raw_max(t1z.x, t2z.x));
            const float hr = raw_min3(raw_max(t1x.y, t2x.y), raw_max(t1y.y, t2y.y), raw_max(t1z.y, t2z.y));
            lok = nd.x != LINK_NONE && tl <= hl && tl < farp && hl > nearp && !(d2.x > lim2);
            rok = nd.y != LINK_NONE && tr <= hr && tr < farp && hr > nearp && !(d2.y > lim2);
            if (MODE == MODE_SHADOW && !st.redo) {   // hopeless boxes (device_util.h, shortcut 2); st.plain implies finite boxes
               const f2 fx = { fmaxf(light.x - na.x, nb.z - light.x), fmaxf(light.x - na.y, nb.w - light.x) };
               const f2 fy = { fmaxf(light.y - na.z, nc.x - light.y), fmaxf(light.y - na.w, nc.y - light.y) };
               const f2 fz = { fmaxf(light.z - nb.x, nc.z - light.z), fmaxf(light.z - nb.y, nc.w - light.z) };
               const f2 far2 = (fx * fx + fy * fy + fz * fz) * 1.0001f + a.light_prune_c;
               lok = lok && !(d2.x > far2.x);
               rok = rok && !(d2.y > far2.y);
            }
         } else {
            // a ray with a zero direction component (1/d infinite) or non-finite data: the reference's own branch
            // structure, so that NaNs fall where they fall there
            if (nd.x != LINK_NONE)
               lok = slab_test(r, nearp, farp, na.x, na.z, nb.x, nb.z, nc.x, nc.z, tl) &&
                     !(origin_box_dist2(r, na.x, na.z, nb.x, nb.z, nc.x, nc.z) > lim2);
            if (nd.y != LINK_NONE)
               rok = slab_test(r, nearp, farp, na.y, na.w, nb.y, nb.w, nc.y, nc.w, tr) &&
                     !(origin_box_dist2(r, na.y, na.w, nb.y, nb.w, nc.y, nc.w) > lim2);
            if (MODE == MODE_SHADOW && !st.redo && a.sc.boxes_finite) {
               lok = lok && !(origin_box_dist2(r, na.x, na.z, nb.x, nb.z, nc.x, nc.z) > light_far2(light, na.x, na.z, nb.x, nb.z, nc.x, nc.z) * 1.0001f + a.light_prune_c);
               rok = rok && !(origin_box_dist2(r, na.y, na.w, nb.y, nb.w, nc.y, nc.w) > light_far2(light, na.y, na.w, nb.y, nb.w, nc.y, nc.w) * 1.0001f + a.light_prune_c);
            }
         }
         const int le = nd.x ^ nd.z;   // stack / cursor form of each child: leaf = ~(its first descriptor); the record's masks
         const int re = nd.y ^ nd.w;
         // A leaf child's descriptor words ride in this record: what is left to test in it is known here.  A leaf all of
         // whose faces this ray has already missed (in neighbouring leaves of its group) is no visit at all.
         const int lg = __float_as_int(q[4].x), rg = __float_as_int(q[4].z);
         const unsigned ltodo = __float_as_uint(q[4].y) & ~((((lg ^ gb) & DESC_BASE_MASK) == 0) ? missed : 0u);
         const unsigned rtodo = __float_as_uint(q[4].w) & ~((((rg ^ gb) & DESC_BASE_MASK) == 0) ? missed : 0u);
         if (nd.z != 0 && ltodo == 0u && !(lg & DESC_MORE)) {
            if (STATS > 1 && lok) d_repeat += (unsigned) __popc(__float_as_uint(q[4].y));
            lok = false;
         }
         if (nd.w != 0 && rtodo == 0u && !(rg & DESC_MORE)) {
            if (STATS > 1 && rok) d_repeat += (unsigned) __popc(__float_as_uint(q[4].w));
            rok = false;
         }
         int took = 0;   // 1 / 2: stepped into the left / right child; 0: back to the stack
         if (lok && rok) {
            // nearest child first; the order never decides WHETHER a child is visited
            const bool swap = tr < tl;
            cur = swap ? re : le;
            took = swap ? 2 : 1;
            stack[sp * BLOCK_THREADS] = swap ? le : re;
            ++sp;
         } else if (lok) {
            cur = le;
            took = 1;
         } else if (rok) {
            cur = re;
            took = 2;
         } else if (sp > bot) {
            --sp;
            cur = stack[sp * BLOCK_THREADS];
         } else {
            cur = STACK_EMPTY;
         }
         todo = 0u;   // a leaf off the stack reads its descriptor first
         if (took != 0 && cur < 0) {
            const int g = took == 1 ? lg : rg;
            if (((g ^ gb) & DESC_BASE_MASK) != 0) missed = 0u;   // another group
            if (STATS > 1) d_repeat += (unsigned) __popc((took == 1 ? __float_as_uint(q[4].y) : __float_as_uint(q[4].w)) & missed);
            gb = g;
            todo = took == 1 ? ltodo : rtodo;   // 0 only for a leaf that goes on in further descriptors: the leaf step moves on
         }
      }

      bool want_resolve = false;
      {
         // (D') leaves of tens of faces (the reference's depth limit of 20 on a mesh of millions: layout.h chunks of up to
         // 32 faces).  Two faces per lane and trip would leave a leaf lane sixteen trips in one chunk with the node lanes
         // masked off beside it.  Instead the WAVE tests chunks: each pass takes the next two leaf lanes, broadcasts their
         // rays, and lane j of each half tests face j of that lane's chunk -- 64 triangle tests per pass at full lane use.
         // The chunk's owner gets the mask of faces that passed; the first waits in `pend` (resolved at the top of the
         // next trip), any others stay in `todo` and are met again.  Which lane tests a face cannot matter: the tests
         // read only the ray and the face, and the fold is order independent.
         if (is_leaf && todo == 0u) {
            const int g = __float_as_int(q[0].y);
            const unsigned mask = __float_as_uint(q[0].z);
            if (((g ^ gb) & DESC_BASE_MASK) != 0) missed = 0u;   // another group (gb == -1: none yet)
            gb = g;
            todo = mask & ~missed;
         }
         // A burst is a hundred-odd instructions per pass for the whole wave, and a node step a few dozen: the lanes at
         // nodes go on walking (cheap trips) until enough lanes wait with a chunk, or few lanes are at nodes.
         unsigned long long el = __ballot(is_leaf && todo != 0u);
         if ((int) __popcll(el) < BURST_MIN && n_node >= BURST_NODE_FEW && !exhausted) el = 0ull;
         // A pass packs the chunks of the next few leaf lanes side by side over the 64 lanes: a chunk of n faces (bits
         // 0 .. n-1 of `todo`; n = its highest face + 1) takes n lanes, and lane `base + j` tests face j.  The passes are
         // software-pipelined: the faces of the next pass are on their way while this one is tested.
         struct Pass {
            int src, base;      // per lane: the leaf lane whose chunk this lane tests a face of (-1: none), first lane of that chunk
            int my_base;        // per leaf lane: first lane of its chunk in this pass (-1: not in this pass)
            bool have;
            float4 fA, fB, fC;
         };
         auto next_pass = [&](Pass &ps) {
            int off = 0;
            ps.src = -1;
            ps.base = 0;
            ps.my_base = -1;
            while (el != 0ull) {
               const int sl = __ffsll((unsigned long long) el) - 1;
               const int len = 32 - __clz(__builtin_amdgcn_readlane((int) todo, sl));
               if (off + len > 64) break;
               el &= el - 1ull;
               if (lane >= off) {
                  ps.src = sl;
                  ps.base = off;
               }
               if (lane == sl) ps.my_base = off;
               off += len;
            }
            if (lane >= off) ps.src = -1;
            const int from = ps.src >= 0 ? ps.src : lane;
            const unsigned s_todo = (unsigned) __shfl((int) todo, from, 64);
            const int s_gb = __shfl(gb, from, 64);
            const int j = lane - ps.base;
            ps.have = ps.src >= 0 && ((s_todo >> j) & 1u) != 0u;
            const float4 *t = gfaces + 3 * (ps.have ? (size_t) (s_gb & DESC_BASE_MASK) + (size_t) j : (size_t) 0);
            ps.fA = t[0];
            ps.fB = t[1];
            ps.fC = t[2];
         };
         unsigned res = 0u;     // per leaf lane: the faces of its chunk that passed
         bool served = false;   // its chunk was in one of this trip's passes
         auto test_pass = [&](const Pass &pc) {
            const int from = pc.src >= 0 ? pc.src : lane;
            Ray sr;
            sr.o = mk3(__shfl(r.o.x, from, 64), __shfl(r.o.y, from, 64), __shfl(r.o.z, from, 64));
            sr.d = mk3(__shfl(r.d.x, from, 64), __shfl(r.d.y, from, 64), __shfl(r.d.z, from, 64));
            sr.inv = sr.d;   // not used by the triangle tests
            const bool met = pc.have && tri_meets(pc.fA, pc.fB, pc.fC, sr, eps);
            const unsigned long long bal = __ballot(met);
            if (pc.my_base >= 0) {   // (the lanes above this chunk's belong to the next one; what follows from the answer waits for the end of the burst)
               res = (unsigned) (bal >> pc.my_base) & todo;
               served = true;
            }
         };
         if (el != 0ull) {
            Pass p0, p1;   // ping-pong: one is tested while the other's faces arrive
            next_pass(p0);
            for (;;) {
               bool more = el != 0ull;
               if (more) next_pass(p1);
               test_pass(p0);
               if (!more) break;
               more = el != 0ull;
               if (more) next_pass(p0);
               test_pass(p1);
               if (!more) break;
            }
         }
         if (served) {
            const unsigned passed = res;
            if (STATS) (FUSED && st.shadow ? n_tri_s : n_tri) += (unsigned) __popc(todo);
            missed |= todo & ~passed;
            if (passed != 0u) {
               if (pend < 0) {
                  const unsigned first = passed & (0u - passed);
                  pend = ((gb & DESC_BASE_MASK) + __ffs((int) first) - 1) | (gb & DESC_IDENT);
                  s_pkey[threadIdx.x] = ~cur;
                  todo = passed & ~first;
               } else {
                  todo = passed;   // one waits already: it is worked out first
               }
               want_resolve = todo != 0u || (FUSED && st.shadow);   // faces that could not be noted, or a shadow ray's first face
            } else {
               todo = 0u;
            }
         }
         if (is_leaf && todo == 0u) {   // this chunk is finished: the leaf's next one, or back to the stack
            if (gb & DESC_MORE) {
               cur -= 1;
            } else if (sp > bot) {
               --sp;
               cur = stack[sp * BLOCK_THREADS];
            } else {
               cur = STACK_EMPTY;
            }
         }
      }
      if (DEFER) resolve_req = __ballot(want_resolve) != 0ull;
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
      if (FUSED) {   // the shadow rays' share is booked as the shadow launch's would be
         add_stat(a.counters, ST_KIND_RAYS + 2, n_spawn);
         add_stat(a.counters, ST_KIND_BOX + 2, n_box_s);
         add_stat(a.counters, ST_KIND_TRI + 2, n_tri_s);
      }
      if (STATS > 1) {
         add_stat(a.counters, ST_NODE_LANE, d_node_lane);
         add_stat(a.counters, ST_NODE_WAVE, d_node_wave);
         add_stat(a.counters, ST_LEAF_LANE, d_leaf_lane);
         add_stat(a.counters, ST_LEAF_WAVE, d_leaf_wave);
         add_stat(a.counters, ST_OUTER_WAVE, d_outer);
         add_stat(a.counters, ST_REFILL, d_refill);
         add_stat(a.counters, 12, d_steal);
         add_stat(a.counters, 63, d_repeat);
         if (kind != 1) {
   #pragma unroll
            for (int q = 0; q < 8; q++) add_stat(a.counters, ST_TRIP_HIST + (kind ? 8 : 0) + q, d_hist[q]);
         }
         {
            unsigned m = max(d_ray_trips_max, d_ray_trips);
            for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned) __shfl_down((int) m, off, 64));
            const unsigned long long any = __ballot(n_ray > 0);
            if (lane == 0) {
               unsigned long long *tl = (unsigned long long *) (a.counters + CNT_STATS) + ST_KIND_TIMELINE + 6 * kind;
               atomicMax(tl + 4, (unsigned long long) m);
               if (any) atomicAdd(tl + 5, 1ull);
            }
         }
      }
      if (lane == 0) {
         const unsigned long long t_end = wall_clock64();
         unsigned long long *st = (unsigned long long *) (a.counters + CNT_STATS);
         atomicMax(st + 13, t_end - t_begin);                               // longest wave lifetime
         atomicAdd(st + 14, t_end - t_begin);                               // sum of wave lifetimes
         atomicAdd(st + 15, t_exhausted ? t_end - t_exhausted : 0ull);      // sum of time spent after the queue ran dry
         unsigned long long *tl = st + ST_KIND_TIMELINE + 6 * kind;
         atomicMax(tl + 0, ~t_begin);
         if (t_exhausted) {
            atomicMax(tl + 1, ~t_exhausted);
            atomicMax(tl + 2, t_exhausted);
         }
         atomicMax(tl + 3, t_end);
         atomicMax(st + ST_TAIL_TRIPS + 2 * (kind ? 1 : 0), (unsigned long long) d_tail_trips);
         atomicAdd(st + ST_TAIL_TRIPS + 2 * (kind ? 1 : 0) + 1, (unsigned long long) d_tail_trips);
      }
   }
}

// ---------------------------------------------------------------- shade hits
// LEVEL 0: primary queue (Q0).  LEVEL 1: a secondary (mirror bounce) queue; aux >> 16 = bounces so
// far, slot = index of the path's primary queue entry.  PLAIN: every material of the scene is a ColorInformation (the
// table search, bounce rays and the checkerboard are compiled out: 161 -> far fewer registers, no list in LDS).
#ifndef SHADE_WAVES
#define SHADE_WAVES 4   // 128 VGPRs (20 bytes of scratch): the table search is latency-bound, Buddha-class shade 0.74 -> 0.69 ms
#endif
template <int LEVEL, int PLAIN>
__global__ __launch_bounds__(BLOCK_THREADS) __attribute__((amdgpu_waves_per_eu(PLAIN ? 5 : SHADE_WAVES, PLAIN ? 5 : SHADE_WAVES))) void shade_kernel(const FrameArgs p, const float *rays, const float *hits,
                                                              const int32_t *hit_obj, int count_index, float *rays_next,
                                                              int next_count_index)
{
   __shared__ unsigned s_cnt[BLOCK_THREADS / 64 + 1];
   unsigned n_sec = 0;
   {
   const int band = (int) (blockIdx.x % (unsigned) N_BANDS);   // a workgroup serves one band (grid is a multiple of N_BANDS)
   const unsigned count = p.counters[count_index + band];
   for (unsigned base = (blockIdx.x / N_BANDS) * BLOCK_THREADS; base < count; base += (gridDim.x / N_BANDS) * BLOCK_THREADS) {
   const bool active = base + threadIdx.x < count;
   const unsigned i = (unsigned) band * p.band_cap + base + threadIdx.x;
   bool push_shadow = false, push_sec = false;
   unsigned path = 0;
   V3 sec_o = mk3(0, 0, 0), sec_d = mk3(0, 0, -1);
   int sec_aux = 0;
   if (active) {
      const float4 *q = (const float4 *) rays + 2 * (size_t) i;
      const float4 r0 = q[0], r1 = q[1];
      const float4 *hq = (const float4 *) hits + 2 * (size_t) i;
      const float4 h0 = hq[0], h1 = hq[1];
      const int obj = hit_obj[i];
      const unsigned slot_or_path = __float_as_uint(r0.w);
      const int aux = __float_as_int(r1.w);
      // aux of a bounce ray: bits 0-15 the object it left (skipped by the walk), 16-26 bounces so far (the reference's
      // `depth`), 27-30 RefractionMaterial terms gathered so far, 31 set when the surface it left owes one more
      const int bounces = LEVEL == 0 ? 0 : ((aux >> 16) & 0x7ff);
      int n_terms = LEVEL == 0 ? 0 : ((aux >> 27) & 15);
      path = LEVEL == 0 ? i : slot_or_path;
      // the sample slot and pixel of this path
      const unsigned slot = LEVEL == 0 ? slot_or_path : __float_as_uint(((const float4 *) p.rays0)[2 * (size_t) path].w);
      float mat[3];
      bool have_mat = false;
      if (obj < 0) {
         if (LEVEL == 0) {
            // RenderThread.java:53-55
            ((float4 *) p.samples)[slot] = make_float4(p.light_ambient[0], p.light_ambient[1], p.light_ambient[2], 0.0f);
         } else {
            // ReflectionMaterial.java:19-20, RefractionMaterial.java:39-40
            mat[0] = p.light_ambient[0];
            mat[1] = p.light_ambient[1];
            mat[2] = p.light_ambient[2];
            have_mat = true;
         }
      } else {
         const DevObject &ob = p.sc.objs[obj];
         const DevMaterial &m = ob.mat;
         const V3 P = mk3(h0.x, h0.y, h0.z), N = mk3(h1.x, h1.y, h1.z), D = mk3(r1.x, r1.y, r1.z);
         if (LEVEL == 1 && aux < 0) {
            // the refracting surface this ray left was entered from outside and the ray hit something:
            // RefractionMaterial.java:43-53's term, kept as its exponent until the path ends
            const float4 pp = ((const float4 *) p.refr_parent)[path];
            const float *pd = p.sc.objs[aux & 0xffff].mat.diffuse;
            const float distance = distance3(mk3(pp.x, pp.y, pp.z), P);
            const float f = 0.15f * -distance;
            ((float4 *) p.refr_terms)[(size_t) path * MAX_REFRACTION_TERMS + n_terms] = make_float4(pd[0] * f, pd[1] * f, pd[2] * f, 0.0f);
            n_terms++;
         }
         const bool cut = ob.kind != OBJ_MESH && bounces >= p.max_recursion_depth;   // SceneObject.java:65-67; KDTree.java:77-80 has none
         if (PLAIN) {
            mat[0] = m.diffuse[0];   // ColorInformation.java:35-37
            mat[1] = m.diffuse[1];
            mat[2] = m.diffuse[2];
            have_mat = true;
         } else if (!cut && m.kind == MAT_REFLECTION && bounces < p.max_mirror_bounces) {
            // ReflectionMaterial.java:16-17 + Camera.java:196-200 (offset origin, Ray ctor normalises again)
            sec_o = add3(P, scale3(N, p.sc.epsilon));
            sec_d = normalize3(reflect3(N, D));
            sec_aux = (obj & 0xffff) | ((bounces + 1) << 16) | (n_terms << 27);
            push_sec = true;
            n_sec++;
         } else if (!cut && m.kind == MAT_REFRACTION && bounces < p.max_mirror_bounces && n_terms < MAX_REFRACTION_TERMS) {
            // RefractionMaterial.java:33-35 + Camera.java:196-200
            const bool entering = dot3(N, D) <= 0.0f;   // RefractionMaterial.java:43
            sec_o = add3(P, scale3(N, p.sc.epsilon));
            sec_d = normalize3(refraction_direction(N, D, p.sc.ext[obj].refraction_index));
            sec_aux = (obj & 0xffff) | ((bounces + 1) << 16) | (n_terms << 27) | (entering ? (int) 0x80000000 : 0);
            if (entering) ((float4 *) p.refr_parent)[path] = make_float4(P.x, P.y, P.z, 0.0f);
            push_sec = true;
            n_sec++;
         } else if (!cut && m.kind == MAT_BRDF) {
            // BRDFMaterial.getMaterialColor materials/BRDFMaterial.java:158-167
            const V3 light = mk3(p.light_origin[0], p.light_origin[1], p.light_origin[2]);
            const float lum = brdf_luminocity(p.brdf_dirs, p.brdf_weights + (size_t) m.brdf_id * (size_t) p.n_dirs, p.n_dirs,
                                              p.brdf_samples, D, P, N, light);
            float hsv[3];
            rgb_to_hsv(m.diffuse, hsv);
            hsv[2] = lum;
            hsv_to_rgb(hsv, mat);
            have_mat = true;
         } else if (!cut && m.kind == MAT_CHECKERBOARD) {
            checker_color(ob.bmin, ob.bmax, P, p.sc.ext[obj], mat);
            have_mat = true;
         } else {
            // ColorInformation.java:35-37; the cut-off and the bounce guards answer with getDiffuse() too
            mat[0] = m.diffuse[0];
            mat[1] = m.diffuse[1];
            mat[2] = m.diffuse[2];
            have_mat = true;
         }
      }
      if (LEVEL == 1 && have_mat && n_terms > 0) {
         // the path has ended: RefractionMaterial.java:50-52's additions unwind deepest first
         for (int t = n_terms - 1; t >= 0; t--) {
            const float4 x = ((const float4 *) p.refr_terms)[(size_t) path * MAX_REFRACTION_TERMS + t];
            mat[0] = (float) ((double) mat[0] + exp((double) x.x));
            mat[1] = (float) ((double) mat[1] + exp((double) x.y));
            mat[2] = (float) ((double) mat[2] + exp((double) x.z));
         }
      }
      if (LEVEL == 0) {
         const unsigned wave = slot >> 6;
         int unit, j;
         fast_divmod(wave, p.div_ms, unit, j);
         if (j == 0 && (p.hit_id || p.hit_obj || p.hit_w)) {
            int px, py;
            long long pix;
            (void) micro_pixel(p, p.micro_begin + unit, (int) (slot & 63u), px, py, pix);
            if (obj < 0) {
               write_miss_outputs(p, pix);
            } else {
               if (p.hit_id) p.hit_id[pix] = __float_as_int(h1.w);
               if (p.hit_obj) p.hit_obj[pix] = obj;
               if (p.hit_w) p.hit_w[pix] = h0.w;
            }
         }
      }
      if (have_mat) {
         if (LEVEL == 0 && p.phong_in_shade) {
            // the shadow ray's answer is known already (MODE_FUSED) and the material colour is final: PhongLightingModel
            // right here, with the values phong_kernel would read back
            const DevMaterial &m0 = p.sc.objs[obj].mat;
            float out[3];
            phong_combine(mat, m0.diffuse, m0.specular, m0.shininess, p.light_emission, p.light_ambient,
                          mk3(p.light_origin[0], p.light_origin[1], p.light_origin[2]), mk3(h0.x, h0.y, h0.z), mk3(h1.x, h1.y, h1.z),
                          mk3(r1.x, r1.y, r1.z), p.mat0[4 * (size_t) path + 3] != 0.0f, out);
            ((float4 *) p.samples)[slot] = make_float4(out[0], out[1], out[2], 0.0f);
         } else if (p.lighting_model == 1) {
            if (p.fused_shadow) {   // .w holds the shadow ray's answer already (MODE_FUSED)
               float *m3 = p.mat0 + 4 * (size_t) path;
               m3[0] = mat[0];
               m3[1] = mat[1];
               m3[2] = mat[2];
            } else {
               ((float4 *) p.mat0)[path] = make_float4(mat[0], mat[1], mat[2], 0.0f);
            }
            push_shadow = true;
         } else {
            ((float4 *) p.samples)[slot] = make_float4(mat[0], mat[1], mat[2], 0.0f);   // LightingModel.java:13-15
         }
      }
   }
   // a path's shadow ray and bounce ray stay in the path's band
   const unsigned si = (unsigned) band * p.band_cap + block_queue_push<BLOCK_THREADS / 64>(p.counters + CNT_SHADOW + band, push_shadow, s_cnt);
   if (push_shadow) p.shadow[si] = path;
   if (!PLAIN && rays_next) {
      const unsigned qi = (unsigned) band * p.band_cap + block_queue_push<BLOCK_THREADS / 64>(p.counters + next_count_index + band, push_sec, s_cnt);
      if (push_sec) store_ray(rays_next, qi, sec_o, sec_d, path, sec_aux);
   }
   }
   }
   if (p.stats) add_stat(p.counters, ST_SECONDARY, n_sec);
}

// ---------------------------------------------------------------- Phong combine
__global__ __launch_bounds__(BLOCK_THREADS) void phong_kernel(const FrameArgs p)
{
   {
   const int band = (int) (blockIdx.x % (unsigned) N_BANDS);
   const unsigned count = p.counters[CNT_SHADOW + band];
   for (unsigned i = (blockIdx.x / N_BANDS) * BLOCK_THREADS + threadIdx.x; i < count; i += (gridDim.x / N_BANDS) * BLOCK_THREADS) {
   const unsigned path = p.shadow[(unsigned) band * p.band_cap + i];
   const float4 *q = (const float4 *) p.rays0 + 2 * (size_t) path;
   const float4 r0 = q[0], r1 = q[1];
   const float4 *hq = (const float4 *) p.hits0 + 2 * (size_t) path;
   const float4 h0 = hq[0], h1 = hq[1];
   const float4 m4 = ((const float4 *) p.mat0)[path];
   const int obj = p.hit_obj0[path];
   const DevMaterial &m0 = p.sc.objs[obj].mat;
   const float mat[3] = { m4.x, m4.y, m4.z };
   float out[3];
   phong_combine(mat, m0.diffuse, m0.specular, m0.shininess, p.light_emission, p.light_ambient,
                 mk3(p.light_origin[0], p.light_origin[1], p.light_origin[2]), mk3(h0.x, h0.y, h0.z), mk3(h1.x, h1.y, h1.z),
                 mk3(r1.x, r1.y, r1.z), m4.w != 0.0f, out);
   ((float4 *) p.samples)[__float_as_uint(r0.w)] = make_float4(out[0], out[1], out[2], 0.0f);
   }
   }
}

// ---------------------------------------------------------------- resolve
// RTStatics.computeColorAverage (sample order) + Camera.setPixel (NaN scrub, 8-bit pack)
__global__ __launch_bounds__(BLOCK_THREADS) void resolve_kernel(const FrameArgs p)
{
   const int lane = threadIdx.x & 63;
   const long long local_micro = (long long) blockIdx.x * (BLOCK_THREADS / 64) + (threadIdx.x >> 6);
   if (local_micro >= (long long) (p.micro_end - p.micro_begin)) return;
   int px, py;
   long long pix;
   if (!micro_pixel(p, p.micro_begin + (int) local_micro, lane, px, py, pix)) return;
   const int ms = p.multi_samples;
   const float4 *s = (const float4 *) p.samples + (size_t) local_micro * (size_t) ms * 64 + lane;
   const unsigned long long *qm = p.queued + (size_t) local_micro * (size_t) ms;
   const float4 ambient = make_float4(p.light_ambient[0], p.light_ambient[1], p.light_ambient[2], 0.0f);
   float r = 0.0f, g = 0.0f, b = 0.0f;
   for (int j = 0; j < ms; j++) {
      const float4 c = ((qm[j] >> lane) & 1ull) ? s[(size_t) j * 64] : ambient;   // (a sample that never became a ray: generate_kernel)
      r += c.x;
      g += c.y;
      b += c.z;
   }
   float fr = r / (float) ms, fg = g / (float) ms, fb = b / (float) ms;
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
}

// ---------------------------------------------------------------- 8-bit images (Camera.image, Camera.getNormalizedImage)
__device__ __forceinline__ unsigned pack_color(float r, float g, float b)   // new Color(bound(r), bound(g), bound(b)).getRGB()
{
   const float br = jminf(1.0f, jmaxf(0.0f, r)), bg = jminf(1.0f, jmaxf(0.0f, g)), bb = jminf(1.0f, jmaxf(0.0f, b));
   const unsigned ir = (unsigned) (int) ((double) (br * 255.0f) + 0.5);
   const unsigned ig = (unsigned) (int) ((double) (bg * 255.0f) + 0.5);
   const unsigned ib = (unsigned) (int) ((double) (bb * 255.0f) + 0.5);
   return 0xFF000000u | ((ir & 255u) << 16) | ((ig & 255u) << 8) | (ib & 255u);
}
// floats as unsigned keys whose order is the floats' order (-0 below +0, as Math.min / Math.max have it)
__device__ __forceinline__ unsigned float_key(float f)
{
   const unsigned u = __float_as_uint(f);
   return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// Camera.updateNormalizedImage, first loop (Camera.java:226-241): min and max of the HSV value over the pixels
// without a NaN.  mm[0] = key of the min (start: 0xffffffff), mm[1] = key of the max (start: 0).
__global__ __launch_bounds__(BLOCK_THREADS) void image_minmax_kernel(const float *rgb, unsigned long long n, unsigned *mm)
{
   __shared__ unsigned s_lo[BLOCK_THREADS / 64], s_hi[BLOCK_THREADS / 64];
   unsigned lo = 0xffffffffu, hi = 0u;
   for (unsigned long long i = (unsigned long long) blockIdx.x * BLOCK_THREADS + threadIdx.x; i < n; i += (unsigned long long) gridDim.x * BLOCK_THREADS) {
      const float c[3] = { rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] };
      if (c[0] != c[0] || c[1] != c[1] || c[2] != c[2]) continue;
      float hsv[3];
      rgb_to_hsv(c, hsv);
      const unsigned k = float_key(hsv[2]);
      lo = min(lo, k);
      hi = max(hi, k);
   }
   for (int off = 32; off > 0; off >>= 1) {
      lo = min(lo, (unsigned) __shfl_down((int) lo, off, 64));
      hi = max(hi, (unsigned) __shfl_down((int) hi, off, 64));
   }
   if ((threadIdx.x & 63) == 0) {
      s_lo[threadIdx.x >> 6] = lo;
      s_hi[threadIdx.x >> 6] = hi;
   }
   __syncthreads();
   if (threadIdx.x == 0) {
      for (int w = 1; w < BLOCK_THREADS / 64; w++) {
         lo = min(lo, s_lo[w]);
         hi = max(hi, s_hi[w]);
      }
      atomicMin(mm, lo);
      atomicMax(mm + 1, hi);
   }
}

// normalized == 0: Camera.setPixel's 8-bit image (Camera.java:103-107); else the second loop of
// updateNormalizedImage (Camera.java:246-258): value := (value - min) / (max - min), back to RGB, packed.
__global__ __launch_bounds__(BLOCK_THREADS) void image_pack_kernel(const float *rgb, unsigned long long n, const unsigned *mm, int normalized,
                                                                   unsigned *argb)
{
   const unsigned long long i = (unsigned long long) blockIdx.x * BLOCK_THREADS + threadIdx.x;
   if (i >= n) return;
   float c[3] = { rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] };
   if (normalized) {
      const float lo = key_float(mm[0]), hi = key_float(mm[1]);
      float hsv[3], out[3];
      rgb_to_hsv(c, hsv);
      hsv[2] = (hsv[2] - lo) / (hi - lo);
      hsv_to_rgb(hsv, out);
      c[0] = out[0];
      c[1] = out[1];
      c[2] = out[2];
   }
   argb[i] = pack_color(c[0], c[1], c[2]);
}

// ---------------------------------------------------------------- multi-GPU: tile-compact frames back into image order
// One thread per (rank, owned-tile slot, pixel of the tile).  Every rank's block has the same layout:
// [rgb: 3 n][hit_id: n][hit_obj: n][hit_w: n] words with n = slots_per_rank * tile^2 (mi355rt_tiles_unpack_device).
__global__ __launch_bounds__(BLOCK_THREADS) void tiles_unpack_kernel(const float *all, int world, int slots_per_rank, int tile, int width, int height,
                                                                     float *rgb, unsigned *argb8, int32_t *hit_id, int32_t *hit_obj, float *hit_w)
{
   const long long tt = (long long) tile * tile;
   const long long n = (long long) slots_per_rank * tt;
   const long long i = (long long) blockIdx.x * BLOCK_THREADS + threadIdx.x;
   if (i >= n * world) return;
   const int r = (int) (i / n);
   const long long e = i % n;
   const int k = (int) (e / tt), within = (int) (e % tt);
   const int tiles_x = (width + tile - 1) / tile, tiles_y = (height + tile - 1) / tile;
   const long long t = (long long) r + (long long) k * world;
   if (t >= (long long) tiles_x * tiles_y) return;
   const int px = (int) (t % tiles_x) * tile + within % tile, py = (int) (t / tiles_x) * tile + within / tile;
   if (px >= width || py >= height) return;
   const float *blk = all + (size_t) r * 6 * (size_t) n;
   const long long pix = (long long) py * width + px;
   if (rgb) {
      rgb[3 * pix + 0] = blk[3 * e + 0];
      rgb[3 * pix + 1] = blk[3 * e + 1];
      rgb[3 * pix + 2] = blk[3 * e + 2];
   }
   if (argb8) argb8[pix] = pack_color(blk[3 * e + 0], blk[3 * e + 1], blk[3 * e + 2]);   // Camera.image (Camera.java:103-107)
   if (hit_id) hit_id[pix] = __float_as_int(blk[3 * n + e]);
   if (hit_obj) hit_obj[pix] = __float_as_int(blk[4 * n + e]);
   if (hit_w) hit_w[pix] = blk[5 * n + e];
}

hipError_t launch_tiles_unpack(const float *all, int world, int slots_per_rank, int tile, int width, int height, float *rgb, unsigned *argb8,
                               int32_t *hit_id, int32_t *hit_obj, float *hit_w, hipStream_t stream)
{
   const unsigned long long n = (unsigned long long) slots_per_rank * tile * tile * (unsigned long long) world;
   if (!n) return hipSuccess;
   hipLaunchKernelGGL(tiles_unpack_kernel, dim3((unsigned) ((n + BLOCK_THREADS - 1) / BLOCK_THREADS)), dim3(BLOCK_THREADS), 0, stream, all, world,
                      slots_per_rank, tile, width, height, rgb, argb8, hit_id, hit_obj, hit_w);
   return hipGetLastError();
}

// ---------------------------------------------------------------- ray batches (mi355rt_trace_rays)
__global__ void batch_pack_kernel(const TraceBatchArgs a)
{
   const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= a.n) return;
   const V3 o = mk3(a.origins[3 * (size_t) i], a.origins[3 * (size_t) i + 1], a.origins[3 * (size_t) i + 2]);
   const V3 d = normalize3(mk3(a.dirs[3 * (size_t) i], a.dirs[3 * (size_t) i + 1], a.dirs[3 * (size_t) i + 2]));   // Ray ctor
   store_ray(a.rays, i, o, d, i, -1);
   if (i < (unsigned) N_BANDS) {   // band b holds rays [b * band_cap, min(n, (b + 1) * band_cap))
      const unsigned long long lo = (unsigned long long) i * a.band_cap;
      a.counters[CNT_Q0 + i] = lo >= a.n ? 0u : (unsigned) ((a.n - lo) < a.band_cap ? (a.n - lo) : a.band_cap);
   }
}

__global__ void batch_unpack_kernel(const TraceBatchArgs a)
{
   const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= a.n) return;
   const float4 *hq = (const float4 *) a.hits + 2 * (size_t) i;
   const float4 h0 = hq[0], h1 = hq[1];
   const int obj = a.hit_obj[i];
   if (a.out_id) a.out_id[i] = obj >= 0 ? __float_as_int(h1.w) : -1;
   if (a.out_obj) a.out_obj[i] = obj;
   if (a.out_w) a.out_w[i] = obj >= 0 ? h0.w : 0.0f;
   if (a.out_point) {
      a.out_point[3 * (size_t) i] = h0.x;
      a.out_point[3 * (size_t) i + 1] = h0.y;
      a.out_point[3 * (size_t) i + 2] = h0.z;
   }
   if (a.out_normal) {
      a.out_normal[3 * (size_t) i] = h1.x;
      a.out_normal[3 * (size_t) i + 1] = h1.y;
      a.out_normal[3 * (size_t) i + 2] = h1.z;
   }
}

// device arithmetic probe used by the GPU tests: the primitives parity rests on
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

namespace {

unsigned blocks_for(unsigned long long n) { return (unsigned) ((n + BLOCK_THREADS - 1) / BLOCK_THREADS); }

// Workgroups per CU of a persistent launch.  Every launch ends on its longest rays, and the more waves are in
// flight the more long rays are started late, so a short queue gets fewer waves (measured on 1/8 of the bench
// frame: primary 1.24 ms at 3 per CU against 1.36 at 6 and 2.09 at 1; shadow 0.38 at 1 against 0.58 at 4).  The
// queue length is only known on the device; `slots` (samples of the pass) stands in for it.
int blocks_per_cu_for(int forced, int n_cus, unsigned long long slots, unsigned long long slots_per_lane, int lo, int hi)
{
   if (forced > 0) return forced;
   const unsigned long long per = (unsigned long long) n_cus * BLOCK_THREADS * slots_per_lane;
   const long long want = (long long) ((slots + per / 2) / per);
   return (int) (want < lo ? lo : (want > hi ? hi : want));
}

unsigned persistent_grid(int n_cus, unsigned long long max_items, int blocks_per_cu)
{
   const unsigned long long want = (unsigned long long) n_cus * (unsigned long long) blocks_per_cu;   // blocks of 4 waves per CU (LDS: WALK_LDS_BYTES each)
   const unsigned long long need = (max_items + BLOCK_THREADS - 1) / BLOCK_THREADS;
   return (unsigned) (need < want ? (need ? need : 1ull) : want);
}

// grid-stride kernels (queue length known only on the device): a few blocks per CU at most
unsigned bounded_grid(int n_cus, unsigned long long max_items)
{
   const unsigned long long want = (unsigned long long) n_cus * 8ull;
   const unsigned long long need = (max_items + BLOCK_THREADS - 1) / BLOCK_THREADS;
   const unsigned long long g = need < want ? (need ? need : 1ull) : want;
   return (unsigned) ((g + N_BANDS - 1) / N_BANDS * N_BANDS);   // each workgroup serves one band
}

}   // namespace

namespace {

// Leaves of tens of faces (meshes of millions of faces under the reference's depth limit) take the walker's burst form.
// The switch MI355RT_BIG_LEAVES=0/1 forces either form (the tests run both and require identical frames).
bool big_leaves(const SceneArgs &sc, const Switches &sw) { return sw.big_leaves >= 0 ? sw.big_leaves != 0 : sc.big_leaves != 0; }

template <int MODE>
hipError_t launch_walk(const WalkArgs &a, unsigned grid, int stats_level, bool big, hipStream_t stream)
{
   if (big) {   // (no diagnostics build of this form: level 2 counts as level 1)
      if (stats_level >= 1) hipLaunchKernelGGL((walk_kernel<1, MODE, 1>), dim3(grid), dim3(BLOCK_THREADS), 0, stream, a);
      else hipLaunchKernelGGL((walk_kernel<0, MODE, 1>), dim3(grid), dim3(BLOCK_THREADS), 0, stream, a);
      return hipGetLastError();
   }
   if (stats_level > 1) hipLaunchKernelGGL((walk_kernel<2, MODE, 0>), dim3(grid), dim3(BLOCK_THREADS), 0, stream, a);
   else if (stats_level == 1) hipLaunchKernelGGL((walk_kernel<1, MODE, 0>), dim3(grid), dim3(BLOCK_THREADS), 0, stream, a);
   else hipLaunchKernelGGL((walk_kernel<0, MODE, 0>), dim3(grid), dim3(BLOCK_THREADS), 0, stream, a);
   return hipGetLastError();
}

}   // namespace

hipError_t launch_trace(const SceneArgs &sc, const float *rays, float *hits, int32_t *hit_obj, uint32_t *counters, int count_index,
                        int cursor_index, unsigned band_cap, unsigned long long max_items, int n_cus, int stats, int diag, hipStream_t stream,
                        const float *fuse_light, float *fuse_mat0)
{
   WalkArgs a;
   a.sc = sc;
   a.rays = rays;
   a.hits = hits;
   a.hit_obj = hit_obj;
   a.hits0 = nullptr;
   a.mat0 = fuse_mat0;
   a.shadow = nullptr;
   for (int c = 0; c < 3; c++) a.light[c] = fuse_light ? fuse_light[c] : 0.0f;
   a.light_prune_c = 0.0f;
   a.counters = counters;
   a.count_index = count_index;
   a.cursor_index = cursor_index;
   a.band_cap = band_cap;
   a.stats = stats;
   const Switches sw = switches();
   const bool big = big_leaves(sc, sw);
   a.trim_unit = sw.blocks_per_cu > 0 || !sw.trim ? 0u : (unsigned) n_cus * BLOCK_THREADS * (unsigned) sw.trim_primary_x2 / 2u;   // 2.5 rays per lane
   a.trim_step = (unsigned) n_cus;
   a.trim_lo = 2u;
   {   // Short passes are dealt out spread (walk_args.h).  Measured on shares of the bench frame, one frame at a time / two in
       // flight: 1.0 M samples 1.52 / 0.86 ms in order, 1.35 / 0.80 with 64 columns per 16,384 rays; 2.1 M samples 2.00 / 1.25 in
       // order, 1.75 / 1.43 with 16 per 4,096, 1.86 / 1.29 with 8 per 1,024 (balance costs coherence: what one frame gains two
       // in flight pay); 4.1 M 2.53 / 1.81 against 2.44 / 2.07; the whole 8.3 M-sample frame 3.69 / 2.99 against 3.78 / 3.38.
      const int sl = sw.spread_sl, bl = sw.spread_bl;
      const unsigned long long spread_max = (unsigned long long) (sw.spread_primary_max > 0 ? sw.spread_primary_max : 0);
      a.primary_spread = 0u;
      if (max_items < spread_max) a.primary_spread = max_items < 1500000ull ? (6u | (14u << 8)) : (3u | (10u << 8));
      if (a.primary_spread != 0u && sl >= 0 && bl >= sl && bl < 32) a.primary_spread = (unsigned) sl | ((unsigned) bl << 8);
   }
   const unsigned grid = persistent_grid(n_cus, max_items, blocks_per_cu_for(sw.blocks_per_cu, n_cus, max_items, 5, 2, big ? WALK_BIG_WAVES_PER_SIMD : WALK_WAVES_PER_SIMD));
   if (fuse_light) return launch_walk<MODE_FUSED>(a, grid, stats ? (diag ? 2 : 1) : 0, big, stream);
   return launch_walk<MODE_CLOSEST>(a, grid, stats ? (diag ? 2 : 1) : 0, big, stream);
}

hipError_t launch_generate(const FrameArgs &p, hipStream_t stream)
{
   const unsigned long long n_micro = (unsigned long long) (p.micro_end - p.micro_begin);
   const unsigned long long waves = (n_micro + N_BANDS - 1) / N_BANDS * N_BANDS * (unsigned long long) p.multi_samples;
   if (!n_micro) return hipSuccess;
   const unsigned long long waves_per_band = waves / N_BANDS;
   const unsigned grid = (unsigned) ((waves_per_band + GEN_WAVES - 1) / GEN_WAVES) * N_BANDS;
   hipLaunchKernelGGL(generate_kernel, dim3(grid), dim3(GEN_WAVES * 64), 0, stream, p);
   return hipGetLastError();
}

hipError_t launch_shade(const FrameArgs &p, int level, const float *rays, const float *hits, const int32_t *hit_obj, int count_index,
                        float *rays_next, int next_count_index, unsigned long long max_items, int n_cus, hipStream_t stream)
{
   if (!max_items) return hipSuccess;
   const unsigned grid = bounded_grid(n_cus, max_items);
   if (level == 0 && p.plain_materials)
      hipLaunchKernelGGL((shade_kernel<0, 1>), dim3(grid), dim3(BLOCK_THREADS), 0, stream, p, rays, hits, hit_obj, count_index, rays_next,
                         next_count_index);
   else if (level == 0)
      hipLaunchKernelGGL((shade_kernel<0, 0>), dim3(grid), dim3(BLOCK_THREADS), 0, stream, p, rays, hits, hit_obj, count_index, rays_next,
                         next_count_index);
   else
      hipLaunchKernelGGL((shade_kernel<1, 0>), dim3(grid), dim3(BLOCK_THREADS), 0, stream, p, rays, hits, hit_obj, count_index, rays_next,
                         next_count_index);
   return hipGetLastError();
}

hipError_t launch_shadow(const FrameArgs &p, int cursor_index, unsigned long long max_items, int n_cus, hipStream_t stream)
{
   WalkArgs a;
   a.sc = p.sc;
   a.rays = nullptr;
   a.hits = nullptr;
   a.hit_obj = nullptr;
   a.hits0 = p.hits0;
   a.mat0 = p.mat0;
   a.shadow = p.shadow;
   a.light[0] = p.light_origin[0];
   a.light[1] = p.light_origin[1];
   a.light[2] = p.light_origin[2];
   {   // shortcut 2's constant (device_util.h): a box is skipped when dist(o, box) > far(light, box) + EPSILON + 2 m, squared with
       // far <= K; m bounds the rounding of the distances (origins lie within |N| * EPSILON of the scene's box)
      const float l1 = std::fabs(a.light[0]) + std::fabs(a.light[1]) + std::fabs(a.light[2]);
      const float l2 = std::sqrt(a.light[0] * a.light[0] + a.light[1] * a.light[1] + a.light[2] * a.light[2]);
      const float m = 1e-5f * (p.sc.scene_abs_max + 3.0f * (p.sc.scene_abs_max + p.sc.max_normal_len * p.sc.epsilon) + l1);
      const float K = l2 + 1.7320508f * p.sc.scene_abs_max + m;
      const float e = p.sc.epsilon + 2.0f * m;
      a.light_prune_c = 2.0f * K * e * 1.001f + e * e;   // NaN (unknown normal lengths, non-finite light) disables the skip
   }
   a.counters = p.counters;
   a.count_index = CNT_SHADOW;
   a.cursor_index = cursor_index;
   a.band_cap = p.band_cap;
   a.stats = p.stats;
   const Switches sw = switches();
   const bool big = big_leaves(p.sc, sw);
   a.trim_unit = sw.blocks_per_cu_shadow > 0 || !sw.trim ? 0u : (unsigned) n_cus * BLOCK_THREADS * (unsigned) (sw.trim_shadow_x2 > 0 ? sw.trim_shadow_x2 : (big ? 3 : 12)) / 2u;   // 6 rays per lane; 1.5 where a ray is hundreds of triangle tests (measured: Thai- and Lucy-class)
   a.trim_step = (unsigned) n_cus;
   a.trim_lo = 1u;
   a.primary_spread = 0u;
   const unsigned grid = persistent_grid(n_cus, max_items, blocks_per_cu_for(sw.blocks_per_cu_shadow, n_cus, max_items, 25, 1, big ? WALK_BIG_WAVES_PER_SIMD : 4));
   const int level = p.stats ? (p.diag ? 2 : 1) : 0;
   if (p.shadow_any_hit) return launch_walk<MODE_SHADOW_ANY>(a, grid, level, big, stream);
   return launch_walk<MODE_SHADOW>(a, grid, level, big, stream);
}

hipError_t launch_phong(const FrameArgs &p, unsigned long long max_items, int n_cus, hipStream_t stream)
{
   if (!max_items) return hipSuccess;
   hipLaunchKernelGGL(phong_kernel, dim3(bounded_grid(n_cus, max_items)), dim3(BLOCK_THREADS), 0, stream, p);
   return hipGetLastError();
}

hipError_t launch_resolve(const FrameArgs &p, hipStream_t stream)
{
   const unsigned long long micro = (unsigned long long) (p.micro_end - p.micro_begin);
   if (!micro) return hipSuccess;
   hipLaunchKernelGGL(resolve_kernel, dim3(blocks_for(micro * 64)), dim3(BLOCK_THREADS), 0, stream, p);
   return hipGetLastError();
}

hipError_t launch_image(const float *rgb, unsigned long long n_pixels, int normalized, unsigned *minmax2, unsigned *argb, int n_cus, hipStream_t stream)
{
   if (!n_pixels) return hipSuccess;
   if (normalized) {
      const unsigned init[2] = { 0xffffffffu, 0u };
      hipError_t e = hipMemcpyAsync(minmax2, init, sizeof(init), hipMemcpyHostToDevice, stream);
      if (e != hipSuccess) return e;
      const unsigned long long want = (unsigned long long) n_cus * 4ull, need = (n_pixels + BLOCK_THREADS - 1) / BLOCK_THREADS;
      hipLaunchKernelGGL(image_minmax_kernel, dim3((unsigned) (need < want ? need : want)), dim3(BLOCK_THREADS), 0, stream, rgb, n_pixels, minmax2);
   }
   hipLaunchKernelGGL(image_pack_kernel, dim3(blocks_for(n_pixels)), dim3(BLOCK_THREADS), 0, stream, rgb, n_pixels, minmax2, normalized, argb);
   return hipGetLastError();
}

hipError_t launch_batch_pack(const TraceBatchArgs &a, hipStream_t stream)
{
   hipLaunchKernelGGL(batch_pack_kernel, dim3(blocks_for(a.n)), dim3(BLOCK_THREADS), 0, stream, a);
   return hipGetLastError();
}

hipError_t launch_batch_unpack(const TraceBatchArgs &a, hipStream_t stream)
{
   hipLaunchKernelGGL(batch_unpack_kernel, dim3(blocks_for(a.n)), dim3(BLOCK_THREADS), 0, stream, a);
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
