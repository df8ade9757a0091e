// analytic_kernels.hip -- spheres and planes in the wavefront pipeline.
//
// The walker (trace_kernels.hip) answers a queue of rays for the MESH objects.  Scenes that also hold spheres or
// planes run one of these kernels right after it, over the same queue: each ray is tested against the few
// analytic objects and the result is folded into the walker's record with the reference's rule
// (Camera.java:204-212: objects in array order, strictly smaller w wins, so equal w goes to the lower index).
// One lane per ray, a handful of objects each: bandwidth-bound over 68 B of queue records per ray, and absent
// from frames whose scene has only meshes.
#include <hip/hip_runtime.h>

#include "analytic.h"
#include "device_util.h"
#include "kernels.h"
#include "layout.h"

namespace mi355rt {

// Camera.getClosestIntersection's loop (Camera.java:204-212) for the sphere / plane objects, merged into hits[].
__global__ __launch_bounds__(BLOCK_THREADS) void analytic_closest_kernel(const SceneArgs sc, const float *rays, float *hits, int32_t *hit_obj,
                                                                         uint32_t *counters, int count_index, unsigned band_cap,
                                                                         int count_hits)
{
   const int band = (int) (blockIdx.x % (unsigned) N_BANDS);   // a workgroup serves one band (grid is a multiple of N_BANDS)
   const unsigned count = counters[count_index + band];
   unsigned new_hits = 0;
   for (unsigned i = (blockIdx.x / N_BANDS) * BLOCK_THREADS + threadIdx.x; i < count; i += (gridDim.x / N_BANDS) * BLOCK_THREADS) {
      const size_t idx = (size_t) band * band_cap + i;
      const float4 *q = (const float4 *) rays + 2 * idx;
      const float4 r0 = q[0], r1 = q[1];
      const int mirror = (int) (short) (__float_as_int(r1.w) & 0xffff);
      Ray r;
      make_ray(r, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z));
      float4 *hq = (float4 *) hits + 2 * idx;
      const int walk_obj = hit_obj[idx];
      int best_obj = walk_obj;
      float best_w = walk_obj >= 0 ? hq[0].w : 0.0f;
      V3 bP = mk3(0, 0, 0), bN = mk3(0, 0, 0);
      for (int k = 0; k < sc.n_objs; k++) {
         const DevObject &ob = sc.objs[k];
         if (ob.kind == OBJ_MESH || k == mirror) continue;   // Camera.java:205 skips mirrorObject
         const DevObjectExt &e = sc.ext[k];
         if (!analytic_intersects(ob, e, r, sc.epsilon, sc.near_plane, sc.far_plane)) continue;
         V3 P, N;
         float w;
         if (!analytic_child_intersection(ob, e, r, sc.epsilon, P, N, w)) continue;
         if ((w > sc.epsilon) && (best_obj < 0 || w < best_w || (w == best_w && k < best_obj))) {   // Camera.java:208
            best_w = w;
            best_obj = k;
            bP = P;
            bN = N;
         }
      }
      if (best_obj != walk_obj) {
         hq[0] = make_float4(bP.x, bP.y, bP.z, best_w);
         hq[1] = make_float4(bN.x, bN.y, bN.z, __int_as_float(0));
         hit_obj[idx] = best_obj;
         if (walk_obj < 0) new_hits++;
      }
   }
   if (count_hits) add_stat(counters, ST_HITS, new_hits);
}

// RTStatics.shadowIntersects' loop (RTStatics.java:625-636) for the sphere / plane objects: sets the path's
// occluded flag (the walker has already written 0 or 1 for the meshes).
__global__ __launch_bounds__(BLOCK_THREADS) void analytic_shadow_kernel(const FrameArgs p)
{
   const int band = (int) (blockIdx.x % (unsigned) N_BANDS);
   const unsigned count = p.counters[CNT_SHADOW + band];
   const V3 light = mk3(p.light_origin[0], p.light_origin[1], p.light_origin[2]);
   const float eps = p.sc.epsilon;
   for (unsigned i = (blockIdx.x / N_BANDS) * BLOCK_THREADS + threadIdx.x; i < count; i += (gridDim.x / N_BANDS) * BLOCK_THREADS) {
      const unsigned path = p.shadow[(size_t) band * p.band_cap + i];
      if (p.mat0[4 * (size_t) path + 3] != 0.0f) continue;   // a mesh already occludes
      const float4 *hq = (const float4 *) p.hits0 + 2 * (size_t) path;
      const float4 h0 = hq[0], h1 = hq[1];
      const V3 P0 = mk3(h0.x, h0.y, h0.z), N0 = mk3(h1.x, h1.y, h1.z);
      Ray r;
      make_ray(r, add3(P0, scale3(N0, eps)), normalize3(normalize3(sub3(light, P0))));   // RTStatics.java:618-622, Ray ctor
      bool occluded = false;
      for (int k = 0; k < p.sc.n_objs && !occluded; k++) {
         const DevObject &ob = p.sc.objs[k];
         if (ob.kind == OBJ_MESH) continue;
         V3 P, N;
         float w;
         if (!analytic_child_intersection(ob, p.sc.ext[k], r, eps, P, N, w)) continue;
         if (w > eps) {   // RTStatics.java:628-633
            const float light_distance = distance3(P, light);
            if (w < light_distance + eps) occluded = true;
         }
      }
      if (occluded) p.mat0[4 * (size_t) path + 3] = 1.0f;
   }
}

namespace {

unsigned band_grid(int n_cus, unsigned long long max_items)
{
   const unsigned long long want = (unsigned long long) n_cus * 8ull;
   const unsigned long long need = (max_items + BLOCK_THREADS - 1) / BLOCK_THREADS;
   const unsigned long long g = need < want ? (need ? need : 1ull) : want;
   return (unsigned) ((g + N_BANDS - 1) / N_BANDS * N_BANDS);   // each workgroup serves one band
}

}   // namespace

hipError_t launch_analytic_closest(const SceneArgs &sc, const float *rays, float *hits, int32_t *hit_obj, uint32_t *counters, int count_index,
                                   unsigned band_cap, unsigned long long max_items, int n_cus, int count_hits, hipStream_t stream)
{
   if (!max_items || sc.n_analytic == 0) return hipSuccess;
   hipLaunchKernelGGL(analytic_closest_kernel, dim3(band_grid(n_cus, max_items)), dim3(BLOCK_THREADS), 0, stream, sc, rays, hits, hit_obj,
                      counters, count_index, band_cap, count_hits);
   return hipGetLastError();
}

hipError_t launch_analytic_shadow(const FrameArgs &p, unsigned long long max_items, int n_cus, hipStream_t stream)
{
   if (!max_items || p.sc.n_analytic == 0) return hipSuccess;
   hipLaunchKernelGGL(analytic_shadow_kernel, dim3(band_grid(n_cus, max_items)), dim3(BLOCK_THREADS), 0, stream, p);
   return hipGetLastError();
}

}   // namespace mi355rt
