// kernels.h -- launch wrappers implemented in trace_kernels.hip (internal to libmi355rt).
#pragma once
#include <hip/hip_runtime_api.h>

#include "layout.h"

namespace mi355rt {

hipError_t launch_generate(const FrameArgs &p, hipStream_t stream);
// stats: 0 none, 1 count boxes / triangles, 2 also count hits (the primary queue)
hipError_t launch_trace(const SceneArgs &sc, const float *rays, float *hits, int32_t *hit_obj, uint32_t *counters, int count_index,
                        int cursor_index, unsigned band_cap, unsigned long long max_items, int n_cus, int stats, int diag, hipStream_t stream,
                        const float *fuse_light = nullptr, float *fuse_mat0 = nullptr);   // both set: MODE_FUSED (primary + shadow rays in one launch)
hipError_t launch_shade(const FrameArgs &p, int level, const float *rays, const float *hits, const int32_t *hit_obj, int count_index,
                        float *rays_next, int next_count_index, unsigned long long max_items, int n_cus, hipStream_t stream);
hipError_t launch_shadow(const FrameArgs &p, int cursor_index, unsigned long long max_items, int n_cus, hipStream_t stream);
// spheres and planes (analytic_kernels.hip): fold their hits into the records the walker left; no-ops for mesh-only scenes
hipError_t launch_analytic_closest(const SceneArgs &sc, const float *rays, float *hits, int32_t *hit_obj, uint32_t *counters, int count_index,
                                   unsigned band_cap, unsigned long long max_items, int n_cus, int count_hits, hipStream_t stream);
hipError_t launch_analytic_shadow(const FrameArgs &p, unsigned long long max_items, int n_cus, hipStream_t stream);
hipError_t launch_phong(const FrameArgs &p, unsigned long long max_items, int n_cus, hipStream_t stream);
hipError_t launch_resolve(const FrameArgs &p, hipStream_t stream);
hipError_t launch_image(const float *rgb, unsigned long long n_pixels, int normalized, unsigned *minmax2, unsigned *argb, int n_cus, hipStream_t stream);
hipError_t launch_tiles_unpack(const float *all, int world, int slots_per_rank, int tile, int width, int height, float *rgb, unsigned *argb8,
                               int32_t *hit_id, int32_t *hit_obj, float *hit_w, hipStream_t stream);
hipError_t launch_batch_pack(const TraceBatchArgs &a, hipStream_t stream);
hipError_t launch_batch_unpack(const TraceBatchArgs &a, hipStream_t stream);
hipError_t launch_arith_probe(unsigned long long n, const float *a, const float *b, float *out_div, float *out_sqrt,
                              float *out_rnorm, float *out_pow, hipStream_t stream);

}   // namespace mi355rt
