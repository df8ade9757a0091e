// walk_args.h -- what a walker launch is given (trace_kernels.hip).  Internal to libmi355rt.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "layout.h"

namespace mi355rt {

enum { MODE_CLOSEST = 0, MODE_SHADOW = 1, MODE_SHADOW_ANY = 2, MODE_FUSED = 3 };

struct WalkArgs {
   SceneArgs sc;
   const float *rays;        // MODE_CLOSEST: the queue
   float *hits;
   int32_t *hit_obj;
   const float *hits0;       // shadow modes: primary hits, material/flag records, list of paths
   float *mat0;
   const uint32_t *shadow;
   float light[3];
   float light_prune_c;      // MODE_SHADOW: a box is hopeless when dist2(origin, box) > 1.0001 * far2(light, box) + this (device_util.h); NaN = never
   uint32_t *counters;
   int32_t count_index, cursor_index;   // first of N_BANDS counters / cursors
   uint32_t band_cap;                   // slots per band sub-queue
   int32_t stats;
   // The host sizes the grid from the pass's samples (the queue's length is only known on the device); a workgroup
   // beyond round(rays queued / trim_unit) x trim_step, at least trim_lo, leaves at once (0: no trimming).
   uint32_t trim_unit, trim_step, trim_lo;
   // closest-hit queues: 0 = rays are taken in queue order (image order: coherent waves, what long passes want); else the
   // queue is dealt out as the shadow queue is -- transposed, spread_log2 columns per block of 2^block_log2 rays -- so
   // that the long rays of one patch of the image land in different waves (short passes are all launch end: the slowest
   // wave of a 1/8 share of the bench frame took 1.0 ms, the average one 0.3).  low byte spread_log2, next byte block_log2.
   uint32_t primary_spread;
};

}   // namespace mi355rt
