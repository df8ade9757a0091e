// kernels.h -- launch wrappers implemented in trace_kernels.hip (internal to libmi355rt).
#pragma once
#include <hip/hip_runtime_api.h>

#include "layout.h"

namespace mi355rt {

hipError_t launch_render(const RenderParams &p, bool stats, hipStream_t stream);
hipError_t launch_trace_rays(const TraceParams &p, hipStream_t stream);
hipError_t launch_arith_probe(unsigned long long n, const float *a, const float *b, float *out_div, float *out_sqrt,
                              float *out_rnorm, float *out_pow, hipStream_t stream);

}   // namespace mi355rt
