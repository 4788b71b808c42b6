// Wave-level helpers of the output-layer kernels (learner.hip: head_kernel; serve_step.hip evaluates the same heads inside
// the env step's launch with the same operations in the same order).
#pragma once
#include "arl_common.h"

namespace {

__device__ __forceinline__ float readlane_f(float x, int uniform_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), uniform_lane));
}
// butterfly sum over the 64 lanes (every lane ends with the same total)
__device__ __forceinline__ float wave_sum_f(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

}  // namespace
