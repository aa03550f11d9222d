// Shared declarations for the gfx950 kernels of the VCA-GAN hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vca_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define VCA_CHECK_ARG(cond) \
  do {                      \
    if (!(cond)) return -22; \
  } while (0)

static inline int vca_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// Activation codes shared by the conv prologue and the pointwise kernels.
enum { VCA_ACT_NONE = 0, VCA_ACT_LEAKY = 1, VCA_ACT_PRELU = 2, VCA_ACT_RELU = 3, VCA_ACT_TANH = 4 };

__device__ __forceinline__ float vca_act(float z, int act, float slope) {
  // slope: the leaky slope, or this channel's PReLU weight
  if (act == VCA_ACT_LEAKY || act == VCA_ACT_PRELU) return z > 0.f ? z : z * slope;
  if (act == VCA_ACT_RELU) return z > 0.f ? z : 0.f;
  return z;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
