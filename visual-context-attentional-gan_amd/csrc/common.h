// Shared declarations for the gfx950 kernels of the VCA-GAN hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/vca_hip.h"

// Element type of the 16-bit matrix operands.  The library is built twice from the same sources: with bf16
// (libvca_hip.so: fp32's exponent range, 8 significant bits) and, with -DVCA_E16_FP16, with IEEE half
// (libvca_hip_f16.so: 11 significant bits, normal range 6.1e-5 .. 65504, so gradients need a loss scale).  Entry
// points, layouts and kernel names are the same in both; vca_elem_type() tells which one a library is.
#ifdef VCA_E16_FP16
typedef _Float16 vca_e16;
#define VCA_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define VCA_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define VCA_ELEM_TYPE 1
#else
typedef __bf16 vca_e16;
#define VCA_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define VCA_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define VCA_ELEM_TYPE 0
#endif
typedef vca_e16 bf16x8 __attribute__((ext_vector_type(8)));   // 8 operand elements of either type: one MFMA fragment

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define VCA_CHECK_ARG(cond) \
  do {                      \
    if (!(cond)) return -22; \
  } while (0)

// Kernels that need more than 64 KiB of dynamic LDS must be told so once per DEVICE (the attribute lives with the
// device's copy of the function), and library calls may come from several host threads.  `done` is the call site's
// own bit mask of devices already served.  Returns 0 or a hipError_t.
static inline int vca_max_lds_once(const void* kernel, std::atomic<unsigned long long>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return (int)e;
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}
#define VCA_MAX_LDS(kernel)                                                             \
  do {                                                                                  \
    static std::atomic<unsigned long long> done_{0};                                    \
    const int e_ = vca_max_lds_once(reinterpret_cast<const void*>(kernel), done_);      \
    if (e_) return e_;                                                                  \
  } while (0)

static inline int vca_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// Zero-fill of a float buffer as an ordinary kernel.  Used instead of hipMemsetAsync in front of kernels that
// accumulate into the buffer with atomics: inside a captured graph a memset becomes a memset node, which the
// runtime executes on its own path, and replays were observed to start the accumulating kernel before the fill had
// landed (results changed with unrelated host activity between replays).  A kernel node is ordered like any other.
static __global__ __launch_bounds__(256) void vca_zero_kernel(float* p, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    float4* p4 = reinterpret_cast<float4*>(p);
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = 0.f;
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = 0.f;
  }
}

static inline void vca_zero_async(float* p, size_t n, hipStream_t s) {
  size_t g = (n / 4 + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(vca_zero_kernel, dim3((unsigned)g), dim3(256), 0, s, p, n);
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
