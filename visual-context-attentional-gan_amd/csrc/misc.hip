// Loss reductions and the fused Adam update.
#include "common.h"

namespace {

constexpr int THR = 256;

__device__ __forceinline__ float block_sum(float v) {
  __shared__ float part[THR / 64];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < THR / 64; ++i) t += part[i];
  __syncthreads();
  return t;  // valid in thread 0
}

__device__ __forceinline__ float softplus_f(float z) {
  // torch.nn.functional.softplus (beta 1, threshold 20)
  return z > 20.f ? z : log1pf(expf(z));
}

// out[0] += scale * sum softplus(sign * x)            (generator.py:363-366: the mean over the logits)
__global__ __launch_bounds__(THR) void softplus_sum_kernel(int n, const float* __restrict__ x, float sign,
                                                           float scale, float* out) {
  float s = 0.f;
  for (int i = blockIdx.x * THR + threadIdx.x; i < n; i += gridDim.x * THR) s += softplus_f(sign * x[i]);
  s = block_sum(s);
  if (threadIdx.x == 0) atomicAdd(out, s * scale);
}

// gx = gout[0] * scale * sign * sigmoid(sign * x)
__global__ __launch_bounds__(THR) void softplus_bwd_kernel(int n, const float* __restrict__ x, float sign,
                                                           float scale, const float* __restrict__ gout,
                                                           float* __restrict__ gx) {
  const float g = gout[0] * scale * sign;
  for (int i = blockIdx.x * THR + threadIdx.x; i < n; i += gridDim.x * THR) {
    const float z = sign * x[i];
    gx[i] = g / (1.f + expf(-z));
  }
}

// out[0] += scale * sum |a*mul + add - (b*mul + add)| = scale*|mul| * sum |a - b|
__global__ __launch_bounds__(THR) void l1_sum_kernel(size_t n, const float* __restrict__ a,
                                                     const float* __restrict__ b, float scale, float* out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < n; i += (size_t)gridDim.x * THR)
    s += fabsf(a[i] - b[i]);
  s = block_sum(s);
  if (threadIdx.x == 0) atomicAdd(out, s * scale);
}

__global__ __launch_bounds__(THR) void l1_bwd_kernel(size_t n, const float* __restrict__ a,
                                                     const float* __restrict__ b, float scale,
                                                     const float* __restrict__ gout, float* __restrict__ ga) {
  const float g = gout[0] * scale;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < n; i += (size_t)gridDim.x * THR) {
    const float d = a[i] - b[i];
    ga[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
  }
}

// torch.optim.Adam (coupled L2 weight decay, optional amsgrad) over one flat parameter buffer
__global__ __launch_bounds__(THR) void adam_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ vmax, float lr, float b1, float b2,
                                                   float eps, float wd, float bc1, float bc2_sqrt,
                                                   const int* __restrict__ step_dev) {
  const size_t n4 = n >> 2;
  if (step_dev) {   // step count lives on the device (the launch is replayed from a HIP graph)
    const double t = (double)step_dev[0];
    bc1 = (float)(1.0 - pow((double)b1, t));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  }
  const float step = lr / bc1;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < n4; i += (size_t)gridDim.x * THR) {
    float4 P = reinterpret_cast<float4*>(p)[i];
    const float4 G = reinterpret_cast<const float4*>(g)[i];
    float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
    float4 X = vmax ? reinterpret_cast<float4*>(vmax)[i] : make_float4(0, 0, 0, 0);
    float pp[4] = {P.x, P.y, P.z, P.w}, gg[4] = {G.x, G.y, G.z, G.w}, mm[4] = {M.x, M.y, M.z, M.w},
          vv[4] = {V.x, V.y, V.z, V.w}, xx[4] = {X.x, X.y, X.z, X.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gg[j] + wd * pp[j];
      mm[j] = b1 * mm[j] + (1.f - b1) * gr;
      vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;
      float den;
      if (vmax) {
        xx[j] = fmaxf(xx[j], vv[j]);
        den = sqrtf(xx[j]) / bc2_sqrt + eps;
      } else {
        den = sqrtf(vv[j]) / bc2_sqrt + eps;
      }
      pp[j] -= step * mm[j] / den;
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (vmax) reinterpret_cast<float4*>(vmax)[i] = make_float4(xx[0], xx[1], xx[2], xx[3]);
  }
}

int grid_for(size_t n) {
  size_t g = (n + THR - 1) / THR;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int vca_softplus_sum_f32(int n, const float* x, float sign, float scale, float* out, void* stream) {
  VCA_CHECK_ARG(n > 0 && x && out);
  hipLaunchKernelGGL(softplus_sum_kernel, dim3(grid_for(n)), dim3(THR), 0, (hipStream_t)stream, n, x, sign, scale, out);
  return vca_launch_status();
}

extern "C" int vca_softplus_bwd_f32(int n, const float* x, float sign, float scale, const float* gout, float* gx,
                                    void* stream) {
  VCA_CHECK_ARG(n > 0 && x && gout && gx);
  hipLaunchKernelGGL(softplus_bwd_kernel, dim3(grid_for(n)), dim3(THR), 0, (hipStream_t)stream, n, x, sign, scale,
                     gout, gx);
  return vca_launch_status();
}

extern "C" int vca_l1_sum_f32(long long n, const float* a, const float* b, float scale, float* out, void* stream) {
  VCA_CHECK_ARG(n > 0 && a && b && out);
  hipLaunchKernelGGL(l1_sum_kernel, dim3(grid_for((size_t)n)), dim3(THR), 0, (hipStream_t)stream, (size_t)n, a, b,
                     scale, out);
  return vca_launch_status();
}

extern "C" int vca_l1_bwd_f32(long long n, const float* a, const float* b, float scale, const float* gout, float* ga,
                              void* stream) {
  VCA_CHECK_ARG(n > 0 && a && b && gout && ga);
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for((size_t)n)), dim3(THR), 0, (hipStream_t)stream, (size_t)n, a, b,
                     scale, gout, ga);
  return vca_launch_status();
}

extern "C" int vca_adam_f32(long long n, float* p, const float* g, float* m, float* v, float* vmax, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int step, const int* step_dev,
                            void* stream) {
  VCA_CHECK_ARG(n > 0 && (n % 4) == 0 && p && g && m && v && (step >= 1 || step_dev));
  if (step < 1) step = 1;
  const double bc1d = 1.0 - pow((double)beta1, (double)step), bc2d = 1.0 - pow((double)beta2, (double)step);
  const float bc1 = (float)bc1d;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for((size_t)n / 4)), dim3(THR), 0, (hipStream_t)stream, (size_t)n, p, g, m,
                     v, vmax, lr, beta1, beta2, eps, weight_decay, bc1, (float)sqrt(bc2d), step_dev);
  return vca_launch_status();
}
