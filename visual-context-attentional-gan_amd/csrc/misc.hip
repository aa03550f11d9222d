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
                                                   const int* __restrict__ step_dev,
                                                   const float* __restrict__ lr_dev) {
  const size_t n4 = n >> 2;
  if (lr_dev) lr = lr_dev[0];   // learning rate lives on the device: a replayed launch follows the schedule
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

// ---------------------------------------------------------------------------------------------------------------
// Dynamic loss scaling (half-precision gradients: the backward pass is seeded with `scale` instead of 1, every
// gradient in the flat buffers is `scale` times too large, and an overflow anywhere shows up as Inf / NaN there).
// state: int[4] = good steps in a row, overflow flag of the discriminator side, of the generator side, skipped steps.
// scale: float[2] = scale, 1 / scale.  Everything lives in device words so that a captured step keeps adapting.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(THR) void nonfinite_kernel(size_t n4, const float4* __restrict__ g, int* __restrict__ flag) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < n4; i += (size_t)gridDim.x * THR) {
    const float4 v = g[i];
    // (x - x) is 0 for finite x and NaN for Inf / NaN
    const float t = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
    bad |= !(t == 0.f);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ void loss_scale_count_kernel(int* __restrict__ step_dev, const int* __restrict__ flag) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && flag[0] == 0) step_dev[0] += 1;
}

__global__ void loss_scale_update_kernel(float* __restrict__ scale, int* __restrict__ state, float growth,
                                         float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = scale[0];
  if (state[1] | state[2]) {
    s = fmaxf(s * backoff, 1.f);
    state[0] = 0;
    state[3] += 1;
  } else if (++state[0] >= interval) {
    s = fminf(s * growth, 16777216.f);
    state[0] = 0;
  }
  state[1] = 0;
  state[2] = 0;
  scale[0] = s;
  scale[1] = 1.f / s;
}

// adam_kernel on gradients that carry the loss scale: g / scale enters the update, and the whole launch is a no-op
// when the side's overflow flag is up (parameters, moments and step count stay as they are).
__global__ __launch_bounds__(THR) void adam_scaled_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          float* __restrict__ vmax, float b1, float b2, float eps,
                                                          float wd, const int* __restrict__ step_dev,
                                                          const float* __restrict__ lr_dev,
                                                          const float* __restrict__ scale,
                                                          const int* __restrict__ flag) {
  if (flag[0]) return;
  const size_t n4 = n >> 2;
  const double t = (double)step_dev[0];
  const float bc1 = (float)(1.0 - pow((double)b1, t)), bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  const float step = lr_dev[0] / bc1, inv = scale[1];
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < n4; i += (size_t)gridDim.x * THR) {
    float4 P = reinterpret_cast<float4*>(p)[i];
    const float4 G = reinterpret_cast<const float4*>(g)[i];
    float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
    float4 X = vmax ? reinterpret_cast<float4*>(vmax)[i] : make_float4(0, 0, 0, 0);
    float pp[4] = {P.x, P.y, P.z, P.w}, gg[4] = {G.x, G.y, G.z, G.w}, mm[4] = {M.x, M.y, M.z, M.w},
          vv[4] = {V.x, V.y, V.z, V.w}, xx[4] = {X.x, X.y, X.z, X.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gg[j] * inv + wd * pp[j];
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

// ---------------------------------------------------------------------------------------------------------------
// All loss heads of a phase in one launch (train.py:200-231: 18 softplus means, 4 L1 means, 3 R1 squared norms, the
// sync-loss means and the weighted sums that combine them).  A head e reduces v_e = scale_e * sum_i f_e(x_i[, y_i])
// and adds w_ek * v_e into up to three slots of the output vector (a reported term and the totals it is part of).
// ---------------------------------------------------------------------------------------------------------------
enum { HEAD_SOFTPLUS = 0, HEAD_L1 = 1, HEAD_SQSUM = 2, HEAD_SUM = 3 };
constexpr int HEAD_CHUNK = 16384;   // elements one workgroup reduces

// The head table travels in the kernel arguments (at most VCA_MAX_LOSS_HEADS heads): no device table to keep alive,
// and a launch captured into a graph carries its own copy.
__device__ __forceinline__ int head_of_block(const vca_loss_heads& t, int b) {
  int e = 0;   // last head whose block_begin <= b
  for (int i = 1; i < t.nhead; ++i)
    if (t.h[i].block_begin <= b) e = i;
  return e;
}

__global__ __launch_bounds__(THR) void loss_heads_fwd_kernel(const vca_loss_heads tab, float* __restrict__ out) {
  const int e = head_of_block(tab, blockIdx.x);
  const vca_loss_head h = tab.h[e];
  const long long beg = (long long)(blockIdx.x - h.block_begin) * HEAD_CHUNK;
  const long long end = beg + HEAD_CHUNK < h.n ? beg + HEAD_CHUNK : h.n;
  const float* __restrict__ x = h.x;
  const float* __restrict__ y = h.y;
  float s = 0.f;
  for (long long i = beg + threadIdx.x; i < end; i += THR) {
    const float v = x[i];
    if (h.kind == HEAD_SOFTPLUS) s += softplus_f(h.sign * v);
    else if (h.kind == HEAD_L1) s += fabsf(v - y[i]);
    else if (h.kind == HEAD_SQSUM) s += v * v;
    else s += v;
  }
  s = block_sum(s);
  if (threadIdx.x == 0) {
    s *= h.scale;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (h.slot[k] >= 0) atomicAdd(out + h.slot[k], s * h.weight[k]);
  }
}

// gx_e = (sum_k gout[slot_ek] * w_ek) * scale_e * f_e'(x); heads without a gx pointer need no gradient
__global__ __launch_bounds__(THR) void loss_heads_bwd_kernel(const vca_loss_heads tab,
                                                             const float* __restrict__ gout) {
  const int e = head_of_block(tab, blockIdx.x);
  const vca_loss_head h = tab.h[e];
  if (!h.gx) return;
  float g = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (h.slot[k] >= 0) g += gout[h.slot[k]] * h.weight[k];
  g *= h.scale;
  const long long beg = (long long)(blockIdx.x - h.block_begin) * HEAD_CHUNK;
  const long long end = beg + HEAD_CHUNK < h.n ? beg + HEAD_CHUNK : h.n;
  const float* __restrict__ x = h.x;
  const float* __restrict__ y = h.y;
  float* __restrict__ gx = h.gx;
  for (long long i = beg + threadIdx.x; i < end; i += THR) {
    const float v = x[i];
    float r;
    if (h.kind == HEAD_SOFTPLUS) r = g * h.sign / (1.f + expf(-h.sign * v));
    else if (h.kind == HEAD_L1) { const float d = v - y[i]; r = d > 0.f ? g : (d < 0.f ? -g : 0.f); }
    else if (h.kind == HEAD_SQSUM) r = 2.f * g * v;
    else r = g;
    gx[i] = r;
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
                            const float* lr_dev, void* stream) {
  VCA_CHECK_ARG(n > 0 && (n % 4) == 0 && p && g && m && v && (step >= 1 || step_dev));
  if (step < 1) step = 1;
  const double bc1d = 1.0 - pow((double)beta1, (double)step), bc2d = 1.0 - pow((double)beta2, (double)step);
  const float bc1 = (float)bc1d;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for((size_t)n / 4)), dim3(THR), 0, (hipStream_t)stream, (size_t)n, p, g, m,
                     v, vmax, lr, beta1, beta2, eps, weight_decay, bc1, (float)sqrt(bc2d), step_dev, lr_dev);
  return vca_launch_status();
}

extern "C" int vca_nonfinite_f32(long long n, const float* g, int* flag, void* stream) {
  VCA_CHECK_ARG(n > 0 && (n % 4) == 0 && g && flag && (reinterpret_cast<uintptr_t>(g) & 15) == 0);
  hipLaunchKernelGGL(nonfinite_kernel, dim3(grid_for((size_t)n / 4)), dim3(THR), 0, (hipStream_t)stream, (size_t)n / 4,
                     reinterpret_cast<const float4*>(g), flag);
  return vca_launch_status();
}

extern "C" int vca_loss_scale_count(int* step_dev, const int* flag, void* stream) {
  VCA_CHECK_ARG(step_dev && flag);
  hipLaunchKernelGGL(loss_scale_count_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev, flag);
  return vca_launch_status();
}

extern "C" int vca_loss_scale_update(float* scale, int* state, float growth, float backoff, int interval,
                                     void* stream) {
  VCA_CHECK_ARG(scale && state && growth >= 1.f && backoff > 0.f && backoff <= 1.f && interval >= 1);
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scale, state, growth,
                     backoff, interval);
  return vca_launch_status();
}

extern "C" int vca_adam_scaled_f32(long long n, float* p, const float* g, float* m, float* v, float* vmax, float beta1,
                                   float beta2, float eps, float weight_decay, const int* step_dev,
                                   const float* lr_dev, const float* scale, const int* flag, void* stream) {
  VCA_CHECK_ARG(n > 0 && (n % 4) == 0 && p && g && m && v && step_dev && lr_dev && scale && flag);
  hipLaunchKernelGGL(adam_scaled_kernel, dim3(grid_for((size_t)n / 4)), dim3(THR), 0, (hipStream_t)stream, (size_t)n, p,
                     g, m, v, vmax, beta1, beta2, eps, weight_decay, step_dev, lr_dev, scale, flag);
  return vca_launch_status();
}

extern "C" int vca_loss_head_chunk(void) { return HEAD_CHUNK; }

static bool heads_ok(const vca_loss_heads* t) {
  if (!t || t->nhead < 1 || t->nhead > VCA_MAX_LOSS_HEADS || t->total_blocks < 1) return false;
  int next = 0;
  for (int i = 0; i < t->nhead; ++i) {
    const vca_loss_head& h = t->h[i];
    if (!h.x || h.n < 1 || h.kind < 0 || h.kind > 3 || (h.kind == HEAD_L1 && !h.y) || h.block_begin != next) return false;
    for (int k = 0; k < 3; ++k)
      if (h.slot[k] >= VCA_MAX_LOSS_SLOTS) return false;
    next += (int)((h.n + HEAD_CHUNK - 1) / HEAD_CHUNK);
  }
  return next == t->total_blocks;
}

extern "C" int vca_loss_heads_fwd_f32(const vca_loss_heads* heads, float* out, void* stream) {
  VCA_CHECK_ARG(heads_ok(heads) && out);
  hipLaunchKernelGGL(loss_heads_fwd_kernel, dim3(heads->total_blocks), dim3(THR), 0, (hipStream_t)stream, *heads, out);
  return vca_launch_status();
}

extern "C" int vca_loss_heads_bwd_f32(const vca_loss_heads* heads, const float* gout, void* stream) {
  VCA_CHECK_ARG(heads_ok(heads) && gout);
  hipLaunchKernelGGL(loss_heads_bwd_kernel, dim3(heads->total_blocks), dim3(THR), 0, (hipStream_t)stream, *heads,
                     gout);
  return vca_launch_status();
}
