// Data-movement kernels of the step on channels-last maps: channel concatenation with broadcast (the decoder's
// [phonemes tiled over frequency | noise] input, the attention merges, the discriminators' conditioning vector),
// window means (AvgPool2d(4) head, spatial means of the discriminator heads), dropout, and the mel pyramid.
#include <mutex>

#include "common.h"

namespace {

constexpr int THR = 256;

int grid_for(size_t n) {
  size_t g = (n + THR - 1) / THR;
  if (g > 4096) g = 4096;
  return g < 1 ? 1 : (int)g;
}

// y[n][h][w][0..C1) = x1[row1][.],  y[n][h][w][C1..C1+C2) = x2[row2][.], row_k = n*skn + h*skh + w*skw (a stride of 0
// broadcasts that dimension).
__global__ __launch_bounds__(THR) void concat2_kernel(int N, int H, int W, int C1, int C2,
                                                      const float* __restrict__ x1, long long s1n, long long s1h,
                                                      long long s1w, const float* __restrict__ x2, long long s2n,
                                                      long long s2h, long long s2w, float* __restrict__ y) {
  const int C = C1 + C2;
  const size_t total = (size_t)N * H * W * C;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int c = (int)(i % C);
    size_t r = i / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    y[i] = c < C1 ? x1[(size_t)(n * s1n + h * s1h + w * s1w) * C1 + c]
                  : x2[(size_t)(n * s2n + h * s2h + w * s2w) * C2 + (c - C1)];
  }
}

// Adjoint of concat2 for one of the two sources: gx[row][c] (+)= gy[n][h][w][c0 + c].  A source without broadcast is
// a plain strided copy; a broadcast source is accumulated with atomics into a zeroed gx.
__global__ __launch_bounds__(THR) void concat2_bwd_kernel(int N, int H, int W, int C, int c0, int Cs,
                                                          const float* __restrict__ gy, long long sn, long long sh,
                                                          long long sw, int atomic, float* __restrict__ gx) {
  const size_t total = (size_t)N * H * W * Cs;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int c = (int)(i % Cs);
    size_t r = i / Cs;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    const float g = gy[(((size_t)n * H + h) * W + w) * C + c0 + c];
    float* dst = gx + (size_t)(n * sn + h * sh + w * sw) * Cs + c;
    if (atomic) atomicAdd(dst, g);
    else *dst = g;
  }
}

// y[n][c] = mean over h < wh, w < ww of x[n][h][w][c]
__global__ __launch_bounds__(THR) void window_mean_kernel(int N, int H, int W, int C, int wh, int ww,
                                                          const float* __restrict__ x, float* __restrict__ y) {
  const size_t total = (size_t)N * C;
  const float inv = 1.f / (float)(wh * ww);
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int c = (int)(i % C);
    const int n = (int)(i / C);
    float s = 0.f;
    for (int h = 0; h < wh; ++h)
      for (int w = 0; w < ww; ++w) s += x[(((size_t)n * H + h) * W + w) * C + c];
    y[i] = s * inv;
  }
}

__global__ __launch_bounds__(THR) void window_mean_bwd_kernel(int N, int H, int W, int C, int wh, int ww,
                                                              const float* __restrict__ gy, float* __restrict__ gx) {
  const size_t total = (size_t)N * H * W * C;
  const float inv = 1.f / (float)(wh * ww);
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int c = (int)(i % C);
    size_t r = i / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    gx[i] = (h < wh && w < ww) ? gy[(size_t)n * C + c] * inv : 0.f;
  }
}

// Counter-based generator: a 64-bit mix of (seed, stream position) -> 24 uniform bits.
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.f / 16777216.f);
}

// y = keep ? x / (1-p) : 0; the keep mask is stored for the backward pass.  `step` (device word, may be NULL)
// is mixed into the stream position so that replays of a captured graph draw fresh masks.
__global__ __launch_bounds__(THR) void dropout_kernel(size_t n, const float* __restrict__ x, float p,
                                                      unsigned long long seed, const int* __restrict__ step,
                                                      float* __restrict__ y, unsigned char* __restrict__ mask) {
  const unsigned long long base = (unsigned long long)(step ? *step : 0) << 40;
  const float scale = 1.f / (1.f - p);
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < n; i += (size_t)gridDim.x * THR) {
    const bool keep = uniform01(seed, base + i) >= p;
    mask[i] = keep;
    y[i] = keep ? x[i] * scale : 0.f;
  }
}

// Standard normal draws (Box-Muller on two uniforms of the counter-based generator; one pair per two outputs).
// `step` as in dropout_kernel: replays of a captured graph draw fresh values.
__global__ __launch_bounds__(THR) void randn_kernel(size_t n, unsigned long long seed, const int* __restrict__ step,
                                                    float* __restrict__ y) {
  const unsigned long long base = (unsigned long long)(step ? *step : 0) << 40;
  const size_t pairs = (n + 1) >> 1;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < pairs; i += (size_t)gridDim.x * THR) {
    const float u1 = 1.f - uniform01(seed, base + 2 * i);          // (0, 1]
    const float u2 = uniform01(seed ^ 0xA5A5A5A5A5A5A5A5ull, base + 2 * i + 1);
    const float r = sqrtf(-2.f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    y[2 * i] = r * cs;
    if (2 * i + 1 < n) y[2 * i + 1] = r * sn;
  }
}

__global__ __launch_bounds__(THR) void dropout_bwd_kernel(size_t n, const float* __restrict__ gy,
                                                          const unsigned char* __restrict__ mask, float scale,
                                                          float* __restrict__ gx) {
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < n; i += (size_t)gridDim.x * THR)
    gx[i] = mask[i] ? gy[i] * scale : 0.f;
}

// F.interpolate(x, scale_factor=1/f, mode='bilinear', align_corners=False) for f in {2, 4} on single-channel maps
// (train.py:170-171): the source coordinate of output o is f*o + (f-1)/2, i.e. exactly half way between the pixels
// f*o + f/2 - 1 and f*o + f/2, in both directions: the mean of that 2x2 block.
__global__ __launch_bounds__(THR) void mel_down_kernel(int N, int H, int W, int f, const float* __restrict__ x,
                                                       float* __restrict__ y) {
  const int OH = H / f, OW = W / f;
  const size_t total = (size_t)N * OH * OW;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int ow = (int)(i % OW);
    size_t r = i / OW;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    const int h0 = f * oh + f / 2 - 1, w0 = f * ow + f / 2 - 1;
    const int h1 = min(h0 + 1, H - 1), w1 = min(w0 + 1, W - 1);
    const float* p = x + (size_t)n * H * W;
    y[i] = 0.25f * (p[(size_t)h0 * W + w0] + p[(size_t)h0 * W + w1] + p[(size_t)h1 * W + w0] + p[(size_t)h1 * W + w1]);
  }
}

// dst[c] += sum_p x[p][c]: the bias gradient of a convolution / linear layer from its output gradient, added
// straight into the parameter's gradient.  Two kernels and no atomics: up to 256 workgroups each reduce a strided
// share of the rows into partial[g][C] (float4 columns, rows folded through LDS), then one thread per column sums
// the partials.
template <int VEC>
__global__ __launch_bounds__(THR) void colsum_partial_kernel(int P, int C, const float* __restrict__ x,
                                                             float* __restrict__ partial) {
  __shared__ float red[THR * VEC];
  const int CV = C / VEC;                      // VEC = 4 when the row length allows 16-byte loads, else 1
  for (int c0 = 0; c0 < CV; c0 += THR) {
    const int TX = min(CV - c0, THR), TY = THR / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const bool active = ty < TY;
    float s[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s[k] = 0.f;
    const int col = (c0 + tx) * VEC;
    if (active)
      for (int r = blockIdx.x * TY + ty; r < P; r += gridDim.x * TY) {
        if constexpr (VEC == 4) {
          const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * C + col);
          s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        } else {
          s[0] += x[(size_t)r * C + col];
        }
      }
#pragma unroll
    for (int k = 0; k < VEC; ++k) red[threadIdx.x * VEC + k] = s[k];
    __syncthreads();
    if (active && ty == 0) {
      for (int j = 1; j < TY; ++j)
#pragma unroll
        for (int k = 0; k < VEC; ++k) s[k] += red[(j * TX + tx) * VEC + k];
#pragma unroll
      for (int k = 0; k < VEC; ++k) partial[(size_t)blockIdx.x * C + col + k] = s[k];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(THR) void colsum_fold_add_kernel(int G, int C, const float* __restrict__ partial,
                                                              float* __restrict__ dst) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float t = 0.f;
  if (c < C)
    for (int g = rl; g < G; g += 8) t += partial[(size_t)g * C + c];
  red[rl][cl] = t;
  __syncthreads();
  if (rl == 0 && c < C) {
#pragma unroll
    for (int j = 1; j < 8; ++j) t += red[j][cl];
    dst[c] += t;
  }
}


// A map whose first channels are one feature row repeated over the H axis (the decoder input: phoneme features tiled
// over the 20 frequency rows, generator.py:249-250) never needs the repeated rows materialised: a KxK convolution of
// it is, per kernel row kh, a 1xK convolution of the single row, and output row h is the sum of the kernel rows that
// stay inside the map, 0 <= h + kh - pad < H.
//   expand:  y[b][h][w][c] = (base ? base[b][h][w][c] : 0) + sum_{kh valid for h} p[b][w][kh*C + c]
//   reduce:  gp[b][w][kh*C + c] = sum_{h valid for kh} gy[b][h][w][c]          (its adjoint)
__global__ __launch_bounds__(THR) void hband_expand_kernel(int B, int H, int W, int C, int K, int pad,
                                                           const float* __restrict__ p,
                                                           const float* __restrict__ base, float* __restrict__ y) {
  const int C4 = C >> 2;
  const size_t total = (size_t)B * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int c = (int)(i % C4) * 4;
    size_t r = i / C4;
    const int w = (int)(r % W);
    r /= W;
    const int h = (int)(r % H), b = (int)(r / H);
    float4 acc = base ? *reinterpret_cast<const float4*>(base + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* prow = p + ((size_t)b * W + w) * K * C + c;
    for (int kh = 0; kh < K; ++kh) {
      const int hs = h + kh - pad;
      if (hs < 0 || hs >= H) continue;
      const float4 v = *reinterpret_cast<const float4*>(prow + (size_t)kh * C);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(y + i * 4) = acc;
  }
}

__global__ __launch_bounds__(THR) void hband_reduce_kernel(int B, int H, int W, int C, int K, int pad,
                                                           const float* __restrict__ gy, float* __restrict__ gp) {
  const int C4 = C >> 2;
  const size_t total = (size_t)B * W * K * C4;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int c = (int)(i % C4) * 4;
    size_t r = i / C4;
    const int kh = (int)(r % K);
    r /= K;
    const int w = (int)(r % W), b = (int)(r / W);
    // rows h with 0 <= h + kh - pad < H
    const int h0 = max(0, pad - kh), h1 = min(H, H + pad - kh);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = h0; h < h1; ++h) {
      const float4 v = *reinterpret_cast<const float4*>(gy + (((size_t)b * H + h) * W + w) * C + c);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(gp + i * 4) = acc;
  }
}

}  // namespace

extern "C" int vca_colsum_groups(int P, int C) {
  if (P <= 0 || C <= 0) return -22;
  const int C4 = (C & 3) ? C : C / 4, TX = C4 < THR ? C4 : THR, TY = THR / TX;
  int g = (P + TY * 4 - 1) / (TY * 4);      // at least 4 rows per thread
  if (g > 256) g = 256;
  return g < 1 ? 1 : g;
}

extern "C" int vca_colsum_add_f32(int P, int C, const float* x, float* partial, float* dst, void* stream) {
  VCA_CHECK_ARG(P > 0 && C > 0 && x && partial && dst);
  const int g = vca_colsum_groups(P, C);
  if ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    hipLaunchKernelGGL(colsum_partial_kernel<4>, dim3(g), dim3(THR), 0, (hipStream_t)stream, P, C, x, partial);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<1>, dim3(g), dim3(THR), 0, (hipStream_t)stream, P, C, x, partial);
  hipLaunchKernelGGL(colsum_fold_add_kernel, dim3((C + 31) / 32), dim3(THR), 0, (hipStream_t)stream, g, C, partial,
                     dst);
  return vca_launch_status();
}

// ---- the same column sum for callers that cannot allocate: the partial sums live in a fixed device array of the
// library instead of a caller-supplied workspace.  Every GPU holds its own copy of the array at its own address,
// resolved once per device (with that device current) and used through that pointer only.
namespace {
constexpr int NOALLOC_GROUPS = 256;                        // vca_colsum_groups never returns more
constexpr int NOALLOC_MAX_C = 4096;
constexpr int NOALLOC_MAX_DEV = 64;
static __device__ __align__(256) float colsum_scratch[NOALLOC_GROUPS * NOALLOC_MAX_C];   // 4 MiB per GPU
float* scratch_ptr[NOALLOC_MAX_DEV] = {nullptr};
std::mutex scratch_mutex;

int scratch_for_current_device(float** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= NOALLOC_MAX_DEV) return -22;
  std::lock_guard<std::mutex> lock(scratch_mutex);
  if (!scratch_ptr[dev]) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(colsum_scratch));
    if (e != hipSuccess) return (int)e;
    if (!p) return -14;
    scratch_ptr[dev] = static_cast<float*>(p);
  }
  *out = scratch_ptr[dev];
  return 0;
}
}  // namespace

extern "C" int vca_noalloc_init(void) {
  float* p = nullptr;
  return scratch_for_current_device(&p);
}

extern "C" long long vca_colsum_noalloc_max_floats(void) { return (long long)NOALLOC_GROUPS * NOALLOC_MAX_C; }

extern "C" int vca_colsum_add_noalloc_f32(int P, int C, const float* x, float* dst, void* stream) {
  VCA_CHECK_ARG(P > 0 && C > 0 && x && dst);
  const int g = vca_colsum_groups(P, C);
  if (g > NOALLOC_GROUPS || (long long)g * C > (long long)NOALLOC_GROUPS * NOALLOC_MAX_C) return VCA_E_SCRATCH_LIMIT;
  float* partial = nullptr;
  const int rc = scratch_for_current_device(&partial);
  if (rc != 0) return rc;
  // colsum_partial_kernel writes every (group, column) entry the fold kernel reads, so whatever the previous call
  // left in the array needs no clearing.
  if ((C & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    hipLaunchKernelGGL(colsum_partial_kernel<4>, dim3(g), dim3(THR), 0, (hipStream_t)stream, P, C, x, partial);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<1>, dim3(g), dim3(THR), 0, (hipStream_t)stream, P, C, x, partial);
  hipLaunchKernelGGL(colsum_fold_add_kernel, dim3((C + 31) / 32), dim3(THR), 0, (hipStream_t)stream, g, C, partial,
                     dst);
  return vca_launch_status();
}

extern "C" int vca_concat2_f32(int N, int H, int W, int C1, int C2, const float* x1, long long s1n, long long s1h,
                               long long s1w, const float* x2, long long s2n, long long s2h, long long s2w, float* y,
                               void* stream) {
  VCA_CHECK_ARG(N > 0 && H > 0 && W > 0 && C1 > 0 && C2 > 0 && x1 && x2 && y);
  hipLaunchKernelGGL(concat2_kernel, dim3(grid_for((size_t)N * H * W * (C1 + C2))), dim3(THR), 0, (hipStream_t)stream,
                     N, H, W, C1, C2, x1, s1n, s1h, s1w, x2, s2n, s2h, s2w, y);
  return vca_launch_status();
}

extern "C" int vca_concat2_bwd_f32(int N, int H, int W, int C, int c0, int Cs, const float* gy, long long sn,
                                   long long sh, long long sw, int accumulate, float* gx, void* stream) {
  VCA_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && c0 >= 0 && Cs > 0 && c0 + Cs <= C && gy && gx);
  hipLaunchKernelGGL(concat2_bwd_kernel, dim3(grid_for((size_t)N * H * W * Cs)), dim3(THR), 0, (hipStream_t)stream, N,
                     H, W, C, c0, Cs, gy, sn, sh, sw, accumulate, gx);
  return vca_launch_status();
}

extern "C" int vca_window_mean_f32(int N, int H, int W, int C, int wh, int ww, const float* x, float* y, void* stream) {
  VCA_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && wh > 0 && wh <= H && ww > 0 && ww <= W && x && y);
  hipLaunchKernelGGL(window_mean_kernel, dim3(grid_for((size_t)N * C)), dim3(THR), 0, (hipStream_t)stream, N, H, W, C,
                     wh, ww, x, y);
  return vca_launch_status();
}

extern "C" int vca_window_mean_bwd_f32(int N, int H, int W, int C, int wh, int ww, const float* gy, float* gx,
                                       void* stream) {
  VCA_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && wh > 0 && wh <= H && ww > 0 && ww <= W && gy && gx);
  hipLaunchKernelGGL(window_mean_bwd_kernel, dim3(grid_for((size_t)N * H * W * C)), dim3(THR), 0, (hipStream_t)stream,
                     N, H, W, C, wh, ww, gy, gx);
  return vca_launch_status();
}

extern "C" int vca_dropout_f32(long long n, const float* x, float p, unsigned long long seed, const int* step_dev,
                               float* y, unsigned char* mask, void* stream) {
  VCA_CHECK_ARG(n > 0 && x && y && mask && p >= 0.f && p < 1.f);
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for((size_t)n)), dim3(THR), 0, (hipStream_t)stream, (size_t)n, x, p,
                     seed, step_dev, y, mask);
  return vca_launch_status();
}

extern "C" int vca_randn_f32(long long n, unsigned long long seed, const int* step_dev, float* y, void* stream) {
  VCA_CHECK_ARG(n > 0 && y);
  hipLaunchKernelGGL(randn_kernel, dim3(grid_for((size_t)(n + 1) / 2)), dim3(THR), 0, (hipStream_t)stream, (size_t)n,
                     seed, step_dev, y);
  return vca_launch_status();
}

extern "C" int vca_dropout_bwd_f32(long long n, const float* gy, const unsigned char* mask, float p, float* gx,
                                   void* stream) {
  VCA_CHECK_ARG(n > 0 && gy && mask && gx && p >= 0.f && p < 1.f);
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for((size_t)n)), dim3(THR), 0, (hipStream_t)stream, (size_t)n, gy,
                     mask, 1.f / (1.f - p), gx);
  return vca_launch_status();
}

extern "C" int vca_mel_down_f32(int N, int H, int W, int factor, const float* x, float* y, void* stream) {
  VCA_CHECK_ARG(N > 0 && (factor == 2 || factor == 4) && H >= factor && W >= factor && x && y);
  hipLaunchKernelGGL(mel_down_kernel, dim3(grid_for((size_t)N * (H / factor) * (W / factor))), dim3(THR), 0,
                     (hipStream_t)stream, N, H, W, factor, x, y);
  return vca_launch_status();
}

extern "C" int vca_hband_expand_f32(int B, int H, int W, int C, int K, int pad, const float* p, const float* base,
                                    float* y, void* stream) {
  VCA_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && K > 0 && pad >= 0 && pad < K && p && y);
  hipLaunchKernelGGL(hband_expand_kernel, dim3(grid_for((size_t)B * H * W * (C / 4))), dim3(THR), 0,
                     (hipStream_t)stream, B, H, W, C, K, pad, p, base, y);
  return vca_launch_status();
}

extern "C" int vca_hband_reduce_f32(int B, int H, int W, int C, int K, int pad, const float* gy, float* gp,
                                    void* stream) {
  VCA_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && K > 0 && pad >= 0 && pad < K && gy && gp);
  hipLaunchKernelGGL(hband_reduce_kernel, dim3(grid_for((size_t)B * W * K * (C / 4))), dim3(THR), 0,
                     (hipStream_t)stream, B, H, W, C, K, pad, gy, gp);
  return vca_launch_status();
}
