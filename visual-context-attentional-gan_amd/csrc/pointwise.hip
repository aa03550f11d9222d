// Per-channel statistics, batch-norm + activation (forward and backward) and pooling on channels-last
// fp32 tensors viewed as a [P][C] matrix (P = all pixels).  All kernels are HBM-bound streaming passes:
// one thread handles 4 consecutive channels (16-byte accesses), a wave covers 256 contiguous floats of a
// row, and the grid is capped at ~2048 workgroups with a grid-stride loop over rows.
#include "common.h"

namespace {

constexpr int THR = 256;
constexpr int BN_SLOTS = 16;   // partial-sum buffers of the batch-norm backward reduction

struct RowMap {
  int tx, ty, TX, TY;  // this thread's float4 column / row slot, and the workgroup's layout
  bool active;
};

// Layout of a 256-thread workgroup over a [rows][C/4] float4 matrix.
__device__ __forceinline__ RowMap row_map(int C4) {
  RowMap m;
  m.TX = C4 < THR ? C4 : THR;
  m.TY = THR / m.TX;
  m.tx = threadIdx.x % m.TX;
  m.ty = threadIdx.x / m.TX;
  m.active = m.ty < m.TY;
  return m;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ float act_fwd(float z, int act, float slope) {
  if (act == VCA_ACT_TANH) return tanhf(z);
  return vca_act(z, act, slope);
}
// derivative of the activation with respect to its input, given the input z (and output y for tanh)
__device__ __forceinline__ float act_grad(float z, int act, float slope) {
  if (act == VCA_ACT_LEAKY || act == VCA_ACT_PRELU) return z > 0.f ? 1.f : slope;
  if (act == VCA_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == VCA_ACT_TANH) {
    const float t = tanhf(z);
    return 1.f - t * t;
  }
  return 1.f;
}

// ---------------------------------------------------------------------------------------------
// sums[0][.] += sums[1..BN_SLOTS)[.]; n = rows * C.  Optionally row r of the total is also added into dst[r]
// (the batch-norm backward hands its beta / gamma / PReLU-slope sums straight to the parameters' gradients).
struct FoldDst { float* d[3]; };
__global__ void bn_fold_slots_kernel(int n, float* __restrict__ sums, int C, FoldDst dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float t = sums[i];
  for (int k = 1; k < BN_SLOTS; ++k) t += sums[(size_t)k * n + i];
  sums[i] = t;
  const int r = i / C;
  if (r < 3 && dst.d[r]) dst.d[r][i - r * C] += t;
}

// ---------------------------------------------------------------------------------------------
// column statistics: out[0..C) = sum (x - pivot), out[C..2C) = sum (x - pivot)^2, pivot = x[0][c]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(THR) void colstats_kernel(int P, int C, const float* __restrict__ x,
                                                       float* __restrict__ out) {
  __shared__ float red[2][THR * 4];
  const int C4 = C >> 2;
  for (int c0 = 0; c0 < C4; c0 += THR) {  // C4 > 256 only for very wide maps
    RowMap m = row_map(min(C4 - c0, THR));
    float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
    if (m.active) {
      const int col = (c0 + m.tx) * 4;
      const float4 pv = ld4(x + col);
      for (int r = blockIdx.x * m.TY + m.ty; r < P; r += gridDim.x * m.TY) {
        float4 v = ld4(x + (size_t)r * C + col);
        v.x -= pv.x; v.y -= pv.y; v.z -= pv.z; v.w -= pv.w;
        s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
        s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
      }
    }
    st4(&red[0][threadIdx.x * 4], s1);
    st4(&red[1][threadIdx.x * 4], s2);
    __syncthreads();
    if (m.active && m.ty == 0) {
      for (int j = 1; j < m.TY; ++j) {
        const float4 a = ld4(&red[0][(j * m.TX + m.tx) * 4]), b = ld4(&red[1][(j * m.TX + m.tx) * 4]);
        s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
        s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
      }
      const int col = (c0 + m.tx) * 4;
      float* dst = out + (size_t)(blockIdx.x % BN_SLOTS) * 2 * C;   // partial buffers: see bn_act_bwd_reduce_kernel
      atomicAdd(dst + col + 0, s1.x); atomicAdd(dst + col + 1, s1.y);
      atomicAdd(dst + col + 2, s1.z); atomicAdd(dst + col + 3, s1.w);
      atomicAdd(dst + C + col + 0, s2.x); atomicAdd(dst + C + col + 1, s2.y);
      atomicAdd(dst + C + col + 2, s2.z); atomicAdd(dst + C + col + 3, s2.w);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// finalize train-mode batch-norm statistics: mean / invstd / scale / shift, running-stat update
// ---------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(int C, float n, const float* __restrict__ sums, int nslots,
                                   const float* __restrict__ pivot,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, float* running_mean, float* running_var,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                   float* __restrict__ scale_out, float* __restrict__ shift_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float t1 = sums[c], t2 = sums[C + c];
  for (int k = 1; k < nslots; ++k) {   // partial buffers of a convolution epilogue
    t1 += sums[(size_t)k * 2 * C + c];
    t2 += sums[(size_t)k * 2 * C + C + c];
  }
  const float m1 = t1 / n;
  float var = t2 / n - m1 * m1;
  var = var > 0.f ? var : 0.f;
  const float mean = m1 + (pivot ? pivot[c] : 0.f);
  const float invstd = rsqrtf(var + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean_out[c] = mean;
  invstd_out[c] = invstd;
  scale_out[c] = g * invstd;
  shift_out[c] = b - mean * g * invstd;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    const float unb = n > 1.f ? var * n / (n - 1.f) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
  }
}

// eval-mode: scale/shift from running statistics
__global__ void bn_eval_coeff_kernel(int C, const float* __restrict__ rm, const float* __restrict__ rv,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                     float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                     float* __restrict__ scale_out, float* __restrict__ shift_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = rsqrtf(rv[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  mean_out[c] = rm[c];
  invstd_out[c] = invstd;
  scale_out[c] = g * invstd;
  shift_out[c] = b - rm[c] * g * invstd;
}

// ---------------------------------------------------------------------------------------------
// y = act(x*scale + shift + res) * oscale            (scale/shift/res optional)
// ---------------------------------------------------------------------------------------------
struct PwArgs {
  int P, C, act;
  float slope, oscale;
  const float* x; const float* scale; const float* shift; const float* alpha; const float* res;
  float* y;
};

__global__ __launch_bounds__(THR) void affine_act_kernel(PwArgs a) {
  const int C4 = a.C >> 2;
  const size_t total = (size_t)a.P * C4;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int col = (int)(i % C4) * 4;
    float4 v = ld4(a.x + i * 4);
    float z[4] = {v.x, v.y, v.z, v.w};
    if (a.scale) {
      const float4 sc = ld4(a.scale + col), sh = ld4(a.shift + col);
      z[0] = z[0] * sc.x + sh.x; z[1] = z[1] * sc.y + sh.y; z[2] = z[2] * sc.z + sh.z; z[3] = z[3] * sc.w + sh.w;
    }
    if (a.res) {
      const float4 r = ld4(a.res + i * 4);
      z[0] += r.x; z[1] += r.y; z[2] += r.z; z[3] += r.w;
    }
    if (a.act) {
      float sl[4] = {a.slope, a.slope, a.slope, a.slope};
      if (a.act == VCA_ACT_PRELU) {
        const float4 al = ld4(a.alpha + col);
        sl[0] = al.x; sl[1] = al.y; sl[2] = al.z; sl[3] = al.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) z[j] = act_fwd(z[j], a.act, sl[j]);
    }
    st4(a.y + i * 4, make_float4(z[0] * a.oscale, z[1] * a.oscale, z[2] * a.oscale, z[3] * a.oscale));
  }
}

// ---------------------------------------------------------------------------------------------
// backward of  y = act(z) * oscale,  z = x*scale + shift + res,  with train-mode batch statistics:
//   pass 1 (reduce): sums[0..C) = sum gz, sums[C..2C) = sum gz * xhat, sums[2C..3C) = sum gy*oscale*min(z,0)
//   pass 2 (apply):  gx = scale * (gz - sum_gz/n - xhat * sum_gzxhat/n)   (train)   or scale*gz (stats fixed)
//                    gres = gz
// ---------------------------------------------------------------------------------------------
struct BwArgs {
  int P, C, act, train;
  float slope, oscale, inv_n;
  const float* gy; const float* x; const float* scale; const float* shift; const float* alpha; const float* res;
  const float* mean; const float* invstd;
  float* sums;   // [3*C]
  float* gx; float* gres;
  const float* gadd;   // optional: added to gx (the gradient of a second consumer of x)
};

__device__ __forceinline__ void bw_point(const BwArgs& a, size_t i, int col, float gz[4], float xh[4], float zneg[4],
                                         float gyo[4]) {
  const float4 v = ld4(a.x + i * 4), g = ld4(a.gy + i * 4);
  float x[4] = {v.x, v.y, v.z, v.w};
  float z[4] = {v.x, v.y, v.z, v.w};
  gyo[0] = g.x * a.oscale; gyo[1] = g.y * a.oscale; gyo[2] = g.z * a.oscale; gyo[3] = g.w * a.oscale;
  if (a.scale) {
    const float4 sc = ld4(a.scale + col), sh = ld4(a.shift + col);
    z[0] = z[0] * sc.x + sh.x; z[1] = z[1] * sc.y + sh.y; z[2] = z[2] * sc.z + sh.z; z[3] = z[3] * sc.w + sh.w;
  }
  if (a.res) {
    const float4 r = ld4(a.res + i * 4);
    z[0] += r.x; z[1] += r.y; z[2] += r.z; z[3] += r.w;
  }
  float sl[4] = {a.slope, a.slope, a.slope, a.slope};
  if (a.act == VCA_ACT_PRELU) {
    const float4 al = ld4(a.alpha + col);
    sl[0] = al.x; sl[1] = al.y; sl[2] = al.z; sl[3] = al.w;
  }
  float mu[4] = {0, 0, 0, 0}, is[4] = {1, 1, 1, 1};
  if (a.mean) {
    const float4 m = ld4(a.mean + col), s = ld4(a.invstd + col);
    mu[0] = m.x; mu[1] = m.y; mu[2] = m.z; mu[3] = m.w;
    is[0] = s.x; is[1] = s.y; is[2] = s.z; is[3] = s.w;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    gz[j] = gyo[j] * act_grad(z[j], a.act, sl[j]);
    xh[j] = (x[j] - mu[j]) * is[j];
    zneg[j] = z[j] < 0.f ? z[j] : 0.f;
  }
}

__global__ __launch_bounds__(THR) void bn_act_bwd_reduce_kernel(BwArgs a) {
  __shared__ float red[3][THR * 4];
  const int C4 = a.C >> 2;
  for (int c0 = 0; c0 < C4; c0 += THR) {
    RowMap m = row_map(min(C4 - c0, THR));
    float s[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    if (m.active) {
      const int col = (c0 + m.tx) * 4;
      for (int r = blockIdx.x * m.TY + m.ty; r < a.P; r += gridDim.x * m.TY) {
        float gz[4], xh[4], zn[4], gyo[4];
        bw_point(a, ((size_t)r * a.C + col) >> 2, col, gz, xh, zn, gyo);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[0][j] += gz[j];
          s[1][j] += gz[j] * xh[j];
          s[2][j] += gyo[j] * zn[j];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) st4(&red[k][threadIdx.x * 4], make_float4(s[k][0], s[k][1], s[k][2], s[k][3]));
    __syncthreads();
    if (m.active && m.ty == 0) {
      const int col = (c0 + m.tx) * 4;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float4 t = make_float4(s[k][0], s[k][1], s[k][2], s[k][3]);
        for (int j = 1; j < m.TY; ++j) {
          const float4 q = ld4(&red[k][(j * m.TX + m.tx) * 4]);
          t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
        }
        if (k < 2 || a.act == VCA_ACT_PRELU) {
          // one of BN_SLOTS partial buffers per workgroup: atomics on one cache line serialise in L2, and a
          // thousand workgroups adding to the same 3*C words would otherwise set the kernel's duration
          float* dst = a.sums + (size_t)(blockIdx.x % BN_SLOTS) * 3 * a.C + k * a.C + col;
          atomicAdd(dst + 0, t.x); atomicAdd(dst + 1, t.y); atomicAdd(dst + 2, t.z); atomicAdd(dst + 3, t.w);
        }
      }
    }
    __syncthreads();
  }
}

// GADD: a.gadd is present.  A template parameter, not a run-time test: the test inside the loop cost the launches
// without it a third of their speed.
template <bool GADD>
__global__ __launch_bounds__(THR) void bn_act_bwd_apply_kernel(BwArgs a) {
  const int C4 = a.C >> 2;
  const size_t total = (size_t)a.P * C4;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int col = (int)(i % C4) * 4;
    float gz[4], xh[4], zn[4], gyo[4];
    bw_point(a, i, col, gz, xh, zn, gyo);
    if (a.gres) st4(a.gres + i * 4, make_float4(gz[0], gz[1], gz[2], gz[3]));
    if (a.gx) {
      float o[4] = {gz[0], gz[1], gz[2], gz[3]};
      if (a.train) {
        const float4 s0 = ld4(a.sums + col), s1 = ld4(a.sums + a.C + col);
        o[0] -= s0.x * a.inv_n + xh[0] * s1.x * a.inv_n;
        o[1] -= s0.y * a.inv_n + xh[1] * s1.y * a.inv_n;
        o[2] -= s0.z * a.inv_n + xh[2] * s1.z * a.inv_n;
        o[3] -= s0.w * a.inv_n + xh[3] * s1.w * a.inv_n;
      }
      if (a.scale) {
        const float4 sc = ld4(a.scale + col);
        o[0] *= sc.x; o[1] *= sc.y; o[2] *= sc.z; o[3] *= sc.w;
      }
      if (GADD) {
        const float4 e = ld4(a.gadd + i * 4);
        o[0] += e.x; o[1] += e.y; o[2] += e.z; o[3] += e.w;
      }
      st4(a.gx + i * 4, make_float4(o[0], o[1], o[2], o[3]));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// pooling on channels-last maps [N][H][W][C]
// ---------------------------------------------------------------------------------------------
// mode 0: 2x2 average (floor)          y[n][h/2][w/2] = mean of the 2x2 block
// mode 1: adjoint of mode 0            y[n][h][w]     = x[n][h/2][w/2] / 4   (zero outside the floored area)
// mode 2: 2x2 block sum                y[n][h/2][w/2] = sum of the 2x2 block  (adjoint of nearest x2 upsample)
// mode 3: nearest x2 upsample          y[n][h][w]     = x[n][h/2][w/2]
__global__ __launch_bounds__(THR) void pool2_kernel(int mode, int N, int Hs, int Ws, int Hl, int Wl, int C,
                                                    const float* __restrict__ x, float* __restrict__ y) {
  // (Hs, Ws) = small map, (Hl, Wl) = large map
  const int C4 = C >> 2;
  const bool down = (mode == 0 || mode == 2);
  const int Ho = down ? Hs : Hl, Wo = down ? Ws : Wl;
  const size_t total = (size_t)N * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int c = (int)(i % C4) * 4;
    size_t r = i / C4;
    const int w = r % Wo; r /= Wo;
    const int h = r % Ho;
    const int n = r / Ho;
    float4 o = make_float4(0, 0, 0, 0);
    if (down) {
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const float4 v = ld4(x + (((size_t)n * Hl + 2 * h + dy) * Wl + 2 * w + dx) * C + c);
          o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
        }
      if (mode == 0) { o.x *= 0.25f; o.y *= 0.25f; o.z *= 0.25f; o.w *= 0.25f; }
    } else {
      const int hs = h >> 1, ws = w >> 1;
      if (hs < Hs && ws < Ws) {
        o = ld4(x + (((size_t)n * Hs + hs) * Ws + ws) * C + c);
        if (mode == 1) { o.x *= 0.25f; o.y *= 0.25f; o.z *= 0.25f; o.w *= 0.25f; }
      }
    }
    st4(y + i * 4, o);
  }
}

// Entry of a down-sampling discriminator block (generator.py:71-83): the block input feeds LeakyReLU (residual branch)
// and a 2x2 average (shortcut).  One pass over the map writes both; the backward pass is one kernel as well,
// gx = ga * leaky'(x) + gp[h/2][w/2] / 4, instead of an activation backward, a pooling adjoint and the add that joins
// them.  One thread per 2x2 block and 4 channels; rows / columns beyond 2*(H/2), 2*(W/2) are not pooled (floor).
__global__ __launch_bounds__(THR) void leaky_pool2_fwd_kernel(int N, int H, int W, int C, float slope,
                                                              const float* __restrict__ x, float* __restrict__ ya,
                                                              float* __restrict__ yp) {
  const int C4 = C >> 2, Hb = (H + 1) >> 1, Wb = (W + 1) >> 1, Hs = H >> 1, Ws = W >> 1;
  const size_t total = (size_t)N * Hb * Wb * C4;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int c = (int)(i % C4) * 4;
    size_t r = i / C4;
    const int wb = r % Wb; r /= Wb;
    const int hb = r % Hb;
    const int n = r / Hb;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int h = 2 * hb + dy, w = 2 * wb + dx;
        if (h < H && w < W) {
          const size_t o = (((size_t)n * H + h) * W + w) * C + c;
          const float4 v = ld4(x + o);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
          st4(ya + o, make_float4(v.x > 0.f ? v.x : v.x * slope, v.y > 0.f ? v.y : v.y * slope,
                                  v.z > 0.f ? v.z : v.z * slope, v.w > 0.f ? v.w : v.w * slope));
        }
      }
    if (hb < Hs && wb < Ws)
      st4(yp + (((size_t)n * Hs + hb) * Ws + wb) * C + c,
          make_float4(acc.x * 0.25f, acc.y * 0.25f, acc.z * 0.25f, acc.w * 0.25f));
  }
}

__global__ __launch_bounds__(THR) void leaky_pool2_bwd_kernel(int N, int H, int W, int C, float slope,
                                                              const float* __restrict__ ga,
                                                              const float* __restrict__ gp,
                                                              const float* __restrict__ x, float* __restrict__ gx) {
  const int C4 = C >> 2, Hb = (H + 1) >> 1, Wb = (W + 1) >> 1, Hs = H >> 1, Ws = W >> 1;
  const size_t total = (size_t)N * Hb * Wb * C4;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int c = (int)(i % C4) * 4;
    size_t r = i / C4;
    const int wb = r % Wb; r /= Wb;
    const int hb = r % Hb;
    const int n = r / Hb;
    float4 p = make_float4(0, 0, 0, 0);
    if (gp && hb < Hs && wb < Ws) {
      p = ld4(gp + (((size_t)n * Hs + hb) * Ws + wb) * C + c);
      p.x *= 0.25f; p.y *= 0.25f; p.z *= 0.25f; p.w *= 0.25f;
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int h = 2 * hb + dy, w = 2 * wb + dx;
        if (h < H && w < W) {
          const size_t o = (((size_t)n * H + h) * W + w) * C + c;
          float4 out = p;
          if (ga) {
            const float4 v = ld4(x + o), g = ld4(ga + o);
            out.x += v.x > 0.f ? g.x : g.x * slope;
            out.y += v.y > 0.f ? g.y : g.y * slope;
            out.z += v.z > 0.f ? g.z : g.z * slope;
            out.w += v.w > 0.f ? g.w : g.w * slope;
          }
          st4(gx + o, out);
        }
      }
  }
}

// 3x3 stride-2 pad-1 max pool, forward (also writes the argmax tap 0..8) and backward (scatter by gather)
__global__ __launch_bounds__(THR) void maxpool3_fwd_kernel(int N, int H, int W, int OH, int OW, int C,
                                                           const float* __restrict__ x, float* __restrict__ y,
                                                           unsigned char* __restrict__ idx) {
  const int C4 = C >> 2;
  const size_t total = (size_t)N * OH * OW * C4;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int c = (int)(i % C4) * 4;
    size_t r = i / C4;
    const int ow = r % OW; r /= OW;
    const int oh = r % OH;
    const int n = r / OH;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ih = oh * 2 - 1 + t / 3, iw = ow * 2 - 1 + t % 3;
      if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
      const float4 v = ld4(x + (((size_t)n * H + ih) * W + iw) * C + c);
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (vv[j] > best[j] || vv[j] != vv[j]) { best[j] = vv[j]; bi[j] = t; }
    }
    st4(y + i * 4, make_float4(best[0], best[1], best[2], best[3]));
    *reinterpret_cast<uchar4*>(idx + i * 4) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
  }
}

// y = maxpool3x3s2(act(x*scale + shift)) in one pass over x (visual_front.py:12-14: BatchNorm3d -> PReLU ->
// MaxPool3d((1,3,3),(1,2,2),(0,1,1))): the activated full-resolution map is never written.
// Neighbouring windows share input rows and columns.  Workgroups are dealt round-robin over the 8 XCDs, each with an
// L2 of its own, so the work is laid out per image: all workgroups of an image have the same blockIdx % 8 and its
// shared rows are fetched into one L2 once instead of into several.
__global__ __launch_bounds__(THR) void affine_act_pool3_kernel(int N, int H, int W, int OH, int OW, int C, int act,
                                                               float slope, const float* __restrict__ x,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift,
                                                               const float* __restrict__ alpha,
                                                               float* __restrict__ y, unsigned char* __restrict__ idx,
                                                               int wg_per_img) {
  const int C4 = C >> 2;
  const int per_img = OH * OW * C4;
  const int xcd = blockIdx.x & 7, njobs = ((N + 7) >> 3) * wg_per_img;
  for (int job = blockIdx.x >> 3; job < njobs; job += gridDim.x >> 3) {
    const int n = (job / wg_per_img) * 8 + xcd;
    const int r0 = (job % wg_per_img) * THR + threadIdx.x;
    if (n >= N || r0 >= per_img) continue;
    const size_t i = (size_t)n * per_img + r0;
    const int c = (r0 % C4) * 4;
    int r = r0 / C4;
    const int ow = r % OW;
    const int oh = r / OW;
    const float4 sc = ld4(scale + c), sh = ld4(shift + c);
    float sl[4] = {slope, slope, slope, slope};
    if (act == VCA_ACT_PRELU) {
      const float4 al = ld4(alpha + c);
      sl[0] = al.x; sl[1] = al.y; sl[2] = al.z; sl[3] = al.w;
    }
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ih = oh * 2 - 1 + t / 3, iw = ow * 2 - 1 + t % 3;
      if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
      const float4 v = ld4(x + (((size_t)n * H + ih) * W + iw) * C + c);
      const float vv[4] = {act_fwd(v.x * sc.x + sh.x, act, sl[0]), act_fwd(v.y * sc.y + sh.y, act, sl[1]),
                           act_fwd(v.z * sc.z + sh.z, act, sl[2]), act_fwd(v.w * sc.w + sh.w, act, sl[3])};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (vv[j] > best[j] || vv[j] != vv[j]) { best[j] = vv[j]; bi[j] = t; }
    }
    st4(y + i * 4, make_float4(best[0], best[1], best[2], best[3]));
    *reinterpret_cast<uchar4*>(idx + i * 4) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
  }
}

__global__ __launch_bounds__(THR) void maxpool3_bwd_kernel(int N, int H, int W, int OH, int OW, int C,
                                                           const float* __restrict__ gy,
                                                           const unsigned char* __restrict__ idx,
                                                           float* __restrict__ gx) {
  // One thread per 2x2 block of input pixels (rows 2a, 2a+1; columns 2b, 2b+1) and 4 channels: with stride 2 the
  // block is touched by exactly the windows (a | a+1, b | b+1), so each window's gradient and argmax are loaded
  // once for the four pixels (window (oh, ow) covers rows 2oh-1..2oh+1, tap = 3*(ih-2oh+1) + (iw-2ow+1)).
  const int C4 = C >> 2;
  const int HB = (H + 1) >> 1, WB = (W + 1) >> 1;
  const size_t total = (size_t)N * HB * WB * C4;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int c = (int)(i % C4) * 4;
    size_t r = i / C4;
    const int b = r % WB; r /= WB;
    const int a = r % HB;
    const int n = r / HB;
    float o[2][2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[p][q][e] = 0.f;
#pragma unroll
    for (int wy = 0; wy < 2; ++wy) {
      const int oh = a + wy;
      if (oh >= OH) continue;
#pragma unroll
      for (int wx = 0; wx < 2; ++wx) {
        const int ow = b + wx;
        if (ow >= OW) continue;
        const size_t oi = (((size_t)n * OH + oh) * OW + ow) * C + c;
        const uchar4 k4 = *reinterpret_cast<const uchar4*>(idx + oi);
        const float4 g4 = ld4(gy + oi);
        const int k[4] = {k4.x, k4.y, k4.z, k4.w};
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int dy = p + 1 - 2 * wy;          // row 2a+p inside window a+wy
          if (dy < 0 || dy > 2) continue;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int dx = q + 1 - 2 * wx;
            if (dx < 0 || dx > 2) continue;
            const int tap = dy * 3 + dx;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (k[e] == tap) o[p][q][e] += g[e];
          }
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int ih = 2 * a + p;
      if (ih >= H) continue;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int iw = 2 * b + q;
        if (iw >= W) continue;
        st4(gx + (((size_t)n * H + ih) * W + iw) * C + c, make_float4(o[p][q][0], o[p][q][1], o[p][q][2], o[p][q][3]));
      }
    }
  }
}

int grid_for(size_t work_items) {
  size_t g = (work_items + THR - 1) / THR;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int vca_colstats_f32(int P, int C, const float* x, float* sums, void* stream) {
  VCA_CHECK_ARG(P > 0 && C > 0 && (C % 4) == 0 && x && sums);
  const int C4 = C / 4, TX = C4 < THR ? C4 : THR, TY = THR / TX;
  int g = (P + TY - 1) / TY;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(colstats_kernel, dim3(g), dim3(THR), 0, (hipStream_t)stream, P, C, x, sums);
  hipLaunchKernelGGL(bn_fold_slots_kernel, dim3((2 * C + 127) / 128), dim3(128), 0, (hipStream_t)stream, 2 * C, sums,
                     C, FoldDst{{nullptr, nullptr, nullptr}});
  return vca_launch_status();
}

extern "C" int vca_bn_finalize_f32(int C, float count, const float* sums, int nslots, const float* pivot,
                                   const float* gamma, const float* beta, float eps, float momentum,
                                   float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                                   float* shift, void* stream) {
  VCA_CHECK_ARG(C > 0 && count > 0 && sums && nslots >= 1 && mean && invstd && scale && shift);
  VCA_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, C, count, sums,
                     nslots, pivot, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  return vca_launch_status();
}

extern "C" int vca_bn_eval_coeff_f32(int C, const float* running_mean, const float* running_var, const float* gamma,
                                     const float* beta, float eps, float* mean, float* invstd, float* scale,
                                     float* shift, void* stream) {
  VCA_CHECK_ARG(C > 0 && running_mean && running_var && mean && invstd && scale && shift);
  hipLaunchKernelGGL(bn_eval_coeff_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, C, running_mean,
                     running_var, gamma, beta, eps, mean, invstd, scale, shift);
  return vca_launch_status();
}

extern "C" int vca_affine_act_f32(int P, int C, const float* x, const float* scale, const float* shift,
                                  const float* alpha, const float* res, int act, float slope, float oscale, float* y,
                                  void* stream) {
  VCA_CHECK_ARG(P > 0 && C > 0 && (C % 4) == 0 && x && y && act >= 0 && act <= 4);
  VCA_CHECK_ARG((scale == nullptr) == (shift == nullptr));
  VCA_CHECK_ARG(act != VCA_ACT_PRELU || alpha);
  PwArgs a{P, C, act, slope, oscale, x, scale, shift, alpha, res, y};
  hipLaunchKernelGGL(affine_act_kernel, dim3(grid_for((size_t)P * (C / 4))), dim3(THR), 0, (hipStream_t)stream, a);
  return vca_launch_status();
}

extern "C" int vca_affine_act_bwd2_f32(int P, int C, const float* gy, const float* x, const float* scale,
                                       const float* shift, const float* alpha, const float* res, const float* mean,
                                       const float* invstd, int act, float slope, float oscale, int train,
                                       float* sums, float* gx, float* gres, const float* gadd, float* gbeta_acc,
                                       float* ggamma_acc, float* galpha_acc, void* stream) {
  VCA_CHECK_ARG(P > 0 && C > 0 && (C % 4) == 0 && gy && x && act >= 0 && act <= 4);
  VCA_CHECK_ARG((scale == nullptr) == (shift == nullptr) && (mean == nullptr) == (invstd == nullptr));
  VCA_CHECK_ARG(act != VCA_ACT_PRELU || alpha);
  VCA_CHECK_ARG(!train || (sums && mean && scale));
  VCA_CHECK_ARG(act != VCA_ACT_PRELU || sums);
  VCA_CHECK_ARG(!gadd || gx);
  VCA_CHECK_ARG(sums || !(gbeta_acc || ggamma_acc || galpha_acc));
  VCA_CHECK_ARG(!galpha_acc || act == VCA_ACT_PRELU);
  BwArgs a{P, C, act, train, slope, oscale, 1.0f / (float)P, gy, x, scale, shift, alpha, res, mean, invstd, sums, gx, gres,
           gadd};
  hipStream_t s = (hipStream_t)stream;
  if (sums) {
    const int C4 = C / 4, TX = C4 < THR ? C4 : THR, TY = THR / TX;
    int g = (P + TY - 1) / TY;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(g), dim3(THR), 0, s, a);
    hipLaunchKernelGGL(bn_fold_slots_kernel, dim3((3 * C + 127) / 128), dim3(128), 0, s, 3 * C, sums, C,
                       FoldDst{{gbeta_acc, ggamma_acc, galpha_acc}});
  }
  if (gx || gres) {
    const dim3 grid(grid_for((size_t)P * (C / 4)));
    if (gadd) hipLaunchKernelGGL(bn_act_bwd_apply_kernel<true>, grid, dim3(THR), 0, s, a);
    else hipLaunchKernelGGL(bn_act_bwd_apply_kernel<false>, grid, dim3(THR), 0, s, a);
  }
  return vca_launch_status();
}

extern "C" int vca_affine_act_bwd_f32(int P, int C, const float* gy, const float* x, const float* scale,
                                      const float* shift, const float* alpha, const float* res, const float* mean,
                                      const float* invstd, int act, float slope, float oscale, int train,
                                      float* sums, float* gx, float* gres, void* stream) {
  return vca_affine_act_bwd2_f32(P, C, gy, x, scale, shift, alpha, res, mean, invstd, act, slope, oscale, train, sums,
                                 gx, gres, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int vca_pool2_f32(int mode, int N, int Hs, int Ws, int Hl, int Wl, int C, const float* x, float* y,
                             void* stream) {
  VCA_CHECK_ARG(mode >= 0 && mode <= 3 && N > 0 && Hs > 0 && Ws > 0 && C > 0 && (C % 4) == 0 && x && y);
  VCA_CHECK_ARG(Hl / 2 == Hs && Wl / 2 == Ws);
  const bool down = (mode == 0 || mode == 2);
  const size_t total = (size_t)N * (down ? Hs : Hl) * (down ? Ws : Wl) * (C / 4);
  hipLaunchKernelGGL(pool2_kernel, dim3(grid_for(total)), dim3(THR), 0, (hipStream_t)stream, mode, N, Hs, Ws, Hl, Wl,
                     C, x, y);
  return vca_launch_status();
}

extern "C" int vca_leaky_pool2_fwd_f32(int N, int H, int W, int C, float slope, const float* x, float* y_act,
                                       float* y_pool, void* stream) {
  VCA_CHECK_ARG(N > 0 && H > 1 && W > 1 && C > 0 && (C % 4) == 0 && x && y_act && y_pool);
  hipLaunchKernelGGL(leaky_pool2_fwd_kernel, dim3(grid_for((size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4))),
                     dim3(THR), 0, (hipStream_t)stream, N, H, W, C, slope, x, y_act, y_pool);
  return vca_launch_status();
}

extern "C" int vca_leaky_pool2_bwd_f32(int N, int H, int W, int C, float slope, const float* g_act,
                                       const float* g_pool, const float* x, float* gx, void* stream) {
  VCA_CHECK_ARG(N > 0 && H > 1 && W > 1 && C > 0 && (C % 4) == 0 && x && gx && (g_act || g_pool));
  hipLaunchKernelGGL(leaky_pool2_bwd_kernel, dim3(grid_for((size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4))),
                     dim3(THR), 0, (hipStream_t)stream, N, H, W, C, slope, g_act, g_pool, x, gx);
  return vca_launch_status();
}

extern "C" int vca_maxpool3s2_fwd_f32(int N, int H, int W, int C, const float* x, float* y, unsigned char* idx,
                                      void* stream) {
  VCA_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0 && x && y && idx);
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3_fwd_kernel, dim3(grid_for((size_t)N * OH * OW * (C / 4))), dim3(THR), 0,
                     (hipStream_t)stream, N, H, W, OH, OW, C, x, y, idx);
  return vca_launch_status();
}

extern "C" int vca_maxpool3s2_bwd_f32(int N, int H, int W, int C, const float* gy, const unsigned char* idx,
                                      float* gx, void* stream) {
  VCA_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0 && gy && gx && idx);
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3_bwd_kernel, dim3(grid_for((size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4))), dim3(THR), 0,
                     (hipStream_t)stream, N, H, W, OH, OW, C, gy, idx, gx);
  return vca_launch_status();
}

extern "C" int vca_affine_act_pool3_fwd_f32(int N, int H, int W, int C, const float* x, const float* scale,
                                            const float* shift, const float* alpha, int act, float slope, float* y,
                                            unsigned char* idx, void* stream) {
  VCA_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0 && x && scale && shift && y && idx);
  VCA_CHECK_ARG(act >= 0 && act <= 4 && (act != VCA_ACT_PRELU || alpha));
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  VCA_CHECK_ARG((long long)OH * OW * (C / 4) < (1ll << 30));
  const int wg_per_img = (OH * OW * (C / 4) + THR - 1) / THR;
  long long g8 = (long long)((N + 7) / 8) * wg_per_img;      // jobs per XCD lane
  if (g8 > 256) g8 = 256;
  hipLaunchKernelGGL(affine_act_pool3_kernel, dim3((unsigned)(8 * g8)), dim3(THR), 0, (hipStream_t)stream, N, H, W, OH,
                     OW, C, act, slope, x, scale, shift, alpha, y, idx, wg_per_img);
  return vca_launch_status();
}
