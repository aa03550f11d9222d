// Loss heads of the sync discriminator (generator.py:347-361) on the (B, T, C) visual / audio embeddings.
//
// Discriminator mode: rows are L2-normalised, S = vn . an^T / temp is a T x T matrix per sample, and the loss is the
// symmetric InfoNCE over time steps  -1/2 * mean_i [ log_softmax_row(S)_ii + log_softmax_col(S)_ii ].
// Generator mode: 5 - mean_t |cos(v_t, a_t)|.
// One workgroup per sample: T <= 160 keeps S (and its gradient) in LDS.
#include "common.h"

namespace {

constexpr int THR = 256;
constexpr int KC = 32;        // reduction chunk staged in LDS
constexpr int MAXT = 160;
constexpr int MAXTILE = 7;    // 4x4 output tiles per thread: ceil(40*40 / 256)

__device__ __forceinline__ float inv_norm(float sq, float eps) { return 1.f / fmaxf(sqrtf(sq), eps); }

// stats[b][0..T) = 1/|v_i|, [T..2T) = 1/|a_i|, [2T..3T) = row log-sum-exp, [3T..4T) = column log-sum-exp
__global__ __launch_bounds__(THR) void sync_nce_fwd_kernel(int T, int C, const float* __restrict__ v,
                                                           const float* __restrict__ a, float inv_temp,
                                                           float* __restrict__ S_out, float* __restrict__ stats,
                                                           float* __restrict__ loss) {
  extern __shared__ float sm[];
  float* S = sm;                       // [T][T]
  float* vt = S + T * T;               // [T][KC+1]
  float* at = vt + T * (KC + 1);       // [T][KC+1]
  float* inv_v = at + T * (KC + 1);    // [T]
  float* inv_a = inv_v + T;            // [T]
  float* red = inv_a + T;              // [THR/64]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* vb = v + (size_t)b * T * C;
  const float* ab = a + (size_t)b * T * C;

  for (int r = wave; r < 2 * T; r += THR / 64) {   // squared norms: one wave per row
    const float* row = r < T ? vb + (size_t)r * C : ab + (size_t)(r - T) * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += row[c] * row[c];
    s = wave_sum(s);
    if (lane == 0) (r < T ? inv_v : inv_a)[r < T ? r : r - T] = inv_norm(s, 1e-12f);
  }

  const int nb = (T + 3) >> 2, ntile = nb * nb;
  float acc[MAXTILE][16];
#pragma unroll
  for (int t = 0; t < MAXTILE; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  for (int k0 = 0; k0 < C; k0 += KC) {
    __syncthreads();
    for (int i = tid; i < T * KC; i += THR) {
      const int r = i / KC, k = i - r * KC;
      const bool ok = k0 + k < C;
      vt[r * (KC + 1) + k] = ok ? vb[(size_t)r * C + k0 + k] : 0.f;
      at[r * (KC + 1) + k] = ok ? ab[(size_t)r * C + k0 + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < MAXTILE; ++t) {
      const int tile = tid + t * THR;
      if (tile < ntile) {
        const int ti = tile / nb, tj = tile - ti * nb;
        for (int k = 0; k < KC; ++k) {
          float x[4], y[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            x[q] = vt[min(ti * 4 + q, T - 1) * (KC + 1) + k];
            y[q] = at[min(tj * 4 + q, T - 1) * (KC + 1) + k];
          }
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[t][p * 4 + q] += x[p] * y[q];
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MAXTILE; ++t) {
    const int tile = tid + t * THR;
    if (tile < ntile) {
      const int ti = tile / nb, tj = tile - ti * nb;
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = ti * 4 + p, j = tj * 4 + q;
          if (i < T && j < T) {
            const float s = acc[t][p * 4 + q] * inv_v[i] * inv_a[j] * inv_temp;
            S[i * T + j] = s;
            S_out[((size_t)b * T + i) * T + j] = s;
          }
        }
    }
  }
  __syncthreads();
  // log-sum-exp of every row and column, then the diagonal terms
  float part = 0.f;
  float* st = stats + (size_t)b * 4 * T;
  for (int r = tid; r < 2 * T; r += THR) {
    const bool rowwise = r < T;
    const int i = rowwise ? r : r - T;
    float m = -INFINITY;
    for (int j = 0; j < T; ++j) m = fmaxf(m, rowwise ? S[i * T + j] : S[j * T + i]);
    float e = 0.f;
    for (int j = 0; j < T; ++j) e += expf((rowwise ? S[i * T + j] : S[j * T + i]) - m);
    const float lse = m + logf(e);
    st[(rowwise ? 2 : 3) * T + i] = lse;
    part += S[i * T + i] - lse;
  }
  for (int i = tid; i < T; i += THR) {
    st[i] = inv_v[i];
    st[T + i] = inv_a[i];
  }
  part = wave_sum(part);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
    for (int w = 0; w < THR / 64; ++w) tot += red[w];
    loss[b] = -0.5f * tot / (float)T;
  }
}

// dv, da from gloss[b]: G_ij = dL/dS_ij / temp = gloss * (-1/(2T)) / temp * (2 d_ij - softmax_row_ij - softmax_col_ij);
// d vn_i = sum_j G_ij an_j, and through the normalisation dv_i = (d vn_i - vn_i (vn_i . d vn_i)) / |v_i| with
// vn_i . d vn_i = temp * sum_j G_ij S_ij.  The same with the roles swapped for da.
__global__ __launch_bounds__(THR) void sync_nce_bwd_kernel(int T, int C, const float* __restrict__ v,
                                                           const float* __restrict__ a,
                                                           const float* __restrict__ S_in,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gloss, float inv_temp,
                                                           float* __restrict__ dv, float* __restrict__ da) {
  extern __shared__ float sm[];
  float* G = sm;                 // [T][T]
  float* rowdot = G + T * T;     // [T]
  float* coldot = rowdot + T;    // [T]
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* st = stats + (size_t)b * 4 * T;
  const float* inv_v = st;
  const float* inv_a = st + T;
  const float* Sb = S_in + (size_t)b * T * T;
  const float gs = gloss[b] * (-0.5f / (float)T) * inv_temp;
  for (int i = tid; i < T * T; i += THR) {
    const int r = i / T, c = i - r * T;
    const float s = Sb[i];
    G[i] = gs * ((r == c ? 2.f : 0.f) - expf(s - st[2 * T + r]) - expf(s - st[3 * T + c]));
  }
  __syncthreads();
  const float temp = 1.f / inv_temp;
  for (int r = tid; r < 2 * T; r += THR) {
    const bool rowwise = r < T;
    const int i = rowwise ? r : r - T;
    float d = 0.f;
    for (int j = 0; j < T; ++j) d += rowwise ? G[i * T + j] * Sb[i * T + j] : G[j * T + i] * Sb[j * T + i];
    (rowwise ? rowdot : coldot)[i] = d * temp;
  }
  __syncthreads();
  const float* vb = v + (size_t)b * T * C;
  const float* ab = a + (size_t)b * T * C;
  for (int c = tid; c < C; c += THR) {
    for (int i0 = 0; i0 < T; i0 += 8) {
      float accv[8], acca[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) accv[p] = acca[p] = 0.f;
      for (int j = 0; j < T; ++j) {
        const float an = ab[(size_t)j * C + c] * inv_a[j];
        const float vn = vb[(size_t)j * C + c] * inv_v[j];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int i = min(i0 + p, T - 1);
          accv[p] += G[i * T + j] * an;     // d vn_i  = sum_j G_ij an_j
          acca[p] += G[j * T + i] * vn;     // d an_i  = sum_j G_ji vn_j
        }
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int i = i0 + p;
        if (i < T) {
          const size_t o = ((size_t)b * T + i) * C + c;
          if (dv) dv[o] = inv_v[i] * (accv[p] - vb[(size_t)i * C + c] * inv_v[i] * rowdot[i]);
          if (da) da[o] = inv_a[i] * (acca[p] - ab[(size_t)i * C + c] * inv_a[i] * coldot[i]);
        }
      }
    }
  }
}

// loss[b] = 5 - mean_t |cos(v_t, a_t)|; cosv[b][t] keeps the cosine for the backward pass
__global__ __launch_bounds__(THR) void sync_cos_fwd_kernel(int T, int C, const float* __restrict__ v,
                                                           const float* __restrict__ a, float* __restrict__ cosv,
                                                           float* __restrict__ loss) {
  __shared__ float red[THR / 64];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float part = 0.f;
  for (int t = wave; t < T; t += THR / 64) {
    const float* x = v + ((size_t)b * T + t) * C;
    const float* y = a + ((size_t)b * T + t) * C;
    float d = 0.f, nx = 0.f, ny = 0.f;
    for (int c = lane; c < C; c += 64) {
      d += x[c] * y[c];
      nx += x[c] * x[c];
      ny += y[c] * y[c];
    }
    d = wave_sum(d); nx = wave_sum(nx); ny = wave_sum(ny);
    const float cs = d * inv_norm(nx, 1e-8f) * inv_norm(ny, 1e-8f);
    if (lane == 0) cosv[(size_t)b * T + t] = cs;
    part += fabsf(cs);
  }
  if (lane == 0) red[wave] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w = 0; w < THR / 64; ++w) tot += red[w];
    loss[b] = 5.f - tot / (float)T;
  }
}

// d|cos|/da = sign(cos) * (v / (|v||a|) - cos * a / |a|^2), scaled by gloss[b] * (-1/T); dv likewise
__global__ __launch_bounds__(THR) void sync_cos_bwd_kernel(int T, int C, const float* __restrict__ v,
                                                           const float* __restrict__ a,
                                                           const float* __restrict__ cosv,
                                                           const float* __restrict__ gloss, float* __restrict__ dv,
                                                           float* __restrict__ da) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = wave; t < T; t += THR / 64) {
    const size_t o = ((size_t)b * T + t) * C;
    const float* x = v + o;
    const float* y = a + o;
    float nx = 0.f, ny = 0.f;
    for (int c = lane; c < C; c += 64) {
      nx += x[c] * x[c];
      ny += y[c] * y[c];
    }
    nx = wave_sum(nx); ny = wave_sum(ny);
    const float ix = inv_norm(nx, 1e-8f), iy = inv_norm(ny, 1e-8f);
    const float cs = cosv[(size_t)b * T + t];
    const float g = gloss[b] * (-1.f / (float)T) * (cs > 0.f ? 1.f : (cs < 0.f ? -1.f : 0.f));
    for (int c = lane; c < C; c += 64) {
      if (da) da[o + c] = g * (x[c] * ix * iy - cs * y[c] * iy * iy);
      if (dv) dv[o + c] = g * (y[c] * ix * iy - cs * x[c] * ix * ix);
    }
  }
}

}  // namespace

extern "C" int vca_sync_nce_fwd_f32(int B, int T, int C, const float* v, const float* a, float temp, float* S,
                                    float* stats, float* loss, void* stream) {
  VCA_CHECK_ARG(B > 0 && T > 0 && T <= MAXT && C > 0 && v && a && S && stats && loss && temp > 0.f);
  const size_t lds = ((size_t)T * T + 2 * (size_t)T * (KC + 1) + 2 * T + THR / 64) * sizeof(float);
  VCA_MAX_LDS(sync_nce_fwd_kernel);
  hipLaunchKernelGGL(sync_nce_fwd_kernel, dim3(B), dim3(THR), lds, (hipStream_t)stream, T, C, v, a, 1.f / temp, S,
                     stats, loss);
  return vca_launch_status();
}

extern "C" int vca_sync_nce_bwd_f32(int B, int T, int C, const float* v, const float* a, const float* S,
                                    const float* stats, const float* gloss, float temp, float* dv, float* da,
                                    void* stream) {
  VCA_CHECK_ARG(B > 0 && T > 0 && T <= MAXT && C > 0 && v && a && S && stats && gloss && temp > 0.f && (dv || da));
  const size_t lds = ((size_t)T * T + 2 * T) * sizeof(float);
  VCA_MAX_LDS(sync_nce_bwd_kernel);
  hipLaunchKernelGGL(sync_nce_bwd_kernel, dim3(B), dim3(THR), lds, (hipStream_t)stream, T, C, v, a, S, stats, gloss,
                     1.f / temp, dv, da);
  return vca_launch_status();
}

extern "C" int vca_sync_cos_fwd_f32(int B, int T, int C, const float* v, const float* a, float* cosv, float* loss,
                                    void* stream) {
  VCA_CHECK_ARG(B > 0 && T > 0 && C > 0 && v && a && cosv && loss);
  hipLaunchKernelGGL(sync_cos_fwd_kernel, dim3(B), dim3(THR), 0, (hipStream_t)stream, T, C, v, a, cosv, loss);
  return vca_launch_status();
}

extern "C" int vca_sync_cos_bwd_f32(int B, int T, int C, const float* v, const float* a, const float* cosv,
                                    const float* gloss, float* dv, float* da, void* stream) {
  VCA_CHECK_ARG(B > 0 && T > 0 && C > 0 && v && a && cosv && gloss && (dv || da));
  hipLaunchKernelGGL(sync_cos_bwd_kernel, dim3(B), dim3(THR), 0, (hipStream_t)stream, T, C, v, a, cosv, gloss, dv, da);
  return vca_launch_status();
}
