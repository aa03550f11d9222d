// Visual Context Attention core (generator.py:161-167): ctx = softmax(mask(q k^T * scale)) v, fused per
// (batch sample, 32-query tile) workgroup, forward and backward.
//
// All three contractions run on the matrix cores (exact fp32 v_mfma_f32_32x32x2_f32).  Operands are staged in
// LDS in [reduction][row] order so that one lane's operand (A[row = lane&31][k = lane>>5]) is a conflict-free
// ds_read_b32; the score tile never leaves the CU: it is written to LDS, masked with the sample's valid length,
// soft-maxed with wave-level shuffles (8 rows per wave, 64 keys per pass) and re-read, transposed, as the A
// operand of the value contraction.  The probabilities are also stored (B, Tq, S) for the backward kernel, which
// recomputes nothing and adds the key / value gradients of its query tile with fp32 atomics.
#include "common.h"

namespace {

constexpr int DH = 256;     // projection width (AVAttention(256), generator.py:203,205)
constexpr int QT = 32;      // query rows per workgroup
constexpr int LQ = QT + 1;  // padded row length of [k][query] images
constexpr int LD = DH + 4;  // padded row length of [row][d] images
constexpr int MAXS = 160;   // keys (video frames): 75 for GRID, up to 150 for LRS

// C(32x32) += sum_{k<K} A[k][m0 + i] * B[k][n0 + j];  A, B in LDS with leading dimensions lda, ldb
__device__ __forceinline__ void mma32(f32x16& acc, const float* A, int lda, int m0, const float* B, int ldb, int n0,
                                      int K, int lane) {
  const int lr = lane & 31, lk = lane >> 5;
  for (int k = 0; k < K; k += 2) {
    const float a = A[(k + lk) * lda + m0 + lr];
    const float b = B[(k + lk) * ldb + n0 + lr];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
}

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// scores[32][SK32] = X[t0.., :] . Y[:, :]^T over the 256-wide feature axis, into acc[kt] of wave (kt % 4)
__device__ __forceinline__ void score_phase(f32x16 (&acc)[2], const float* __restrict__ X, int Tq, int t0,
                                            const float* __restrict__ Y, int S, int SK32, int SKp, float* Xs,
                                            float* Ys, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  zero16(acc[0]);
  zero16(acc[1]);
  for (int d0 = 0; d0 < DH; d0 += 64) {
    __syncthreads();
    {  // X chunk, transposed: Xs[dd][row]
      const int row = tid & 31, dg = tid >> 5;  // 8 groups of 8 features
      float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (t0 + row < Tq) {
        const float* src = X + (size_t)(t0 + row) * DH + d0 + dg * 8;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) Xs[(dg * 8 + j) * LQ + row] = v[j];
    }
    for (int idx = tid; idx < SK32 * 8; idx += 256) {  // Y chunk, transposed: Ys[dd][key]
      const int key = idx >> 3, dg = idx & 7;
      float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (key < S) {
        const float* src = Y + (size_t)key * DH + d0 + dg * 8;
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) Ys[(dg * 8 + j) * SKp + key] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kt = wave + 4 * s;
      if (kt * 32 < SK32) mma32(acc[s], Xs, LQ, 0, Ys, SKp, kt * 32, 64, lane);
    }
  }
}

// out[32][256] = At[key][row]^T-view . Z[key][:]   (A image [k = key][row]); wave w owns feature tiles 2w, 2w+1
__device__ __forceinline__ void value_phase(f32x16 (&acc)[2], const float* At, const float* __restrict__ Z, int S,
                                            int SK32, float* Zs, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  zero16(acc[0]);
  zero16(acc[1]);
  for (int kc = 0; kc < SK32; kc += 32) {
    __syncthreads();
    for (int idx = tid; idx < 32 * (DH / 4); idx += 256) {  // Zs[kk][d], natural layout
      const int kk = idx / (DH / 4), c4 = idx - kk * (DH / 4);
      float4 v = make_float4(0, 0, 0, 0);
      if (kc + kk < S) v = *reinterpret_cast<const float4*>(Z + (size_t)(kc + kk) * DH + c4 * 4);
      *reinterpret_cast<float4*>(Zs + kk * LD + c4 * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) mma32(acc[j], At + kc * LQ, LQ, 0, Zs, LD, (2 * wave + j) * 32, 32, lane);
  }
}

__device__ __forceinline__ void store_rows(const f32x16 (&acc)[2], float* __restrict__ out, int Tq, int t0, int tid) {
  const int lane = tid & 63, wave = tid >> 6, lr = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (t0 + row < Tq) out[(size_t)(t0 + row) * DH + (2 * wave + j) * 32 + lr] = acc[j][r];
    }
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(int Tq, int S, const float* __restrict__ q,
                                                       const float* __restrict__ k, const float* __restrict__ v,
                                                       const int* __restrict__ lens, float scale,
                                                       float* __restrict__ ctx, float* __restrict__ P) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int SK32 = (S + 31) / 32 * 32, SKp = SK32 + 1;
  float* Xs = sm;                                    // [64][LQ]
  float* Ss = Xs + 64 * LQ;                          // [32][SKp]
  float* Pt = Ss + 32 * SKp;                         // [SK32][LQ]
  float* Ys = Pt + SK32 * LQ;                        // [64][SKp]  (aliased by the value tile [32][LD])
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lk = lane >> 5;
  const int b = blockIdx.y, t0 = blockIdx.x * QT;
  const float* qb = q + (size_t)b * Tq * DH;
  const float* kb = k + (size_t)b * S * DH;
  const float* vb = v + (size_t)b * S * DH;

  f32x16 acc[2];
  score_phase(acc, qb, Tq, t0, kb, S, SK32, SKp, Xs, Ys, tid);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int kt = wave + 4 * s;
    if (kt * 32 < SK32) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Ss[((r & 3) + 8 * (r >> 2) + 4 * lk) * SKp + kt * 32 + lr] = acc[s][r] * scale;
    }
  }
  __syncthreads();
  // masked softmax over the keys: wave w owns rows 8w .. 8w+7
  const int len = lens[b];
  for (int rr = 0; rr < 8; ++rr) {
    const int row = wave * 8 + rr;
    float x[3];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int key = lane + 64 * c;
      x[c] = (key < S && key < len) ? Ss[row * SKp + key] : -INFINITY;
      m = fmaxf(m, x[c]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float e[3], sum = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      e[c] = expf(x[c] - m);     // exp(-inf) = 0 for masked keys; an all-masked row gives NaN like torch
      sum += e[c];
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int key = lane + 64 * c;
      if (key < SK32) {
        const float p = key < S ? e[c] * inv : 0.f;
        Pt[key * LQ + row] = p;
        if (key < S && t0 + row < Tq) P[((size_t)b * Tq + t0 + row) * S + key] = p;
      }
    }
  }
  __syncthreads();
  value_phase(acc, Pt, vb, S, SK32, Ys, tid);
  store_rows(acc, ctx + (size_t)b * Tq * DH, Tq, t0, tid);
}

__global__ __launch_bounds__(256) void attn_bwd_kernel(int Tq, int S, const float* __restrict__ dctx,
                                                       const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const float* __restrict__ P,
                                                       float scale, float* __restrict__ dq, float* __restrict__ dk,
                                                       float* __restrict__ dv) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int SK32 = (S + 31) / 32 * 32, SKp = SK32 + 1;
  float* Xs = sm;                                    // [64][LQ]
  float* dSr = Xs + 64 * LQ;                         // [32][SKp]   dP, then dS (row-major)
  float* Ps = dSr + 32 * SKp;                        // [32][SKp]   probabilities (row-major)
  float* dSt = Ps + 32 * SKp;                        // [SK32][LQ]  dS transposed
  float* Ys = dSt + SK32 * LQ;                       // [64][SKp] / [32][LD] staging region
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lk = lane >> 5;
  const int b = blockIdx.y, t0 = blockIdx.x * QT;
  const float* gb = dctx + (size_t)b * Tq * DH;
  const float* qb = q + (size_t)b * Tq * DH;
  const float* kb = k + (size_t)b * S * DH;
  const float* vb = v + (size_t)b * S * DH;

  // dP = dctx . v^T
  f32x16 acc[2];
  score_phase(acc, gb, Tq, t0, vb, S, SK32, SKp, Xs, Ys, tid);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int kt = wave + 4 * s;
    if (kt * 32 < SK32) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dSr[((r & 3) + 8 * (r >> 2) + 4 * lk) * SKp + kt * 32 + lr] = acc[s][r];
    }
  }
  for (int idx = tid; idx < 32 * SK32; idx += 256) {
    const int row = idx / SK32, key = idx - row * SK32;
    Ps[row * SKp + key] = (key < S && t0 + row < Tq) ? P[((size_t)b * Tq + t0 + row) * S + key] : 0.f;
  }
  __syncthreads();
  // dS = P * (dP - rowsum(dP * P)) * scale
  for (int rr = 0; rr < 8; ++rr) {
    const int row = wave * 8 + rr;
    float p[3], g[3], dot = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int key = lane + 64 * c;
      p[c] = key < SK32 ? Ps[row * SKp + key] : 0.f;
      g[c] = key < SK32 ? dSr[row * SKp + key] : 0.f;
      dot += p[c] * g[c];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int key = lane + 64 * c;
      if (key < SK32) {
        const float ds = p[c] * (g[c] - dot) * scale;
        dSr[row * SKp + key] = ds;
        dSt[key * LQ + row] = ds;
      }
    }
  }
  __syncthreads();
  // dq = dS . k
  value_phase(acc, dSt, kb, S, SK32, Ys, tid);
  store_rows(acc, dq + (size_t)b * Tq * DH, Tq, t0, tid);

  // dk += dS^T . q_tile ;  dv += P^T . dctx_tile   (reduction over the 32 query rows)
  for (int pass = 0; pass < 2; ++pass) {
    const float* src = pass == 0 ? qb : gb;
    const float* At = pass == 0 ? dSr : Ps;   // [k = row][m = key]
    float* dst = (pass == 0 ? dk : dv) + (size_t)b * S * DH;
    __syncthreads();
    for (int idx = tid; idx < 32 * (DH / 4); idx += 256) {
      const int row = idx / (DH / 4), c4 = idx - row * (DH / 4);
      float4 t = make_float4(0, 0, 0, 0);
      if (t0 + row < Tq) t = *reinterpret_cast<const float4*>(src + (size_t)(t0 + row) * DH + c4 * 4);
      *reinterpret_cast<float4*>(Ys + row * LD + c4 * 4) = t;
    }
    __syncthreads();
    for (int mt = 0; mt < SK32 / 32; ++mt) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c;
        zero16(c);
        mma32(c, At, SKp, mt * 32, Ys, LD, (2 * wave + j) * 32, 32, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          if (key < S) atomicAdd(dst + (size_t)key * DH + (2 * wave + j) * 32 + lr, c[r]);
        }
      }
    }
  }
}

size_t lds_bytes(int S, bool bwd) {
  const int SK32 = (S + 31) / 32 * 32, SKp = SK32 + 1;
  size_t stage = (size_t)64 * SKp;
  if (stage < (size_t)32 * LD) stage = (size_t)32 * LD;
  size_t n = (size_t)64 * LQ + (size_t)32 * SKp * (bwd ? 2 : 1) + (size_t)SK32 * LQ + stage;
  return n * sizeof(float);
}

}  // namespace

extern "C" int vca_attention_fwd_f32(int B, int Tq, int S, int D, const float* q, const float* k, const float* v,
                                     const int* lens, float scale, float* ctx, float* probs, void* stream) {
  VCA_CHECK_ARG(B > 0 && Tq > 0 && S > 0 && S <= MAXS && D == DH && q && k && v && lens && ctx && probs);
  VCA_MAX_LDS(attn_fwd_kernel);
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((Tq + QT - 1) / QT, B), dim3(256), lds_bytes(S, false), (hipStream_t)stream,
                     Tq, S, q, k, v, lens, scale, ctx, probs);
  return vca_launch_status();
}

extern "C" int vca_attention_bwd_f32(int B, int Tq, int S, int D, const float* dctx, const float* q, const float* k,
                                     const float* v, const float* probs, float scale, float* dq, float* dk, float* dv,
                                     void* stream) {
  VCA_CHECK_ARG(B > 0 && Tq > 0 && S > 0 && S <= MAXS && D == DH && dctx && q && k && v && probs && dq && dk && dv);
  VCA_MAX_LDS(attn_bwd_kernel);
  hipStream_t s = (hipStream_t)stream;
  vca_zero_async(dk, (size_t)B * S * D, s);
  vca_zero_async(dv, (size_t)B * S * D, s);
  hipLaunchKernelGGL(attn_bwd_kernel, dim3((Tq + QT - 1) / QT, B), dim3(256), lds_bytes(S, true), s, Tq, S, dctx, q, k,
                     v, probs, scale, dq, dk, dv);
  return vca_launch_status();
}
