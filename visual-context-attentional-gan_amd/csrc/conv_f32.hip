// Exact-fp32 implicit-GEMM convolution for gfx950 on v_mfma_f32_32x32x2_f32.
//
// GEMM view (forward): rows = output pixels (channels-last, so a row of the output is one pixel's
// Cout values), columns = output channels, reduction = (filter tap, input channel).  The A operand
// (activations) is gathered tap by tap straight from the channels-last input, staged in LDS as
// [k][pixel]; the B operand is the pre-packed weight [tap][ci][co], staged as [k][co].  One lane of
// the 32x32x2 MFMA supplies A[row = lane&31][k = lane>>5] and B[k = lane>>5][col = lane&31], so both
// LDS images are read with conflict-free ds_read_b32 (32 consecutive dwords per half-wave).
//
// Weight gradient: rows = input channels, columns = output channels, reduction = output pixels; both
// operands are read in their natural channels-last layout.  Pixels are split across workgroups and
// the partial tiles are accumulated with fp32 atomics (256-B contiguous per wave-instruction).
#include "common.h"

namespace {

constexpr int BM = 128;  // output pixels per workgroup (forward)
constexpr int KC = 16;   // reduction slice per K-step
constexpr int LDA = BM + 4;

struct FwdArgs {
  vca_conv_desc d;
  const float* x;
  const float* w;
  const float* bias;
  const float* in_scale;
  const float* in_shift;
  const float* in_alpha;
  const float* res;
  float* y;
  float* stats;
  int P;       // output pixels
  int Cout4;   // packed weight row length
  int nchunk;  // ceil(Cin / KC)
  int nks;     // K-steps
  int vecA;    // Cin % 4 == 0: 16-byte activation loads allowed
};

__device__ __forceinline__ float prologue1(float v, int ci, const FwdArgs& a) {
  if (a.in_scale) v = v * a.in_scale[ci] + a.in_shift[ci];
  const int act = a.d.act;
  if (act) v = vca_act(v, act, act == VCA_ACT_PRELU ? a.in_alpha[ci] : a.d.slope);
  return v;
}

// ------------------------------------------------------------------------------------------------
// forward, general case (any Cin; taps are 2-D)
// ------------------------------------------------------------------------------------------------
template <int BN, int WM, int WN, bool CIN1>
__global__ __launch_bounds__(256) void conv_f32_fwd_kernel(FwdArgs a) {
  constexpr int LDB = BN + 4;
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NT = WTN / 32;
  constexpr int NB4 = (4 * BN + 255) / 256;  // float4 weight loads per thread per K-step
  __shared__ float As[2][KC][LDA];
  __shared__ float Bs[2][KC][LDB];
  __shared__ int taptab[CIN1 ? 256 : 1];

  const vca_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int p0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int ntaps = d.KD * d.KH * d.KW;

  if (CIN1) {
    for (int t = tid; t < 256; t += 256) {
      int v = -1;
      if (t < ntaps) {
        const int kd = t / (d.KH * d.KW), r = t - kd * d.KH * d.KW, kh = r / d.KW, kw = r - kh * d.KW;
        v = (kd << 16) | (kh << 8) | kw;
      }
      taptab[t] = v;
    }
    __syncthreads();
  }

  // staging role for A: this thread owns pixel `am` and reduction entries ah*8 .. ah*8+7 of each slice
  const int am = tid >> 1, ah = tid & 1;
  const int ap = p0 + am;
  const bool apv = ap < a.P;
  int an = 0, aod = 0, aoy = 0, aox = 0;
  if (apv) {
    int r = ap;
    aox = r % d.OW; r /= d.OW;
    aoy = r % d.OH; r /= d.OH;
    aod = r % d.OD; an = r / d.OD;
  }
  const int He = d.Hin << d.ups, We = d.Win << d.ups;

  float ra[8];
  float4 rb[NB4];

  auto issue = [&](int ks) {
    if (!CIN1) {
      const int tap = ks / a.nchunk, chunk = ks - tap * a.nchunk;
      const int kh = tap / d.KW, kw = tap - kh * d.KW;
      int iy = aoy * d.stride + kh - d.pad_h, ix = aox * d.stride + kw - d.pad_w;
      bool v = apv && iy >= 0 && ix >= 0;
      if (d.dil > 1) {
        v = v && (iy % d.dil == 0) && (ix % d.dil == 0);
        iy /= d.dil;
        ix /= d.dil;
      }
      v = v && iy < He && ix < We;
      iy >>= d.ups;
      ix >>= d.ups;
      const int ci0 = chunk * KC + ah * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) ra[j] = 0.f;
      if (v) {
        const float* src = a.x + ((size_t)((an * d.D + aod) * d.Hin + iy) * d.Win + ix) * d.Cin + ci0;
        if (a.vecA && ci0 + 8 <= d.Cin) {
          const float4 v0 = *reinterpret_cast<const float4*>(src);
          const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
          ra[0] = v0.x; ra[1] = v0.y; ra[2] = v0.z; ra[3] = v0.w;
          ra[4] = v1.x; ra[5] = v1.y; ra[6] = v1.z; ra[7] = v1.w;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (ci0 + j < d.Cin) ra[j] = src[j];
        }
        if (a.in_scale || d.act) {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (ci0 + j < d.Cin) ra[j] = prologue1(ra[j], ci0 + j, a);
        }
      }
#pragma unroll
      for (int i = 0; i < NB4; ++i) {
        const int idx = tid + i * 256;
        rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < 4 * BN) {
          const int k = idx / (BN / 4), c4 = idx - k * (BN / 4);
          const int n = n0 + c4 * 4, ci = chunk * KC + k;
          if (n < a.Cout4 && ci < d.Cin)
            rb[i] = *reinterpret_cast<const float4*>(a.w + ((size_t)tap * d.Cin + ci) * a.Cout4 + n);
        }
      }
    } else {
      // Cin == 1: the reduction index is the filter tap itself (3-D taps allowed)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int tap = ks * KC + ah * 8 + j;
        const int code = tap < 256 ? taptab[tap] : -1;
        float v = 0.f;
        if (apv && code >= 0) {
          const int iz = aod + (code >> 16) - d.pad_d;
          const int iy = aoy * d.stride + ((code >> 8) & 255) - d.pad_h;
          const int ix = aox * d.stride + (code & 255) - d.pad_w;
          if (iz >= 0 && iz < d.D && iy >= 0 && iy < d.Hin && ix >= 0 && ix < d.Win)
            v = a.x[((size_t)(an * d.D + iz) * d.Hin + iy) * d.Win + ix];
        }
        ra[j] = v;
      }
#pragma unroll
      for (int i = 0; i < NB4; ++i) {
        const int idx = tid + i * 256;
        rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < 4 * BN) {
          const int k = idx / (BN / 4), c4 = idx - k * (BN / 4);
          const int n = n0 + c4 * 4, tap = ks * KC + k;
          if (n < a.Cout4 && tap < ntaps)
            rb[i] = *reinterpret_cast<const float4*>(a.w + (size_t)tap * a.Cout4 + n);
        }
      }
    }
  };

  auto commit = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 8; ++j) As[buf][ah * 8 + j][am] = ra[j];
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const int idx = tid + i * 256;
      if (idx < 4 * BN) {
        const int k = idx / (BN / 4), c4 = idx - k * (BN / 4);
        *reinterpret_cast<float4*>(&Bs[buf][k][c4 * 4]) = rb[i];
      }
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lk = lane >> 5;
  issue(0);
  commit(0);
  __syncthreads();
  for (int ks = 0; ks < a.nks; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < a.nks) issue(ks + 1);
#pragma unroll
    for (int kk = 0; kk < KC / 2; ++kk) {
      float av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = As[buf][2 * kk + lk][wm * WTM + i * 32 + lr];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = Bs[buf][2 * kk + lk][wn * WTN + j * 32 + lr];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (ks + 1 < a.nks) commit(buf ^ 1);
    __syncthreads();
  }

  // epilogue: D[row][col]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * WTN + j * 32 + lr;
    const bool nv = n < d.Cout;
    const float b = (nv && a.bias) ? a.bias[n] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int p = p0 + wm * WTM + i * 32 + row;
        if (nv && p < a.P) {
          float v = acc[i][j][r];
          s1 += v;
          s2 += v * v;
          v += b;
          const size_t o = (size_t)p * d.Cout + n;
          if (a.res) v += a.res[o];
          a.y[o] = v * d.out_scale;
        }
      }
    }
    if (a.stats) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (nv && lk == 0) {
        atomicAdd(a.stats + n, s1);
        atomicAdd(a.stats + d.Cout + n, s2);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
struct WgArgs {
  vca_conv_desc d;
  const float* x;
  const float* gy;
  const float* in_scale;
  const float* in_shift;
  const float* in_alpha;
  float* dw;
  int P, Cout4;
  int mtiles, ntiles;  // tiles over (Cin | taps) and Cout
  int pchunk;          // pixels per workgroup (multiple of KC)
  int vecA, vecB;
};

template <int BMW, int BNW, int WM, int WN, bool CIN1>
__global__ __launch_bounds__(WM* WN * 64) void conv_f32_wgrad_kernel(WgArgs a) {
  constexpr int NTHR = WM * WN * 64;
  constexpr int LDAW = BMW + 4, LDBW = BNW + 4;
  constexpr int WTM = BMW / WM, WTN = BNW / WN, MT = WTM / 32, NT = WTN / 32;
  constexpr int NA = (KC * BMW / 4 + NTHR - 1) / NTHR;  // 4-wide groups of A per thread
  constexpr int NB = (KC * BNW / 4 + NTHR - 1) / NTHR;
  __shared__ float As[2][KC][LDAW];
  __shared__ float Bs[2][KC][LDBW];

  const vca_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // blockIdx.x -> (tap | tap-tile, m-tile, n-tile)
  int bid = blockIdx.x;
  const int nt_ = bid % a.ntiles; bid /= a.ntiles;
  const int mt_ = bid % a.mtiles; bid /= a.mtiles;
  const int tap = bid;  // general: filter tap; CIN1: unused (0)
  const int m0 = mt_ * BMW, n0 = nt_ * BNW;
  const int pbeg = blockIdx.y * a.pchunk;
  const int pend = min(a.P, pbeg + a.pchunk);
  const int nks = (pend - pbeg + KC - 1) / KC;
  const int He = d.Hin << d.ups, We = d.Win << d.ups;
  const int kh = CIN1 ? 0 : tap / d.KW, kw = CIN1 ? 0 : tap - (tap / d.KW) * d.KW;
  const int KHW = d.KH * d.KW, ntaps = d.KD * KHW;

  float4 ra[NA], rb[NB];

  auto issue = [&](int ks) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + i * NTHR;
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < KC * BMW / 4) {
        const int k = idx / (BMW / 4), c4 = idx - k * (BMW / 4);
        const int p = pbeg + ks * KC + k;
        if (p < pend) {
          int r = p;
          const int ox = r % d.OW; r /= d.OW;
          const int oy = r % d.OH; r /= d.OH;
          const int od = r % d.OD; const int n = r / d.OD;
          if (!CIN1) {
            int iy = oy * d.stride + kh - d.pad_h, ix = ox * d.stride + kw - d.pad_w;
            bool v = iy >= 0 && ix >= 0 && iy < He && ix < We;
            iy >>= d.ups;
            ix >>= d.ups;
            const int ci0 = m0 + c4 * 4;
            if (v && ci0 < d.Cin) {
              const float* src = a.x + ((size_t)((n * d.D + od) * d.Hin + iy) * d.Win + ix) * d.Cin + ci0;
              float t[4] = {0.f, 0.f, 0.f, 0.f};
              if (a.vecA && ci0 + 4 <= d.Cin) {
                const float4 v4 = *reinterpret_cast<const float4*>(src);
                t[0] = v4.x; t[1] = v4.y; t[2] = v4.z; t[3] = v4.w;
              } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                  if (ci0 + j < d.Cin) t[j] = src[j];
              }
              if (a.in_scale || d.act) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const int ci = ci0 + j;
                  if (ci < d.Cin) {
                    float z = t[j];
                    if (a.in_scale) z = z * a.in_scale[ci] + a.in_shift[ci];
                    if (d.act) z = vca_act(z, d.act, d.act == VCA_ACT_PRELU ? a.in_alpha[ci] : d.slope);
                    t[j] = z;
                  }
                }
              }
              ra[i] = make_float4(t[0], t[1], t[2], t[3]);
            }
          } else {
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int tp = m0 + c4 * 4 + j;
              float v = 0.f;
              if (tp < ntaps) {
                const int kd = tp / KHW, rr = tp - kd * KHW, kh2 = rr / d.KW, kw2 = rr - kh2 * d.KW;
                const int iz = od + kd - d.pad_d, iy = oy * d.stride + kh2 - d.pad_h, ix = ox * d.stride + kw2 - d.pad_w;
                if (iz >= 0 && iz < d.D && iy >= 0 && iy < d.Hin && ix >= 0 && ix < d.Win)
                  v = a.x[((size_t)(n * d.D + iz) * d.Hin + iy) * d.Win + ix];
              }
              t[j] = v;
            }
            ra[i] = make_float4(t[0], t[1], t[2], t[3]);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = tid + i * NTHR;
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < KC * BNW / 4) {
        const int k = idx / (BNW / 4), c4 = idx - k * (BNW / 4);
        const int p = pbeg + ks * KC + k;
        const int co0 = n0 + c4 * 4;
        if (p < pend && co0 < d.Cout) {
          const float* src = a.gy + (size_t)p * d.Cout + co0;
          if (a.vecB && co0 + 4 <= d.Cout) {
            rb[i] = *reinterpret_cast<const float4*>(src);
          } else {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (co0 + j < d.Cout) t[j] = src[j];
            rb[i] = make_float4(t[0], t[1], t[2], t[3]);
          }
        }
      }
    }
  };

  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + i * NTHR;
      if (idx < KC * BMW / 4) {
        const int k = idx / (BMW / 4), c4 = idx - k * (BMW / 4);
        *reinterpret_cast<float4*>(&As[buf][k][c4 * 4]) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = tid + i * NTHR;
      if (idx < KC * BNW / 4) {
        const int k = idx / (BNW / 4), c4 = idx - k * (BNW / 4);
        *reinterpret_cast<float4*>(&Bs[buf][k][c4 * 4]) = rb[i];
      }
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lr = lane & 31, lk = lane >> 5;
  if (nks > 0) {
    issue(0);
    commit(0);
  }
  __syncthreads();
  for (int ks = 0; ks < nks; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nks) issue(ks + 1);
#pragma unroll
    for (int kk = 0; kk < KC / 2; ++kk) {
      float av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = As[buf][2 * kk + lk][wm * WTM + i * 32 + lr];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = Bs[buf][2 * kk + lk][wn * WTN + j * 32 + lr];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (ks + 1 < nks) commit(buf ^ 1);
    __syncthreads();
  }

  const int mlim = CIN1 ? ntaps : d.Cin;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = n0 + wn * WTN + j * 32 + lr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int m = m0 + wm * WTM + i * 32 + row;
        if (m < mlim && co < d.Cout) {
          const size_t o = CIN1 ? ((size_t)m * a.Cout4 + co) : (((size_t)tap * d.Cin + m) * a.Cout4 + co);
          atomicAdd(a.dw + o, acc[i][j][r]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight re-layout
// ------------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(int mode, int Cout, int Cin, int taps, const float* __restrict__ src,
                                   float* __restrict__ dst, float beta) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (mode == 0) {  // dst[t][ci][co4]
    const int Cout4 = (Cout + 3) & ~3;
    const size_t total = (size_t)taps * Cin * Cout4;
    if (i >= total) return;
    const int co = i % Cout4;
    const size_t r = i / Cout4;
    const int ci = r % Cin, t = r / Cin;
    dst[i] = co < Cout ? src[((size_t)co * Cin + ci) * taps + t] : 0.f;
  } else if (mode == 1) {  // dst[t][co][ci4] = src[co][ci][taps-1-t]
    const int Cin4 = (Cin + 3) & ~3;
    const size_t total = (size_t)taps * Cout * Cin4;
    if (i >= total) return;
    const int ci = i % Cin4;
    const size_t r = i / Cin4;
    const int co = r % Cout, t = r / Cout;
    dst[i] = ci < Cin ? src[((size_t)co * Cin + ci) * taps + (taps - 1 - t)] : 0.f;
  } else {  // dst[co][ci][t] = beta*dst + src[t][ci][co4]
    const int Cout4 = (Cout + 3) & ~3;
    const size_t total = (size_t)Cout * Cin * taps;
    if (i >= total) return;
    const int t = i % taps;
    const size_t r = i / taps;
    const int ci = r % Cin, co = r / Cin;
    const float v = src[((size_t)t * Cin + ci) * Cout4 + co];
    dst[i] = beta == 0.f ? v : beta * dst[i] + v;
  }
}

// dst[co][ci][t] = beta*dst + src[t][ci][co4], and src is cleared behind the read: the weight-gradient kernels
// accumulate into src with atomics, so a workspace that is always handed back zeroed never needs a fill launch.
// A workgroup transposes 32 co x 64 (ci,t) entries through LDS so both the reads and the writes are coalesced.
__device__ __forceinline__ void unpack_wgrad_tile(int Cout, int Cin, int taps, float* __restrict__ src,
                                                  float* __restrict__ dst, float beta, int clear, int etile,
                                                  int cotile, float (*tile)[33]) {
  const int Cout4 = (Cout + 3) & ~3;
  const int E = Cin * taps;
  const int e0 = etile * 64, co0 = cotile * 32;
  const int tid = threadIdx.x;
  {
    const int cl = tid & 31, co = co0 + cl;
    int el = tid >> 5;
    int ci = (e0 + el) / taps, t = (e0 + el) - ci * taps;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass, el += 8) {
      float v = 0.f;
      if (e0 + el < E && co < Cout4) {
        float* ps = src + ((size_t)t * Cin + ci) * Cout4 + co;
        v = *ps;
        if (clear) *ps = 0.f;
      }
      tile[el][cl] = v;
      t += 8;
      while (t >= taps) { t -= taps; ++ci; }
    }
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int cl = pass * 4 + (tid >> 6), el = tid & 63;
    const int e = e0 + el, co = co0 + cl;
    if (e < E && co < Cout) {
      float* pd = dst + (size_t)co * E + e;
      const float v = tile[el][cl];
      *pd = beta == 0.f ? v : beta * *pd + v;
    }
  }
}

__global__ __launch_bounds__(256) void unpack_wgrad_kernel(int Cout, int Cin, int taps, float* __restrict__ src,
                                                           float* __restrict__ dst, float beta, int clear) {
  __shared__ float tile[64][33];
  unpack_wgrad_tile(Cout, Cin, taps, src, dst, beta, clear, blockIdx.x, blockIdx.y, tile);
}

// The same for every weight-gradient workspace of a backward pass in one launch: descriptors with a running first
// tile; a workgroup finds its descriptor by bisection.  Always dst += src and src cleared.
__global__ __launch_bounds__(256) void unpack_wgrad_batch_kernel(const vca_unpack_desc* __restrict__ table, int ndesc) {
  __shared__ float tile[64][33];
  const int t = blockIdx.x;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].tile_begin <= t) lo = mid;
    else hi = mid - 1;
  }
  const vca_unpack_desc d = table[lo];
  const int local = t - d.tile_begin;
  const int et = (d.Cin * d.taps + 63) / 64, ct = (((d.Cout + 3) & ~3) + 31) / 32;
  if (local >= et * ct) return;
  unpack_wgrad_tile(d.Cout, d.Cin, d.taps, d.src, d.dst, 1.0f, 1, local % et, local / et, tile);
}

bool desc_ok(const vca_conv_desc* d) {
  return d && d->N > 0 && d->D > 0 && d->Hin > 0 && d->Win > 0 && d->Cin > 0 && d->OD > 0 && d->OH > 0 &&
         d->OW > 0 && d->Cout > 0 && d->KD > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad_h >= 0 && d->pad_w >= 0 &&
         d->pad_d >= 0 && d->dil >= 1 && (d->ups == 0 || d->ups == 1) && d->act >= 0 && d->act <= 3;
}

}  // namespace

extern "C" int vca_abi_version(void) { return 3; }

extern "C" int vca_elem_type(void) { return VCA_ELEM_TYPE; }

extern "C" int vca_conv_fwd_f32(const vca_conv_desc* d, const float* x, const float* wpack, const float* bias,
                                const float* in_scale, const float* in_shift, const float* in_alpha,
                                const float* res, float* y, float* stats, void* stream) {
  VCA_CHECK_ARG(desc_ok(d) && x && wpack && y);
  VCA_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr));
  VCA_CHECK_ARG(d->act != VCA_ACT_PRELU || in_alpha);
  const long long P = (long long)d->N * d->OD * d->OH * d->OW;
  VCA_CHECK_ARG(P < (1ll << 31));
  const bool cin1 = d->Cin == 1;
  if (cin1) {
    VCA_CHECK_ARG(d->KD * d->KH * d->KW <= 256 && d->KH < 256 && d->KW < 256 && d->dil == 1 && d->ups == 0 &&
                  d->act == 0 && !in_scale);
    VCA_CHECK_ARG(d->OD == d->D + 2 * d->pad_d - d->KD + 1);
  } else {
    VCA_CHECK_ARG(d->KD == 1 && d->pad_d == 0 && d->OD == d->D);
  }
  FwdArgs a;
  a.d = *d;
  a.x = x; a.w = wpack; a.bias = bias; a.in_scale = in_scale; a.in_shift = in_shift; a.in_alpha = in_alpha;
  a.res = res; a.y = y; a.stats = stats;
  a.P = (int)P;
  a.Cout4 = (d->Cout + 3) & ~3;
  a.nchunk = (d->Cin + KC - 1) / KC;
  a.nks = cin1 ? (d->KD * d->KH * d->KW + KC - 1) / KC : d->KH * d->KW * a.nchunk;
  a.vecA = (d->Cin % 4 == 0) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  const int gx = (int)((P + BM - 1) / BM);
#define LAUNCH(BN_, WM_, WN_)                                                                          \
  do {                                                                                                  \
    dim3 g(gx, (d->Cout + BN_ - 1) / BN_);                                                              \
    if (cin1) hipLaunchKernelGGL((conv_f32_fwd_kernel<BN_, WM_, WN_, true>), g, dim3(256), 0, s, a);    \
    else hipLaunchKernelGGL((conv_f32_fwd_kernel<BN_, WM_, WN_, false>), g, dim3(256), 0, s, a);        \
  } while (0)
  if (d->Cout > 64) LAUNCH(128, 2, 2);
  else if (d->Cout > 32) LAUNCH(64, 2, 2);
  else LAUNCH(32, 4, 1);
#undef LAUNCH
  return vca_launch_status();
}

extern "C" int vca_conv_wgrad_f32(const vca_conv_desc* d, const float* x, const float* gy, const float* in_scale,
                                  const float* in_shift, const float* in_alpha, float* dwpack, void* stream) {
  VCA_CHECK_ARG(desc_ok(d) && x && gy && dwpack);
  VCA_CHECK_ARG((in_scale == nullptr) == (in_shift == nullptr));
  VCA_CHECK_ARG(d->act != VCA_ACT_PRELU || in_alpha);
  VCA_CHECK_ARG(d->dil == 1);
  const long long P = (long long)d->N * d->OD * d->OH * d->OW;
  VCA_CHECK_ARG(P < (1ll << 31));
  const bool cin1 = d->Cin == 1;
  const int ntaps = d->KD * d->KH * d->KW;
  if (cin1) VCA_CHECK_ARG(d->ups == 0 && d->act == 0 && !in_scale && d->OD == d->D + 2 * d->pad_d - d->KD + 1);
  else VCA_CHECK_ARG(d->KD == 1 && d->pad_d == 0 && d->OD == d->D);
  WgArgs a;
  a.d = *d;
  a.x = x; a.gy = gy; a.in_scale = in_scale; a.in_shift = in_shift; a.in_alpha = in_alpha; a.dw = dwpack;
  a.P = (int)P;
  a.Cout4 = (d->Cout + 3) & ~3;
  a.vecA = (d->Cin % 4 == 0) ? 1 : 0;
  a.vecB = (d->Cout % 4 == 0) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  const int mdim = cin1 ? ntaps : d->Cin;
#define LAUNCHW(BMW_, BNW_, WM_, WN_)                                                                       \
  do {                                                                                                       \
    a.mtiles = (mdim + BMW_ - 1) / BMW_;                                                                     \
    a.ntiles = (d->Cout + BNW_ - 1) / BNW_;                                                                  \
    const long long tiles = (long long)(cin1 ? 1 : ntaps) * a.mtiles * a.ntiles;                             \
    long long split = (2048 + tiles - 1) / tiles;                                                            \
    long long maxsplit = (P + 511) / 512;                                                                    \
    if (split > maxsplit) split = maxsplit;                                                                  \
    if (split < 1) split = 1;                                                                                \
    a.pchunk = (int)(((P + split - 1) / split + KC - 1) / KC * KC);                                          \
    const int gy_ = (int)((P + a.pchunk - 1) / a.pchunk);                                                    \
    dim3 g((unsigned)tiles, gy_);                                                                            \
    if (cin1)                                                                                                \
      hipLaunchKernelGGL((conv_f32_wgrad_kernel<BMW_, BNW_, WM_, WN_, true>), g, dim3(WM_* WN_ * 64), 0, s, a); \
    else                                                                                                     \
      hipLaunchKernelGGL((conv_f32_wgrad_kernel<BMW_, BNW_, WM_, WN_, false>), g, dim3(WM_* WN_ * 64), 0, s, a); \
  } while (0)
  if (mdim <= 32) {
    if (d->Cout > 64) LAUNCHW(32, 128, 1, 4);
    else if (d->Cout > 32) LAUNCHW(32, 64, 1, 2);
    else LAUNCHW(32, 32, 1, 1);
  } else {
    if (d->Cout > 64) LAUNCHW(64, 128, 2, 2);
    else if (d->Cout > 32) LAUNCHW(64, 64, 2, 2);
    else LAUNCHW(64, 32, 2, 1);
  }
#undef LAUNCHW
  return vca_launch_status();
}

extern "C" int vca_pack_weight_f32(int mode, int Cout, int Cin, int taps, const float* src, float* dst, float beta,
                                   void* stream) {
  VCA_CHECK_ARG(mode >= 0 && mode <= 2 && Cout > 0 && Cin > 0 && taps > 0 && src && dst);
  size_t total;
  if (mode == 0) total = (size_t)taps * Cin * ((Cout + 3) & ~3);
  else if (mode == 1) total = (size_t)taps * Cout * ((Cin + 3) & ~3);
  else total = (size_t)Cout * Cin * taps;
  const int thr = 256;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + thr - 1) / thr)), dim3(thr), 0,
                     (hipStream_t)stream, mode, Cout, Cin, taps, src, dst, beta);
  return vca_launch_status();
}

extern "C" int vca_unpack_wgrad_batch_f32(const vca_unpack_desc* table, int ndesc, int total_tiles, void* stream) {
  VCA_CHECK_ARG(table && ndesc > 0 && total_tiles > 0);
  hipLaunchKernelGGL(unpack_wgrad_batch_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, table, ndesc);
  return vca_launch_status();
}

extern "C" int vca_unpack_wgrad_f32(int Cout, int Cin, int taps, float* dwpack, float* dst, float beta, int clear,
                                    void* stream) {
  VCA_CHECK_ARG(Cout > 0 && Cin > 0 && taps > 0 && dwpack && dst);
  const long long E = (long long)Cin * taps;
  VCA_CHECK_ARG(E < (1ll << 31));
  const dim3 grid((unsigned)((E + 63) / 64), (unsigned)((((Cout + 3) & ~3) + 31) / 32));
  hipLaunchKernelGGL(unpack_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, Cout, Cin, taps, dwpack, dst, beta,
                     clear);
  return vca_launch_status();
}
