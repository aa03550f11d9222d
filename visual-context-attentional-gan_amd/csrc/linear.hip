// Linear layers and 1x1 / stride-1 convolutions: C[m][n] = sum_k A[m][k] * W[n][k] (+ bias[n]) (+ res[m][n]).
// Replaces nn.Linear and the pointwise nn.Conv2d / nn.Conv1d of the hot path (generator.py:145-148,167 the attention
// projections, :25,68,108 the 1x1 shortcuts, :184 the Postnet's last layer, :299,302 the discriminator heads;
// visual_front.py:20-21 the GRU's input products and the sentence projection; generator.py:335 the sync
// discriminator's Linear) in the forward pass and, with the transposed weight pack, in the input gradient.
//
// These are skinny products (32 ... 4800 rows, 1 ... 1536 columns, reductions of 32 ... 5120) whose time on the
// general convolution kernel was its fixed cost (halo plan, per-slice barriers, split-reduction fold: 25-45 us per
// launch whatever the size).  Here a wave owns a 32-row x 64-column output tile and streams both operands straight
// from L2 into registers: no LDS image, no barrier in the main loop, the next reduction chunk's loads in flight under
// the current chunk's MFMAs.  Activations are fp32 and split into hi/lo bf16 in registers; weights are read in the
// packed split-bf16 layout of vca_pack_weight_bf16 ([chunk of 32 k][column][hi 32 | lo 32], one tap), whose 16-byte
// pieces are exactly the fragments of v_mfma_f32_16x16x32_bf16.  The weight fragment is the row operand, so a lane
// ends up with 4 consecutive columns of one row (one float4 store).  With few tiles the four waves of a workgroup
// split the reduction of ONE tile and meet in LDS; otherwise they take four tiles.
//
// Both A and C may be strided views of channels-last maps (vca_lin_desc): the attention query projection reads the
// decoder map (B, F, T, C) as rows (b, t) with the reduction index running over (f, c), and the output projection
// writes its (B, T, F*C') result as the map (B, F, T, C') - the view / permute copies of generator.py:159,169 never
// happen.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct LinArgs {
  vca_lin_desc d;
  const float* a;
  const unsigned char* w;   // packed weights, one tap
  const float* bias;
  const float* res;
  float* c;
  int npad, nchunk, ntile_n, ntiles, ksplit, vec_store;
};

union Frag8 {
  bf16x8 v;
  uint4 u;
};

__device__ __forceinline__ void split8(const float4& p, const float4& q, Frag8& hi, Frag8& lo) {
  const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 h = (__bf16)v[e];
    hi.v[e] = h;
    lo.v[e] = (__bf16)(v[e] - (float)h);
  }
}

template <int NSPLIT>
__global__ __launch_bounds__(256) void linear_kernel(const LinArgs a) {
  __shared__ float red[3][32][64];   // partial tiles of waves 1-3 when the waves split the reduction
  const vca_lin_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, g = lane >> 4;
  const int tile = a.ksplit ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
  if (tile >= a.ntiles) return;       // (without the reduction split only: whole waves leave, no barrier follows)
  const int nt = tile % a.ntile_n, mt = tile / a.ntile_n;
  const int m0 = mt * 32, n0 = nt * 64;
  const int nsub = min(4, (a.npad - n0) >> 4);      // 16-column sub-tiles that exist in the weight pack

  // this lane's two activation rows (sub-tiles of 16 rows; rows past M are clamped and never stored)
  const float* arow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = min(m0 + 16 * i + lc, d.M - 1);
    arow[i] = a.a + (size_t)(m / d.a_tm) * d.a_bs + (size_t)(m % d.a_tm) * d.a_ts + 8 * g;
  }
  // weight rows: column n0 + 16 j + lc of chunk c starts at ((c * npad + n) * 128) bytes; this lane's piece at + 16 g
  const unsigned char* wrow = a.w + ((size_t)(n0 + lc) * 128 + 16 * g);
  const size_t wchunk = (size_t)a.npad * 128;

  f32x4 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int cstep = a.ksplit ? 4 : 1;
  int c = a.ksplit ? wave : 0;
  // reduction offset of chunk c inside A: (k / a_ck) * a_ks + (k % a_ck), kept incrementally (a_ck is a multiple of 32)
  long long kq = (long long)((c * 32) / d.a_ck) * d.a_ks;
  int kr = (c * 32) % d.a_ck;

  float4 ra[2][2];
  uint4 rw[4][2];
  auto fetch = [&](int cc, long long koff) {
    const bool kin = cc * 32 + 8 * g < d.K;          // K is a multiple of 8: a lane's 8 values are all in or all out
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[i][0] = ra[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kin) {
        const float4* p = reinterpret_cast<const float4*>(arow[i] + koff);
        ra[i][0] = p[0];
        ra[i][1] = p[1];
      }
    }
    const unsigned char* wp = wrow + (size_t)cc * wchunk;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nsub) {
        rw[j][0] = *reinterpret_cast<const uint4*>(wp + (size_t)j * 16 * 128);
        if (NSPLIT == 3) rw[j][1] = *reinterpret_cast<const uint4*>(wp + (size_t)j * 16 * 128 + 64);
      }
  };

  if (c < a.nchunk) fetch(c, kq + kr);
  for (; c < a.nchunk; c += cstep) {
    Frag8 ah[2], al[2], wh[4], wl[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) split8(ra[i][0], ra[i][1], ah[i], al[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wh[j].u = rw[j][0];
      wl[j].u = rw[j][1];
    }
    const int cn = c + cstep;
    if (cn < a.nchunk) {                                // next chunk's loads fly under this chunk's MFMAs
      kr += 32 * cstep;
      while (kr >= d.a_ck) {
        kr -= d.a_ck;
        kq += d.a_ks;
      }
      fetch(cn, kq + kr);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < nsub) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (NSPLIT == 3) {
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j].v, ah[i].v, acc[j][i], 0, 0, 0);
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j].v, al[i].v, acc[j][i], 0, 0, 0);
          }
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j].v, ah[i].v, acc[j][i], 0, 0, 0);
        }
      }
  }

  if (a.ksplit) {
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[wave - 1][(j * 2 + i) * 4 + r][lane] = acc[j][i][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[j][i][r] += red[w][(j * 2 + i) * 4 + r][lane];
  }

  // D[row][col]: row = 4 g + r (column n of C within the sub-tile), col = lc (row m of C)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + 16 * i + lc;
    if (m >= d.M) continue;
    const size_t mo = (size_t)(m / d.c_tm) * d.c_bs + (size_t)(m % d.c_tm) * d.c_ts;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 16 * j + 4 * g;
      if (j >= nsub || n >= d.N) continue;
      const size_t o = mo + (size_t)(n / d.c_cn) * d.c_ns + (n % d.c_cn);
      float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
      if (a.vec_store) {                               // N and c_cn multiples of 4, 16-byte aligned addresses
        if (a.bias) {
          const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);
          v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        }
        if (a.res) {
          const float4 rv = *reinterpret_cast<const float4*>(a.res + o);
          v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
        }
        *reinterpret_cast<float4*>(a.c + o) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < d.N) {
            const size_t oe = mo + (size_t)((n + r) / d.c_cn) * d.c_ns + ((n + r) % d.c_cn);
            float t = v[r];
            if (a.bias) t += a.bias[n + r];
            if (a.res) t += a.res[oe];
            a.c[oe] = t;
          }
      }
    }
  }
}

}  // namespace

extern "C" int vca_linear_bf16(const vca_lin_desc* dp, int nsplit, const float* a, const void* wpack, const float* bias,
                               const float* res, float* c, void* stream) {
  VCA_CHECK_ARG(dp && a && wpack && c && (nsplit == 1 || nsplit == 3));
  const vca_lin_desc& d = *dp;
  VCA_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0 && (d.K % 8) == 0 && d.a_tm > 0 && d.c_tm > 0 && d.a_ck > 0 && d.c_cn > 0);
  // a lane reads 8 consecutive reduction elements as two aligned float4: chunks of 32 must not straddle a_ck
  VCA_CHECK_ARG((d.a_ck % 32) == 0 || d.a_ck >= d.K);
  VCA_CHECK_ARG((d.a_bs % 4) == 0 && (d.a_ts % 4) == 0 && (d.a_ks % 4) == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0);
  VCA_CHECK_ARG((reinterpret_cast<uintptr_t>(wpack) & 15) == 0);
  LinArgs g;
  g.d = d;
  if (d.a_ck >= d.K) g.d.a_ck = (d.K + 31) / 32 * 32 + 32;   // plain rows: never wraps
  g.a = a;
  g.w = reinterpret_cast<const unsigned char*>(wpack);
  g.bias = bias;
  g.res = res;
  g.c = c;
  g.npad = (d.N + 31) / 32 * 32;
  g.nchunk = (d.K + 31) / 32;
  g.ntile_n = (d.N + 63) / 64;
  const long long ntiles = (long long)((d.M + 31) / 32) * g.ntile_n;
  VCA_CHECK_ARG(ntiles < (1ll << 30));
  g.ntiles = (int)ntiles;
  // few tiles and a reduction worth splitting: the four waves of a workgroup share one tile
  g.ksplit = (ntiles < 1024 && g.nchunk >= 4) ? 1 : 0;
  g.vec_store = ((d.N % 4) == 0 && (d.c_cn % 4) == 0 && (d.c_bs % 4) == 0 && (d.c_ts % 4) == 0 && (d.c_ns % 4) == 0 &&
                 (reinterpret_cast<uintptr_t>(c) & 15) == 0 && (!res || (reinterpret_cast<uintptr_t>(res) & 15) == 0) &&
                 (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0)) ? 1 : 0;
  const int grid = g.ksplit ? g.ntiles : (g.ntiles + 3) / 4;
  if (nsplit == 3)
    hipLaunchKernelGGL(linear_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, g);
  else
    hipLaunchKernelGGL(linear_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, g);
  return vca_launch_status();
}
