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
    const vca_e16 h = (vca_e16)v[e];
    hi.v[e] = h;
    lo.v[e] = (vca_e16)(v[e] - (float)h);
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
            acc[j][i] = VCA_MFMA_16x16x32(wl[j].v, ah[i].v, acc[j][i]);
            acc[j][i] = VCA_MFMA_16x16x32(wh[j].v, al[i].v, acc[j][i]);
          }
          acc[j][i] = VCA_MFMA_16x16x32(wh[j].v, ah[i].v, acc[j][i]);
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


// ------------------------------------------------------------------------------------------------
// The same product for the larger shapes (a few GFLOP: attention projections, GRU input products, the sync
// discriminator's Linear): a 128 x 128 tile per workgroup through LDS, so that an operand byte fetched from L2 feeds
// 128 rows or columns instead of 32 / 64, and a ring of GSTAGES stages of 32-deep reduction chunks so that no load
// latency sits between two chunks: while chunk c feeds the MFMAs, the weights of chunk c + GSTAGES - 1 stream into
// their stage by LDS-DMA and that chunk's activations fly into a register set; chunk c + 1's activations (fetched
// GSTAGES - 2 iterations earlier) are split into hi/lo and written to their stage.  One barrier per chunk.  The
// instantiated depth is three (one register set); four measured 5-8 % longer.  LDS images are those of the
// convolution kernel (64-byte rows of 32 bf16, 16-byte chunks XOR-swizzled by ((row >> 2) & 1) << 1: conflict-free
// ds_read_b128 fragments).  With few tiles the reduction is split over blockIdx.z; the partial tiles go to slabs of a
// caller-provided workspace and gemm_fold_kernel adds them, bias and residual into C.
// ------------------------------------------------------------------------------------------------
constexpr int GBM = 128, GBN = 128;

struct GemmArgs {
  LinArgs l;
  float* part;            // [ksplit][M][N] partial products (ksplit > 1)
  int ksplit, cpz;        // reduction ranges; chunks per range
};

__device__ __forceinline__ int gsw16(int row) { return ((row >> 2) & 1) << 1; }

template <int NSPLIT, int GSTAGES>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int AIMG = GBM * 64;                       // bytes of one activation image (hi or lo)
  constexpr int ASTG = AIMG * (NSPLIT == 3 ? 2 : 1);
  constexpr int WROWB = NSPLIT == 3 ? 128 : 64;        // bytes of a packed weight row that are needed
  constexpr int WSTG = GBN * WROWB;
  constexpr int STG = ASTG + WSTG;
  const LinArgs& a = g.l;
  const vca_lin_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lc = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * GBM, n0 = blockIdx.y * GBN, kz = blockIdx.z;
  const int cbeg = kz * g.cpz, cend = min(a.nchunk, cbeg + g.cpz);
  const int nc = cend - cbeg;

  // activation staging: float4 q = tid + 256 j of the chunk's 128 rows x 8 float4
  const float* asrc[4];
  int adst[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = tid + 256 * j, row = q >> 3, c4 = q & 7;
    const int m = min(m0 + row, d.M - 1);
    asrc[j] = a.a + (size_t)(m / d.a_tm) * d.a_bs + (size_t)(m % d.a_tm) * d.a_ts + c4 * 4;
    adst[j] = row * 64 + ((((c4 >> 1) ^ gsw16(row))) << 4) + (c4 & 1) * 8;
  }
  constexpr int NR = GSTAGES - 2;         // register sets of activations in flight
  float4 ra[NR][4];
  long long kq = (long long)((cbeg * 32) / d.a_ck) * d.a_ks;
  int kr = (cbeg * 32) % d.a_ck;          // offset of the NEXT chunk to fetch
  auto fetch_a = [&](int c, float4 (&r)[4]) {   // c: chunk index relative to cbeg (chunks are fetched in order)
    const int k0 = (cbeg + c) * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c4 = (tid + 256 * j) & 7;
      r[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + c4 * 4 < d.K) r[j] = *reinterpret_cast<const float4*>(asrc[j] + kq + kr);
    }
    kr += 32;
    if (kr >= d.a_ck) {
      kr -= d.a_ck;
      kq += d.a_ks;
    }
  };
  auto write_a = [&](int stage, const float4 (&r)[4]) {
    unsigned char* base = smem + stage * STG;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
      union { vca_e16 e[4]; uint2 u; } h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h.e[e] = (vca_e16)v[e];
        l.e[e] = (vca_e16)(v[e] - (float)h.e[e]);
      }
      *reinterpret_cast<uint2*>(base + adst[j]) = h.u;
      if (NSPLIT == 3) *reinterpret_cast<uint2*>(base + AIMG + adst[j]) = l.u;
    }
  };
  // weight stage: [hi | lo] x 128 rows x 64 B, filled by LDS-DMA in 1 KiB pieces (16 rows x 64 B); the swizzle is
  // applied to the per-lane source address because the DMA writes LDS in lane order
  constexpr int NPIECE = WSTG / 1024, PPW = NPIECE / 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_w = [&](int c, int stage) {
    const unsigned char* src = a.w + ((size_t)(cbeg + c) * a.npad) * 128;
    unsigned char* wst = smem + stage * STG + ASTG;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pi = wave_u * PPW + i;
      const int part = pi / (GBN / 16), pr = pi - part * (GBN / 16);
      const int row = pr * 16 + (lane >> 2), slot = lane & 3;
      const int grow = min(n0 + row, a.npad - 1);
      const unsigned char* gp = src + (size_t)grow * 128 + part * 64 + ((slot ^ gsw16(row)) << 4);
      unsigned char* lp = wst + part * (GBN * 64) + pr * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                       (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: chunk 0 staged completely; chunks 1 .. GSTAGES - 2: weights in flight, activations in the register sets
  if (nc > 0) {
    fetch_a(0, ra[0]);
    dma_w(0, 0);
    write_a(0, ra[0]);
#pragma unroll
    for (int x = 1; x <= NR; ++x)
      if (x < nc) {
        fetch_a(x, ra[x % NR]);
        dma_w(x, x);
      }
  }
  // chunk 0's weights must have landed.  The counted waits only rely on the DMA pieces (PPW per wave and chunk,
  // issued unconditionally): the activation loads in between are conditional and may count for nothing.
  {
    const int behind = min(NR, max(nc - 1, 0));
    if (behind >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  int aoff[4], woff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wm * 64 + i * 16 + lr;
    aoff[i] = row * 64 + ((lc ^ gsw16(row)) << 4);
    const int wrow = wn * 64 + i * 16 + lr;
    woff[i] = wrow * 64 + ((lc ^ gsw16(wrow)) << 4);
  }

  // one iteration; `rn` holds chunk c + 1's activations (fetched NR iterations ago) and receives chunk c + 1 + NR's
  auto iterate = [&](int c, float4 (&rn)[4]) {
    const int st = c % GSTAGES;
    if (c + 1 < nc) write_a((c + 1) % GSTAGES, rn);      // (the compiler waits for these registers' loads itself)
    if (c + 1 + NR < nc) {
      fetch_a(c + 1 + NR, rn);
      dma_w(c + 1 + NR, (c + 1 + NR) % GSTAGES);         // the stage chunk c - 1 has left
    }
    const unsigned char* ab = smem + st * STG;
    const unsigned char* wb = ab + ASTG;
    bf16x8 wh[4], wl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wh[j] = *reinterpret_cast<const bf16x8*>(wb + woff[j]);
      if (NSPLIT == 3) wl[j] = *reinterpret_cast<const bf16x8*>(wb + GBN * 64 + woff[j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ab + aoff[i]);
      bf16x8 al;
      if (NSPLIT == 3) al = *reinterpret_cast<const bf16x8*>(ab + AIMG + aoff[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (NSPLIT == 3) {
          acc[i][j] = VCA_MFMA_16x16x32(wl[j], ah, acc[i][j]);
          acc[i][j] = VCA_MFMA_16x16x32(wh[j], al, acc[i][j]);
        }
        acc[i][j] = VCA_MFMA_16x16x32(wh[j], ah, acc[i][j]);
      }
    }
    // chunk c + 1's weights must have landed; the DMA pieces of the chunks issued behind them (c + 2 .. c + 1 + NR,
    // as far as they exist) may fly on
    const int behind = min(NR, nc - 2 - c);
    if (behind >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  for (int c = 0; c < nc; c += NR) {      // chunk c + r + 1 sits in register set (c + r + 1) % NR = (r + 1) % NR
#pragma unroll
    for (int r = 0; r < NR; ++r)
      if (c + r < nc) iterate(c + r, ra[(r + 1) % NR]);
  }

  // epilogue: D[row][col]: row = 4 lc + r (column n within the sub-tile), col = lr (row m)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + lr;
    if (m >= d.M) continue;
    const size_t mo = (size_t)(m / d.c_tm) * d.c_bs + (size_t)(m % d.c_tm) * d.c_ts;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * lc;
      if (n >= d.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (g.ksplit > 1) {                       // plain [z][M][N] slab (N a multiple of 4 is required for this path)
        float* p = g.part + ((size_t)kz * d.M + m) * d.N + n;
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        continue;
      }
      const size_t o = mo + (size_t)(n / d.c_cn) * d.c_ns + (n % d.c_cn);
      if (a.vec_store) {
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

// C = sum of the ksplit slabs (+ bias) (+ res), written through C's strided view; one float4 of a row per thread
__global__ __launch_bounds__(256) void gemm_fold_kernel(const GemmArgs g) {
  const LinArgs& a = g.l;
  const vca_lin_desc& d = a.d;
  const int n4 = d.N >> 2;
  const size_t total = (size_t)d.M * n4, slab = (size_t)d.M * d.N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / n4), n = (int)(i - (size_t)m * n4) * 4;
    float4 v = *reinterpret_cast<const float4*>(g.part + (size_t)m * d.N + n);
    for (int z = 1; z < g.ksplit; ++z) {
      const float4 t = *reinterpret_cast<const float4*>(g.part + z * slab + (size_t)m * d.N + n);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (a.bias) {
      const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    }
    const size_t o = (size_t)(m / d.c_tm) * d.c_bs + (size_t)(m % d.c_tm) * d.c_ts + (size_t)(n / d.c_cn) * d.c_ns +
                     (n % d.c_cn);
    if (a.res) {
      const float4 rv = *reinterpret_cast<const float4*>(a.res + o);
      v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
    }
    *reinterpret_cast<float4*>(a.c + o) = v;
  }
}

// reduction ranges for a tile count: enough workgroups for the chip, at least 4 chunks per range
int gemm_ksplit(long long tiles, int nchunk) {
  int ks = 1;
  while (tiles * ks < 200 && nchunk / (ks + 1) >= 4 && ks < 16) ++ks;
  return ks;
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

static int gemm_fill(const vca_lin_desc* dp, int nsplit, const float* a, const void* wpack, const float* bias,
                     const float* res, float* c, GemmArgs& g, long long& tiles) {
  VCA_CHECK_ARG(dp && (nsplit == 1 || nsplit == 3));
  const vca_lin_desc& d = *dp;
  VCA_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0 && (d.K % 8) == 0 && d.a_tm > 0 && d.c_tm > 0 && d.a_ck > 0 && d.c_cn > 0);
  VCA_CHECK_ARG((d.a_ck % 32) == 0 || d.a_ck >= d.K);
  VCA_CHECK_ARG((d.a_bs % 4) == 0 && (d.a_ts % 4) == 0 && (d.a_ks % 4) == 0);
  LinArgs& l = g.l;
  l.d = d;
  if (d.a_ck >= d.K) l.d.a_ck = (d.K + 31) / 32 * 32 + 32;
  l.a = a;
  l.w = reinterpret_cast<const unsigned char*>(wpack);
  l.bias = bias;
  l.res = res;
  l.c = c;
  l.npad = (d.N + 31) / 32 * 32;
  l.nchunk = (d.K + 31) / 32;
  l.ntile_n = (d.N + GBN - 1) / GBN;
  tiles = (long long)((d.M + GBM - 1) / GBM) * l.ntile_n;
  VCA_CHECK_ARG(tiles < (1ll << 30));
  l.ntiles = (int)tiles;
  l.ksplit = 0;
  l.vec_store = ((d.N % 4) == 0 && (d.c_cn % 4) == 0 && (d.c_bs % 4) == 0 && (d.c_ts % 4) == 0 && (d.c_ns % 4) == 0 &&
                 (reinterpret_cast<uintptr_t>(c) & 15) == 0 && (!res || (reinterpret_cast<uintptr_t>(res) & 15) == 0) &&
                 (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0)) ? 1 : 0;
  // the split path moves float4s of plain [M][N] slabs and folds through the float4 path
  g.ksplit = l.vec_store ? gemm_ksplit(tiles, l.nchunk) : 1;
  g.cpz = (l.nchunk + g.ksplit - 1) / g.ksplit;
  g.part = nullptr;
  return 0;
}

extern "C" long long vca_gemm_bf16_ws_floats(const vca_lin_desc* d) {
  GemmArgs g;
  long long tiles = 0;
  if (gemm_fill(d, 3, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<float*>(16), g, tiles) != 0) return 0;
  return g.ksplit > 1 ? (long long)g.ksplit * d->M * d->N : 0;
}

extern "C" int vca_gemm_bf16(const vca_lin_desc* dp, int nsplit, const float* a, const void* wpack, const float* bias,
                             const float* res, float* c, float* ws, long long ws_floats, void* stream) {
  VCA_CHECK_ARG(dp && a && wpack && c);
  VCA_CHECK_ARG((reinterpret_cast<uintptr_t>(a) & 15) == 0 && (reinterpret_cast<uintptr_t>(wpack) & 15) == 0);
  GemmArgs g;
  long long tiles = 0;
  const int rc = gemm_fill(dp, nsplit, a, wpack, bias, res, c, g, tiles);
  if (rc) return rc;
  if (g.ksplit > 1) {
    VCA_CHECK_ARG(ws && ws_floats >= (long long)g.ksplit * dp->M * dp->N && (reinterpret_cast<uintptr_t>(ws) & 15) == 0);
    g.part = ws;
  }
  // three stages measured 5-8 % shorter than four on every shape of the step (the fourth only lengthens the prologue)
  constexpr int stages = 3;
  const size_t lds = (size_t)stages * (GBM * 64 * (nsplit == 3 ? 2 : 1) + GBN * (nsplit == 3 ? 128 : 64));
  VCA_MAX_LDS((gemm_kernel<3, 3>));
  VCA_MAX_LDS((gemm_kernel<1, 3>));
  const dim3 grid((dp->M + GBM - 1) / GBM, g.l.ntile_n, g.ksplit);
  hipStream_t s = (hipStream_t)stream;
  if (nsplit == 3) hipLaunchKernelGGL((gemm_kernel<3, stages>), grid, dim3(256), lds, s, g);
  else hipLaunchKernelGGL((gemm_kernel<1, stages>), grid, dim3(256), lds, s, g);
  if (g.ksplit > 1) {
    size_t gf = ((size_t)dp->M * (dp->N / 4) + 255) / 256;
    if (gf > 2048) gf = 2048;
    hipLaunchKernelGGL(gemm_fold_kernel, dim3((unsigned)gf), dim3(256), 0, s, g);
  }
  return vca_launch_status();
}
