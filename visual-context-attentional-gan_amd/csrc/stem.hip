// Conv3d stem of the visual front-end (visual_front.py:11): 1 -> 64 channels, 5x7x7 taps, stride (1,2,2), padding
// (2,3,3), on single-channel video [B][T][H][W] -> channels-last [B][T][H/2][W/2][64], split-bf16 MFMA.
//
// One persistent workgroup (8 waves) walks patches of 16x16 output pixels of one frame.  The packed weights (hi/lo bf16,
// [64][35 (kt,kh) rows x 8 kw slots], 74 KiB) are loaded into LDS once per workgroup; per patch the fp32 input halo
// (5 frames x 37 rows x 40 columns) is prefetched into registers under the previous patch's MFMAs and split into
// hi/lo bf16 ONCE when it is written to LDS (each halo element is used by up to 64 (pixel, tap) pairs; converting
// inside the MFMA loop cost a fifth of the kernel's VALU instructions).  The reduction index is laid out as
// (kt, kh, kw') with kw' = kw + 1 in 0..7 (slot 0 has zero weight), so the 8 reduction elements one MFMA lane
// supplies are 8 CONTIGUOUS values of a halo row (x = 2*ox - 4 + kw'): two 8-byte LDS reads per hi and lo image.
// The row offset 2*tx is only 4-byte aligned for odd tx, so every image is kept twice, the second copy shifted by
// two elements: even tx read copy 0, odd tx copy 1, both 8-byte aligned.  No im2col image is ever materialised.  Per-channel sum / sum-of-squares of the outputs (for
// the following train-mode BatchNorm3d, visual_front.py:12) are accumulated in registers across the walk and added
// to `stats` once per workgroup.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int CO = 64, KT = 5, KH = 7, KW = 7;
constexpr int NQ = KT * KH;          // 35 (kt,kh) rows of 8 kw slots
constexpr int KPAD = 288;            // 36 rows * 8 (multiple of 32)
constexpr int WROW = 296;            // bf16 elements per packed weight row (592 B: conflict-free 16-byte reads)
constexpr int PH = 16, PW = 16;      // output patch (8 waves x 32 pixels)
constexpr int NTHR = 512;
constexpr int HR = 2 * (PH - 1) + KH;  // 37 halo rows
constexpr int HC = 40;               // halo columns (2*15 + 7 + 1 = 38, padded)
constexpr int HALO = KT * HR * HC;   // 7400 elements
constexpr int HC2 = 44;              // bf16 row pitch of the LDS images (40 columns + 2 shift, multiple of 4)
constexpr int HIMG = KT * HR * HC2;  // elements of one LDS halo image

__global__ void pack_stem_weight_kernel(const float* __restrict__ w, __bf16* __restrict__ dst) {
  // dst[2][64][WROW]: hi image then lo image
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= CO * WROW) return;
  const int co = i / WROW, k = i - co * WROW;
  float v = 0.f;
  const int q = k >> 3, s = k & 7;
  if (q < NQ && s >= 1) v = w[(co * NQ + q) * KW + (s - 1)];
  const __bf16 h = (__bf16)v;
  dst[i] = h;
  dst[CO * WROW + i] = (__bf16)(v - (float)h);
}

// VEC: W is a multiple of 4, so that a group of 4 halo columns (the halo starts at a multiple of 4) is one aligned
// float4 that lies entirely inside or outside the frame: a quarter of the loads, bounds tests and LDS stores.
template <int NSPLIT, bool VEC>
__global__ __launch_bounds__(NTHR) void stem_fwd_kernel(int B, int T, int H, int W, int OH, int OW,
                                                       const float* __restrict__ x, const __bf16* __restrict__ wp,
                                                       float* __restrict__ y, float* __restrict__ stats,
                                                       int tiles_y, int tiles_x, int npatch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* w_hi = reinterpret_cast<__bf16*>(smem);
  __bf16* w_lo = w_hi + CO * WROW;
  // halo images: [copy 0 | copy 1 (shifted by 2)] x [hi | lo]
  __bf16* h_hi = reinterpret_cast<__bf16*>(smem + (NSPLIT == 3 ? 2 : 1) * CO * WROW * 2);
  __bf16* h_lo = h_hi + 2 * HIMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;

  for (int i = tid; i < (NSPLIT == 3 ? 2 : 1) * CO * WROW / 8; i += NTHR)
    reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(wp)[i];

  // this lane's output pixel inside the patch (GEMM row wave*32 + lr)
  const int ty = wave * 2 + (lr >> 4), tx = lr & 15;
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};

  // halo staging plan of this thread: NL items i = tid + NTHR*j (one element, or 4 consecutive columns with VEC),
  // decomposed once into (kt, row, column)
  constexpr int EPI = VEC ? 4 : 1;                        // elements per item
  constexpr int NITEM = HALO / EPI;
  constexpr int NL = (NITEM + NTHR - 1) / NTHR;
  int plan[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int i = tid + NTHR * j;
    const int c = (i % (HC / EPI)) * EPI, rr = i / (HC / EPI);
    plan[j] = i < NITEM ? ((rr / HR) | ((rr % HR) << 8) | (c << 16)) : -1;
  }
  float hv[NL][EPI];
  auto issue = [&](int patch) {   // all loads of the patch's halo are issued back to back (kept in flight)
    int r = patch;
    const int tix = r % tiles_x; r /= tiles_x;
    const int tiy = r % tiles_y; r /= tiles_y;
    const int t = r % T;
    const int b = r / T;
    const int y0 = 2 * tiy * PH - 3, x0 = 2 * tix * PW - 4;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int pl = plan[j];
      const int it = t + (pl & 255) - 2, iy = y0 + ((pl >> 8) & 255), ix = x0 + (pl >> 16);
      const bool ok = pl >= 0 && it >= 0 && it < T && iy >= 0 && iy < H && ix >= 0 && ix < W;
      if constexpr (VEC) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4*>(x + (((size_t)b * T + it) * H + iy) * W + ix);
        hv[j][0] = v.x; hv[j][1] = v.y; hv[j][2] = v.z; hv[j][3] = v.w;
      } else {
        hv[j][0] = ok ? x[(((size_t)b * T + it) * H + iy) * W + ix] : 0.f;
      }
    }
  };

  if ((int)blockIdx.x < npatch) issue(blockIdx.x);
  for (int patch = blockIdx.x; patch < npatch; patch += gridDim.x) {
    int r = patch;
    const int tix = r % tiles_x; r /= tiles_x;
    const int tiy = r % tiles_y; r /= tiles_y;
    const int t = r % T;
    const int b = r / T;
    const int oy0 = tiy * PH, ox0 = tix * PW;
    __syncthreads();   // previous patch fully consumed (and the weights are in place on the first pass)
#pragma unroll
    for (int j = 0; j < NL; ++j)
      if (plan[j] >= 0) {
        const int pl = plan[j];
        const int c = pl >> 16, rr = (pl & 255) * HR + ((pl >> 8) & 255);
        const int o = rr * HC2 + c;
        if constexpr (VEC) {
          // o is a multiple of 4 elements: one 8-byte store into copy 0, two 4-byte stores into the shifted copy
          union { __bf16 e[4]; uint2 u; unsigned int w[2]; } h, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            h.e[e] = (__bf16)hv[j][e];
            l.e[e] = (__bf16)(hv[j][e] - (float)h.e[e]);
          }
          *reinterpret_cast<uint2*>(h_hi + o) = h.u;
          *reinterpret_cast<unsigned int*>(h_hi + HIMG + o + 2) = h.w[0];
          *reinterpret_cast<unsigned int*>(h_hi + HIMG + o + 4) = h.w[1];
          if (NSPLIT == 3) {
            *reinterpret_cast<uint2*>(h_lo + o) = l.u;
            *reinterpret_cast<unsigned int*>(h_lo + HIMG + o + 2) = l.w[0];
            *reinterpret_cast<unsigned int*>(h_lo + HIMG + o + 4) = l.w[1];
          }
        } else {
          const float v = hv[j][0];
          const __bf16 h = (__bf16)v;
          h_hi[o] = h;
          h_hi[HIMG + o + 2] = h;
          if (NSPLIT == 3) {
            const __bf16 l = (__bf16)(v - (float)h);
            h_lo[o] = l;
            h_lo[HIMG + o + 2] = l;
          }
        }
      }
    __syncthreads();
    if (patch + (int)gridDim.x < npatch) issue(patch + gridDim.x);   // next halo streams in under the MFMAs

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

#pragma unroll 2
    for (int ks = 0; ks < KPAD / 16; ++ks) {
      const int q = ks * 2 + lh;            // (kt, kh) row this lane feeds
      union { bf16x8 v; uint2 u[2]; } ah, al;
      ah.u[0] = ah.u[1] = al.u[0] = al.u[1] = make_uint2(0u, 0u);
      if (q < NQ) {
        const int kt = q / KH, kh = q - kt * KH;
        const int o = (tx & 1) * (HIMG + 2) + (kt * HR + 2 * ty + kh) * HC2 + 2 * tx;   // multiple of 4 elements
        ah.u[0] = *reinterpret_cast<const uint2*>(h_hi + o);
        ah.u[1] = *reinterpret_cast<const uint2*>(h_hi + o + 4);
        if (NSPLIT == 3) {
          al.u[0] = *reinterpret_cast<const uint2*>(h_lo + o);
          al.u[1] = *reinterpret_cast<const uint2*>(h_lo + o + 4);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int off = (j * 32 + lr) * WROW + ks * 16 + lh * 8;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(w_hi + off);
        if (NSPLIT == 3) {
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(w_lo + off);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al.v, bh, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bl, acc[j], 0, 0, 0);
        }
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah.v, bh, acc[j], 0, 0, 0);
      }
    }

    // D[row][col]: col = lane&31 (channel), row = (q&3) + 8*(q>>2) + 4*(lane>>5) (pixel within this wave's 32)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * lh;
        const int oy = oy0 + wave * 2 + (row >> 4), ox = ox0 + (row & 15);
        if (oy < OH && ox < OW) {
          const float v = acc[j][q];
          s1[j] += v;
          s2[j] += v * v;
          y[((((size_t)b * T + t) * OH + oy) * OW + ox) * CO + j * 32 + lr] = v;
        }
      }
    }
  }
  if (stats) {
    // the workgroup's waves meet in LDS (the weight image is not needed any more): one atomic per statistic and
    // workgroup instead of one per wave (atomics on one address are resolved one after the other)
    float* red = reinterpret_cast<float*>(smem);   // [waves][2 * CO]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float a = s1[j] + __shfl_xor(s1[j], 32, 64), c = s2[j] + __shfl_xor(s2[j], 32, 64);
      if (lh == 0) {
        red[wave * 2 * CO + j * 32 + lr] = a;
        red[wave * 2 * CO + CO + j * 32 + lr] = c;
      }
    }
    __syncthreads();
    if (tid < 2 * CO) {
      float t = 0.f;
      for (int k = 0; k < NTHR / 64; ++k) t += red[k * 2 * CO + tid];
      atomicAdd(stats + tid, t);
    }
  }
}

}  // namespace

extern "C" int vca_stem_pack_weight(const float* w, void* wpack, void* stream) {
  VCA_CHECK_ARG(w && wpack);
  hipLaunchKernelGGL(pack_stem_weight_kernel, dim3((CO * WROW + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<__bf16*>(wpack));
  return vca_launch_status();
}

extern "C" int vca_stem_fwd_bf16(int nsplit, int B, int T, int H, int W, const float* x, const void* wpack, float* y,
                                 float* stats, void* stream) {
  VCA_CHECK_ARG((nsplit == 1 || nsplit == 3) && B > 0 && T > 0 && H > 0 && W > 0 && x && wpack && y);
  const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
  const int tiles_y = (OH + PH - 1) / PH, tiles_x = (OW + PW - 1) / PW;
  const long long npatch = (long long)B * T * tiles_y * tiles_x;
  VCA_CHECK_ARG(npatch < (1ll << 31));
  const size_t lds = (size_t)(nsplit == 3 ? 2 : 1) * CO * WROW * 2 + (size_t)4 * HIMG * 2 + 16;
  VCA_MAX_LDS((stem_fwd_kernel<3, true>));
  VCA_MAX_LDS((stem_fwd_kernel<1, true>));
  VCA_MAX_LDS((stem_fwd_kernel<3, false>));
  VCA_MAX_LDS((stem_fwd_kernel<1, false>));
  int grid = 256;   // one 8-wave workgroup per CU (LDS: weights 37/74 KiB + halo images 64 KiB)
  if (grid > npatch) grid = (int)npatch;
  const __bf16* wp = reinterpret_cast<const __bf16*>(wpack);
  const bool vec = (W % 4) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
#define LAUNCH_STEM(NS, V)                                                                                          \
  hipLaunchKernelGGL((stem_fwd_kernel<NS, V>), dim3(grid), dim3(NTHR), lds, (hipStream_t)stream, B, T, H, W, OH, OW, \
                     x, wp, y, stats, tiles_y, tiles_x, (int)npatch)
  if (nsplit == 3) {
    if (vec) LAUNCH_STEM(3, true); else LAUNCH_STEM(3, false);
  } else {
    if (vec) LAUNCH_STEM(1, true); else LAUNCH_STEM(1, false);
  }
#undef LAUNCH_STEM
  return vca_launch_status();
}

// Patch form of the clip for the stem's weight gradient: one row of 64 values per output position,
//   xp[b][t][oy*OW + ox][kh*8 + kw] = clip[b][t][2*oy + kh - 3][2*ox + kw - 3]   (zero outside; kh, kw = 7 unused),
// over which the 5x7x7 stride-(1,2,2) convolution is a 5-tap convolution along t with 64 input channels.  In that
// form the weight gradient is one 64x64 tile of 5 taps on the split-bf16 matrix-core kernel instead of 245
// single-channel taps in fp32.
__global__ __launch_bounds__(256) void stem_patch_kernel(int BT, int H, int W, int OH, int OW,
                                                         const float* __restrict__ x, float* __restrict__ xp) {
  const size_t total = (size_t)BT * OH * OW * 16;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int q = (int)(i & 15);                 // kh = q >> 1, kw = (q & 1) * 4 .. +3
    size_t r = i >> 4;
    const int ox = (int)(r % OW);
    r /= OW;
    const int oy = (int)(r % OH);
    const size_t img = r / OH;
    const int kh = q >> 1, kw0 = (q & 1) * 4;
    const int ys = 2 * oy + kh - 3;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (kh < 7 && ys >= 0 && ys < H) {
      const float* row = x + (img * H + ys) * W;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xs = 2 * ox + kw0 + j - 3;
        if (kw0 + j < 7 && xs >= 0 && xs < W) v[j] = row[xs];
      }
    }
    *reinterpret_cast<float4*>(xp + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

extern "C" int vca_stem_patches_f32(int B, int T, int H, int W, const float* x, float* xp, void* stream) {
  VCA_CHECK_ARG(B > 0 && T > 0 && H > 0 && W > 0 && x && xp);
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  size_t n = (size_t)B * T * OH * OW * 16, g = (n + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(stem_patch_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, B * T, H, W, OH, OW, x,
                     xp);
  return vca_launch_status();
}
