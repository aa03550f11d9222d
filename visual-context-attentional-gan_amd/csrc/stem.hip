// Conv3d stem of the visual front-end (visual_front.py:11): 1 -> 64 channels, 5x7x7 taps, stride (1,2,2), padding
// (2,3,3), on single-channel video [B][T][H][W] -> channels-last [B][T][H/2][W/2][64], split-bf16 MFMA.
//
// A workgroup (4 waves) owns one 16x16 patch of output pixels of one clip and WALKS a chunk of consecutive frames.
// LDS holds a ring of five input-frame halos (37 rows x 40 columns each, hi/lo bf16): per output frame one new input
// frame is staged (its raw fp32 values were fetched into registers under the previous frame's MFMAs and are split into
// hi/lo ONCE when written to LDS) and the five temporal taps read the ring, so a frame is fetched once per patch and
// chunk instead of once per temporal tap.  The weights never touch LDS: a wave owns 32 of the 64 output channels for
// half of the patch and keeps their hi/lo bf16 fragments for all 18 reduction steps in registers (144 VGPRs), which
// leaves room for two workgroups per CU (74 KiB of LDS each).
//
// The reduction index is laid out as (kt, kh, kw') with kw' = kw + 1 in 0..7 (slot 0 has zero weight), so the 8
// reduction elements one MFMA lane supplies are 8 CONTIGUOUS values of a halo row (x = 2*ox - 4 + kw'): two 8-byte LDS
// reads per hi and lo image.  The row offset 2*tx is only 4-byte aligned for odd tx, so every image is kept twice, the
// second copy shifted by two elements: even tx read copy 0, odd tx copy 1, both 8-byte aligned.  The 32 lanes that
// share an LDS cycle are the 16 x 2 pixels of a tile reading the same halo row; with a row pitch of 48 elements (the
// second pixel row lands 48 dwords further) and copy 1 placed 30 dwords (mod 64) behind copy 0 the four 16-dword runs
// they touch tile the 64 banks exactly.  No im2col image is ever materialised.  The MFMAs take the pixels as the row
// operand, so a lane ends up with ONE channel of 16 pixels: a store instruction writes whole 128-byte lines and the
// batch-norm sums are two registers per lane.  Per-channel sum / sum-of-squares of the outputs (for the following train-mode BatchNorm3d, visual_front.py:12) are
// accumulated in registers across the walk and added to one of `stat_slots` partial buffers once per workgroup.
#include "common.h"

namespace {


constexpr int CO = 64, KT = 5, KH = 7, KW = 7;
constexpr int NQ = KT * KH;          // 35 (kt,kh) rows of 8 kw slots
constexpr int NSTEP = 18;            // reduction steps of 16 (two rows each; row 35 is padding)
constexpr int PH = 16, PW = 16;      // output patch
constexpr int NTHR = 256;
constexpr int HR = 2 * (PH - 1) + KH;  // 37 halo rows
constexpr int HC = 40;               // halo columns staged (2*15 + 7 + 1 = 38, in groups of 4)
constexpr int HP = 48;               // bf16 row pitch of an LDS image (24 dwords)
constexpr int IMG = 1852;            // elements per image: 37 * 48 = 1776, padded to 926 dwords = 30 (mod 64)
constexpr int NRING = 5;
constexpr int TCHUNK = 15;           // output frames per workgroup (+4 frames staged without output)
constexpr int WFRAG_U4 = 2 * NSTEP * 2 * 64;   // uint4 entries of the packed weight fragments

// dst[channel half][step][hi|lo][lane] = the 8 bf16 a lane feeds as the row operand of that step's MFMA:
// lane (r = lane & 31, h = lane >> 5) holds channel 32*half + r, reduction row q = 2*step + h, slots kw' = 0..7
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, vca_e16* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * NSTEP * 64) return;
  const int lane = i & 63, s = (i >> 6) % NSTEP, half = i / (64 * NSTEP);
  const int co = half * 32 + (lane & 31), q = 2 * s + (lane >> 5);
  vca_e16* hi = dst + ((size_t)((half * NSTEP + s) * 2 + 0) * 64 + lane) * 8;
  vca_e16* lo = dst + ((size_t)((half * NSTEP + s) * 2 + 1) * 64 + lane) * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = 0.f;
    if (q < NQ && j >= 1) v = w[(co * NQ + q) * KW + (j - 1)];
    const vca_e16 hv = (vca_e16)v;
    hi[j] = hv;
    lo[j] = (vca_e16)(v - (float)hv);
  }
}

union Frag {
  bf16x8 v;
  unsigned long long d[2];
};

// One fragment (8 bf16 of the hi image and, with three products, of the lo image 2 * IMG elements further) as 8-byte
// LDS reads; addr is the byte address of the hi fragment.
template <int NSPLIT>
__device__ __forceinline__ void lds_read_frag(unsigned addr, Frag& h, Frag& l) {
  asm volatile("ds_read_b64 %0, %1" : "=v"(h.d[0]) : "v"(addr));
  asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(h.d[1]) : "v"(addr));
  if (NSPLIT == 3) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(l.d[0]) : "v"(addr), "n"(2 * IMG * 2));
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(l.d[1]) : "v"(addr), "n"(2 * IMG * 2 + 8));
  }
}

// Wait until the fragment's reads have landed while the reads of the BEHIND fragments requested after it (2 or 4
// instructions each) stay in flight.  LDS operations complete in order, so "all but the youngest N" covers it; the
// wait names the fragment's registers so that the MFMAs that consume them depend on it.
template <int NSPLIT, int BEHIND>
__device__ __forceinline__ void lds_wait_frag(Frag& h, Frag& l) {
  constexpr int N = BEHIND * (NSPLIT == 3 ? 4 : 2);
  if (NSPLIT == 3)
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(h.d[0]), "+v"(h.d[1]), "+v"(l.d[0]), "+v"(l.d[1]) : "n"(N));
  else
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(h.d[0]), "+v"(h.d[1]) : "n"(N));
}

constexpr int AHEAD = 2;             // fragment reads are issued this many reduction steps ahead of their MFMAs

// VEC: W is a multiple of 4, so that a group of 4 halo columns (the halo starts at a multiple of 4) is one aligned
// float4 that lies entirely inside or outside the frame.
template <int NSPLIT, bool VEC>
__global__ __launch_bounds__(NTHR, 2) void stem_fwd_kernel(int B, int T, int H, int W, int OH, int OW,
                                                          const float* __restrict__ x,
                                                          const uint4* __restrict__ wfrag, float* __restrict__ y,
                                                          float* __restrict__ stats, int stat_slots, int tiles_y,
                                                          int tiles_x, int tchunks, int nunits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NIMG = NSPLIT == 3 ? 4 : 2;       // [hi copy 0 | hi copy 1 | lo copy 0 | lo copy 1]
  constexpr int SLOT = NIMG * IMG;                // elements per ring slot
  vca_e16* ring = reinterpret_cast<vca_e16*>(smem);
  const unsigned ring_addr = (unsigned)(uintptr_t)smem;      // LDS byte address (the low half of the flat address)

  // Work unit of this workgroup.  Consecutive workgroups are dealt round-robin over the 8 XCDs, so units that share
  // input (neighbouring patches and chunks of one clip) are given to workgroups of equal blockIdx % 8: one L2.
  const int per = (int)gridDim.x >> 3;
  const int unit = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (unit >= nunits) return;
  int u = unit;
  const int tc = u % tchunks; u /= tchunks;
  const int tix = u % tiles_x; u /= tiles_x;
  const int tiy = u % tiles_y;
  const int b = u / tiles_y;
  const int t0 = tc * TCHUNK, t1 = min(T, t0 + TCHUNK);
  const int oy0 = tiy * PH, ox0 = tix * PW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = wave & 1, prow0 = (wave >> 1) * 8;     // channel half; first patch row of this wave's 8 rows
  const int lr = lane & 31, lh = lane >> 5;
  const int tx = lr & 15, tyl = lr >> 4;

  // weight fragments of this wave's 32 channels: registers for the whole walk
  bf16x8 wh[NSTEP], wl[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    union { uint4 u; bf16x8 v; } c;
    c.u = wfrag[((half * NSTEP + s) * 2 + 0) * 64 + lane];
    wh[s] = c.v;
    if (NSPLIT == 3) {
      c.u = wfrag[((half * NSTEP + s) * 2 + 1) * 64 + lane];
      wl[s] = c.v;
    }
  }

  // halo staging plan of this thread: items i = tid, tid + 256 (a group of 4 columns of one halo row)
  constexpr int NITEM = HR * (HC / 4);            // 370
  constexpr int NL = (NITEM + NTHR - 1) / NTHR;   // 2
  const int y0 = 2 * oy0 - 3, x0 = 2 * ox0 - 4;
  int lds_o[NL];
  int goff[NL];                                   // element offset inside a frame, or -1 (a frame holds < 2^31 elements)
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int i = tid + NTHR * j;
    const int row = i / (HC / 4), c4 = (i - row * (HC / 4)) * 4;
    const int iy = y0 + row, ix = x0 + c4;
    lds_o[j] = i < NITEM ? row * HP + c4 : -1;
    bool ok = i < NITEM && iy >= 0 && iy < H;
    if (VEC) ok = ok && ix >= 0 && ix < W;
    goff[j] = ok ? iy * W + ix : -1;
  }
  const float* xb = x + (size_t)b * T * H * W;
  const size_t fstride = (size_t)H * W;
  float hv[NL][4];
  auto fetch = [&](int f) {       // raw fp32 values of input frame f into registers (zeros outside the clip)
    const bool fin = f >= 0 && f < T;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      hv[j][0] = hv[j][1] = hv[j][2] = hv[j][3] = 0.f;
      if (fin && goff[j] >= 0) {
        const float* p = xb + (size_t)f * fstride + goff[j];
        if constexpr (VEC) {
          const float4 v = *reinterpret_cast<const float4*>(p);
          hv[j][0] = v.x; hv[j][1] = v.y; hv[j][2] = v.z; hv[j][3] = v.w;
        } else {
          const int ix = x0 + (lds_o[j] % HP);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ix + e >= 0 && ix + e < W) hv[j][e] = p[e];
        }
      }
    }
  };
  auto stage = [&](int f) {       // split the registers into hi/lo and write both copies of frame f's ring slot
    vca_e16* base = ring + ((f + 2 * NRING) % NRING) * SLOT;
#pragma unroll
    for (int j = 0; j < NL; ++j)
      if (lds_o[j] >= 0) {
        const int o = lds_o[j];     // multiple of 4 elements: one 8-byte store into copy 0, two 4-byte stores into copy 1
        union { vca_e16 e[4]; uint2 u; unsigned int w[2]; } h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h.e[e] = (vca_e16)hv[j][e];
          l.e[e] = (vca_e16)(hv[j][e] - (float)h.e[e]);
        }
        *reinterpret_cast<uint2*>(base + o) = h.u;
        *reinterpret_cast<unsigned int*>(base + IMG + o + 2) = h.w[0];
        *reinterpret_cast<unsigned int*>(base + IMG + o + 4) = h.w[1];
        if (NSPLIT == 3) {
          *reinterpret_cast<uint2*>(base + 2 * IMG + o) = l.u;
          *reinterpret_cast<unsigned int*>(base + 3 * IMG + o + 2) = l.w[0];
          *reinterpret_cast<unsigned int*>(base + 3 * IMG + o + 4) = l.w[1];
        }
      }
  };

  // this lane's read position inside an image: pixel (patch row prow0 + 2*tile + tyl, column tx), copy by parity of tx
  const int lanebase = (tx & 1) * (IMG + 2) + 2 * tx + 2 * (prow0 + tyl) * HP;
  float s1 = 0.f, s2 = 0.f;                     // sum / sum of squares of this lane's channel over its pixels

  for (int f = t0 - 2; f < t0 + 2; ++f) {       // the four frames in front of the first new one
    fetch(f);
    stage(f);
  }
  fetch(t0 + 2);
  for (int t = t0; t < t1; ++t) {
    __syncthreads();                            // frame t-3's slot has been read for the last time
    stage(t + 2);
    if (t + 1 < t1) fetch(t + 3);               // streams in under this frame's MFMAs
    __syncthreads();
    int slot_o[KT];                             // ring slot (element offset) of temporal tap kt: input frame t + kt - 2
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) slot_o[kt] = ((t + kt - 2 + 2 * NRING) % NRING) * SLOT;
    float* yt = y + (((size_t)b * T + t) * OH) * OW * CO + half * 32 + lr;
    // byte offset of this lane's row for reduction step s: rows q = 2s (lanes 0-31) and 2s + 1 (lanes 32-63)
    auto row_off = [&](int s) {
      const int q0 = 2 * s, q1 = (2 * s + 1 < NQ) ? 2 * s + 1 : 2 * s;   // row 35 is padding: any finite data
      const int o0 = slot_o[q0 / KH] + (q0 % KH) * HP, o1 = slot_o[q1 / KH] + (q1 % KH) * HP;
      return 2 * (lh ? o1 : o0);
    };
#pragma unroll 1
    for (int tile = 0; tile < 4; ++tile) {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const unsigned tb = ring_addr + 2u * (unsigned)(lanebase + 4 * tile * HP);
      // The fragment reads are written as single ds_read_b64 instructions: left to the compiler, the two 8-byte halves
      // of a fragment become one ds_read2_b64, which the LDS serves in four groups of 16 lanes instead of two groups
      // of 32 (half the rate, and the bank layout above is built for the 32-lane groups).  They are issued AHEAD
      // steps in front of the MFMAs that consume them.
      Frag fh[AHEAD + 1], fl[AHEAD + 1];
#pragma unroll
      for (int s = 0; s < AHEAD; ++s) lds_read_frag<NSPLIT>(tb + row_off(s), fh[s], fl[s]);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        constexpr int NB = AHEAD + 1;
        if (s + AHEAD < NSTEP) lds_read_frag<NSPLIT>(tb + row_off(s + AHEAD), fh[(s + AHEAD) % NB], fl[(s + AHEAD) % NB]);
        // fragments of the steps behind this one that are already requested may stay in flight
        const int behind = (NSTEP - 1 - s) < AHEAD ? (NSTEP - 1 - s) : AHEAD;
        if (behind == 0) lds_wait_frag<NSPLIT, 0>(fh[s % NB], fl[s % NB]);
        else if (behind == 1) lds_wait_frag<NSPLIT, 1>(fh[s % NB], fl[s % NB]);
        else if (behind == 2) lds_wait_frag<NSPLIT, 2>(fh[s % NB], fl[s % NB]);
        else lds_wait_frag<NSPLIT, 3>(fh[s % NB], fl[s % NB]);
        // pixels are the row operand: D[row = pixel][col = channel], so a lane owns one channel, a register quad four
        // pixels, and every store instruction writes whole 128-byte lines (32 channels of a pixel)
        if (NSPLIT == 3) {
          acc = VCA_MFMA_32x32x16(fh[s % NB].v, wl[s], acc);
          acc = VCA_MFMA_32x32x16(fl[s % NB].v, wh[s], acc);
        }
        acc = VCA_MFMA_32x32x16(fh[s % NB].v, wh[s], acc);
      }
      // D[row][col]: col = lane & 31 (channel within the half), row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5) = pixel
      // of the tile (column row & 15 of patch row prow0 + 2 * tile + (row >> 4))
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * lh;
        const int oy = oy0 + prow0 + 2 * tile + (row >> 4), ox = ox0 + (row & 15);
        if (oy < OH && ox < OW) {
          const float v = acc[i];
          s1 += v;
          s2 += v * v;
          yt[((size_t)oy * OW + ox) * CO] = v;
        }
      }
    }
  }
  if (stats) {
    // the two lane halves hold the same channel (different pixels), and so do the two waves of a channel half: they
    // meet in LDS (the ring is not needed any more); one atomic per statistic and workgroup into a partial buffer
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    float* red = reinterpret_cast<float*>(smem);   // [pixel half][2 * CO]
    __syncthreads();
    if (lh == 0) {
      red[(wave >> 1) * 2 * CO + half * 32 + lr] = s1;
      red[(wave >> 1) * 2 * CO + CO + half * 32 + lr] = s2;
    }
    __syncthreads();
    if (tid < 2 * CO) atomicAdd(stats + (size_t)(unit % stat_slots) * 2 * CO + tid, red[tid] + red[2 * CO + tid]);
  }
}

}  // namespace

extern "C" long long vca_stem_pack_bytes() { return (long long)WFRAG_U4 * 16; }

extern "C" int vca_stem_pack_weight(const float* w, void* wpack, void* stream) {
  VCA_CHECK_ARG(w && wpack);
  hipLaunchKernelGGL(pack_stem_weight_kernel, dim3((2 * NSTEP * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<vca_e16*>(wpack));
  return vca_launch_status();
}

extern "C" int vca_stem_fwd_bf16(int nsplit, int B, int T, int H, int W, const float* x, const void* wpack, float* y,
                                 float* stats, int stat_slots, void* stream) {
  VCA_CHECK_ARG((nsplit == 1 || nsplit == 3) && B > 0 && T > 0 && H > 0 && W > 0 && x && wpack && y);
  VCA_CHECK_ARG(stats == nullptr || stat_slots >= 1);
  VCA_CHECK_ARG((reinterpret_cast<uintptr_t>(wpack) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0);
  const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
  const int tiles_y = (OH + PH - 1) / PH, tiles_x = (OW + PW - 1) / PW;
  const int tchunks = (T + TCHUNK - 1) / TCHUNK;
  const long long nunits = (long long)B * tiles_y * tiles_x * tchunks;
  VCA_CHECK_ARG(nunits < (1ll << 28) && (long long)H * W < (1ll << 31) - 8);
  const size_t lds = (size_t)NRING * (nsplit == 3 ? 4 : 2) * IMG * 2;
  VCA_MAX_LDS((stem_fwd_kernel<3, true>));
  VCA_MAX_LDS((stem_fwd_kernel<1, true>));
  VCA_MAX_LDS((stem_fwd_kernel<3, false>));
  VCA_MAX_LDS((stem_fwd_kernel<1, false>));
  const int grid = (int)((nunits + 7) / 8) * 8;
  const uint4* wp = reinterpret_cast<const uint4*>(wpack);
  const bool vec = (W % 4) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
#define LAUNCH_STEM(NS, V)                                                                                          \
  hipLaunchKernelGGL((stem_fwd_kernel<NS, V>), dim3(grid), dim3(NTHR), lds, (hipStream_t)stream, B, T, H, W, OH, OW, \
                     x, wp, y, stats, stat_slots, tiles_y, tiles_x, tchunks, (int)nunits)
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
