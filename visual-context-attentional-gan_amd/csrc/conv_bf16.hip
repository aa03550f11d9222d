// Split-bf16 implicit-GEMM convolution for gfx950: forward / input gradient on v_mfma_f32_16x16x32_bf16, weight
// gradients on v_mfma_f32_32x32x16_bf16.
//
// Precision: every fp32 operand v is split as v = hi + lo with hi = bf16(v), lo = bf16(v - hi); a product is
// accumulated in fp32 as hi*hi + hi*lo + lo*hi ("bf16x3", ~2^-17 relative error per product, which keeps the
// generated mel-spectrograms within 1e-4 of the fp32 reference; a single bf16 product does not: 2e-2).
// NSPLIT = 1 computes hi*hi only.
//
// Tiling: a workgroup owns a patch of NI images x TH x TW output pixels (<= 128 rows of the GEMM) and BN output
// channels.  For each slice of 32 input channels the input HALO of the patch ((TH-1)*stride+KH) x
// ((TW-1)*stride+KW) positions is staged in LDS once (fp32 -> prologue -> hi/lo bf16) and reused by every
// filter tap: the A fragment of tap (kh,kw) is the same LDS image read at a shifted position, so activation
// traffic from L2 does not scale with the number of taps.  The weights of one (tap, slice) are streamed through
// a double-buffered LDS tile.  LDS rows are 64 B (32 bf16) with XOR-swizzled 16-byte chunks (the swizzle depends on
// the kernel's fragment shape, see below) so that the fragment reads are free of bank conflicts.
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace {


constexpr int BM = 128;
constexpr int MAXPOS = 384;  // halo positions per workgroup (hi + lo images: 48 KiB)

struct HArgs {
  vca_conv_desc d;
  const float* x;
  const unsigned char* w;  // packed [tap][chunk][npad][hi 32 | lo 32] bf16
  const float* bias;
  const float* in_scale;
  const float* in_shift;
  const float* in_alpha;
  const float* res;
  float* y;
  float* stats;
  int npad, nchunk, ntaps;
  int NI, TH, TW, HH, HW, NPOS;   // HW: pitch of a halo row in positions, NPOS: pitch of a slot's halo (both may be padded)
  int HWv;                        // halo positions per row that are actually staged (<= HW)
  int tiles_x, tiles_y, nsub;
  int vecA;
  int vecY;    // y, res, bias and the split workspace are 16-byte aligned: the epilogue moves float4s
  int ksplit;  // >1: blockIdx.z owns a range of channel slices; its partial sums go to slab blockIdx.z of `part`
  float* part; //     (plain stores, folded into y by sum_slabs_kernel) or, without a workspace, into y with atomics
  unsigned part_stride;   // floats per slab
};

// A workgroup's GEMM rows are NI "slots" of TH x TW output pixels.  Slots are consecutive entries of the list of all
// (image, row band, column band) sub-patches, so one workgroup can hold several images of a small map or several
// bands of one image; every slot has its own halo region in LDS.
__device__ __forceinline__ bool slot_origin(int sid, int nsub, int tiles_x, int tiles_y, int TH, int TW, int& n,
                                            int& oy0, int& ox0) {
  const int per_img = tiles_x * tiles_y;
  n = sid / per_img;
  const int r = sid - n * per_img;
  const int tiy = r / tiles_x;
  oy0 = tiy * TH;
  ox0 = (r - tiy * tiles_x) * TW;
  return sid < nsub;
}

__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo) {
  bf16x8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    h[j] = (vca_e16)v[j];
    l[j] = (vca_e16)(v[j] - (float)h[j]);
  }
  hi = *reinterpret_cast<uint4*>(&h);
  lo = *reinterpret_cast<uint4*>(&l);
}

// Matrix instruction: v_mfma_f32_16x16x32_bf16.  One instruction covers the whole 32-channel slice of a 16 x 16
// output tile (lane l feeds row / column l%16 with the 8 channels of 16-byte chunk l/16), so a (tap, slice) step of
// a 64 x 64 wave tile is 16 MFMAs of 16 cycles.  The 32x32x16 form (8 MFMAs of 32 cycles, the same LDS traffic,
// cycles and accumulator registers) was this kernel's first version: on this part the 16x16x32 form holds a higher
// clock under load, and the same loop ran 8 % (split product) to 22 % (single product) faster summed over the layers
// of a training step.  The LDS images use the chunk swizzle c ^ (((row >> 2) & 1) << 1), which makes this
// lane -> (row, chunk) pattern conflict-free for ds_read_b128 over any 16 consecutive rows.
//
// Measured on the 32x32x16 version and rejected (none moved the 128-channel launches by more than 3 %, the narrow
// ones lost up to 40 % through LDS residency): 3 / 6 / 12 filter taps per barrier for the single product, a ring of
// three weight buffers requested two taps ahead, 256 output channels per workgroup, register double-buffered
// fragments, fetching the next slice's halo a stage ahead into registers.  Timing probes: without the halo loads a
// launch is 17-19 % shorter, without the weight tiles 11-14 %.
__device__ __forceinline__ int sw16(int row) { return ((row >> 2) & 1) << 1; }

// x / d for 0 <= x < 1024 and 1 <= d <= 512 from inv = 1.0f / d: four full-rate instructions instead of the ~40 of an
// integer division.  (x + 0.5) / d is at least 0.5 / d >= 9.7e-4 away from an integer, the two fp32 roundings move the
// product by less than 3e-4.
__device__ __forceinline__ int small_div(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

template <int BN, int WM, int WN, int NSPLIT>
__global__ __launch_bounds__(256, 2) void conv_bf16_halo_kernel(HArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NT = WTN / 16;
  constexpr int BROW = (NSPLIT == 3) ? 128 : 64;
  constexpr int BBUF = BN * BROW;

  const vca_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 15, lc = lane >> 4;
  const int npos_total = a.NI * a.NPOS;
  unsigned char* halo_hi = smem;
  unsigned char* halo_lo = smem + npos_total * 64;
  unsigned char* bbase = smem + npos_total * 64 * (NSPLIT == 3 ? 2 : 1);

  const int sid0 = blockIdx.x * a.NI;
  const int n0 = blockIdx.y * BN;
  const int He = d.Hin << d.ups, We = d.Win << d.ups;
  const int tpix = a.TH * a.TW;

  const float inv_tpix = 1.0f / (float)tpix, inv_tw = 1.0f / (float)a.TW;
  int hb[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = wm * WTM + i * 16 + lr;
    const int img = small_div(m, inv_tpix), rem = m - img * tpix;
    const int ty = small_div(rem, inv_tw), tx = rem - ty * a.TW;
    hb[i] = img < a.NI ? img * a.NPOS + ty * d.stride * a.HW + tx * d.stride : 0;
  }
  int boff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int row = wn * WTN + j * 16 + lr;
    boff[j] = row * 64 + ((lc ^ sw16(row)) << 4);
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  constexpr int NPIECE = BN * BROW / 1024;
  constexpr int PPW = (NPIECE + 3) / 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue_b = [&](int tap, int chunk, int buf) {
    const unsigned char* src = a.w + ((size_t)(tap * a.nchunk + chunk) * a.npad) * 128;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pi = wave_u * PPW + i;
      if (pi < NPIECE) {
        const int part = pi / (BN / 16), pr = pi - part * (BN / 16);
        const int row = pr * 16 + (lane >> 2), slot = lane & 3;
        const int grow = min(n0 + row, a.npad - 1);
        const unsigned char* g = src + (size_t)grow * 128 + part * 64 + ((slot ^ sw16(row)) << 4);
        unsigned char* l = bbase + buf * BBUF + part * (BN * 64) + pr * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
      }
    }
  };
  auto wait_b = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  // Source offset of each halo item of this thread, the same for every channel slice.  The slots' (image, first input
  // row, first input column, valid) are worked out once per workgroup, in the still idle weight buffers; the
  // position -> (slot, halo row, halo column) split of an item uses small_div.
  constexpr int NHI = (2 * MAXPOS + 255) / 256;
  int4* slot_tab = reinterpret_cast<int4*>(bbase);
  if (tid < a.NI) {
    int n, oy0, ox0;
    const bool ok = slot_origin(sid0 + tid, a.nsub, a.tiles_x, a.tiles_y, a.TH, a.TW, n, oy0, ox0);
    slot_tab[tid] = make_int4(n, oy0 * d.stride - d.pad_h, ox0 * d.stride - d.pad_w, ok ? 1 : 0);
  }
  __syncthreads();
  const float inv_npos = 1.0f / (float)a.NPOS, inv_hw = 1.0f / (float)a.HW;
  long long hsrc[NHI];
#pragma unroll
  for (int k = 0; k < NHI; ++k) {
    const int it = tid + k * 256;
    hsrc[k] = -1;
    if (it < npos_total * 2) {
      const int pos = it >> 1;
      const int img = small_div(pos, inv_npos), rp = pos - img * a.NPOS;
      const int hy = small_div(rp, inv_hw), hx = rp - hy * a.HW;
      if (hx >= a.HWv || hy >= a.HH) {   // padding of the row / slot pitch: never read by a fragment, never staged
        hsrc[k] = -2;
        continue;
      }
      const int4 st = slot_tab[img];
      const int n = st.x;
      int vy = st.y + hy, vx = st.z + hx;
      bool ok = st.w != 0 && vy >= 0 && vx >= 0;
      if (d.dil > 1) {
        ok = ok && (vy % d.dil == 0) && (vx % d.dil == 0);
        vy /= d.dil;
        vx /= d.dil;
      }
      ok = ok && vy < He && vx < We;
      vy >>= d.ups;
      vx >>= d.ups;
      if (ok) hsrc[k] = ((long long)(n * d.Hin + vy) * d.Win + vx) * d.Cin;
    }
  }

  auto stage_halo = [&](int chunk) {
    float4 pre[NHI][4];
#pragma unroll
    for (int k = 0; k < NHI; ++k) {
      const int it = tid + k * 256;
      const int ci0 = chunk * 32 + (it & 1) * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) pre[k][q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < npos_total * 2 && hsrc[k] >= 0 && ci0 < d.Cin) {
        const float* src = a.x + hsrc[k] + ci0;
        if (a.vecA && ci0 + 16 <= d.Cin) {
#pragma unroll
          for (int q = 0; q < 4; ++q) pre[k][q] = *reinterpret_cast<const float4*>(src + 4 * q);
        } else {
          float v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = (ci0 + j < d.Cin) ? src[j] : 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) pre[k][q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
      }
    }
    __syncthreads();  // every wave is done with the previous slice's halo and weight buffers
#pragma unroll
    for (int k = 0; k < NHI; ++k) {
      const int it = tid + k * 256;
      if (it >= npos_total * 2) break;
      if (hsrc[k] == -2) continue;
      const int pos = it >> 1, half = it & 1;
      float v[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[4 * q] = pre[k][q].x; v[4 * q + 1] = pre[k][q].y; v[4 * q + 2] = pre[k][q].z; v[4 * q + 3] = pre[k][q].w;
      }
      if ((a.in_scale || d.act) && hsrc[k] >= 0) {
        const int ci0 = chunk * 32 + half * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int ci = ci0 + j;
          if (ci < d.Cin) {
            float z = v[j];
            if (a.in_scale) z = z * a.in_scale[ci] + a.in_shift[ci];
            if (d.act) z = vca_act(z, d.act, d.act == VCA_ACT_PRELU ? a.in_alpha[ci] : d.slope);
            v[j] = z;
          }
        }
      }
      const int sw = sw16(pos);
      uint4 h0, l0, h1, l1;
      split8(v, h0, l0);
      split8(v + 8, h1, l1);
      unsigned char* ph = halo_hi + pos * 64;
      *reinterpret_cast<uint4*>(ph + (((2 * half) ^ sw) << 4)) = h0;
      *reinterpret_cast<uint4*>(ph + (((2 * half + 1) ^ sw) << 4)) = h1;
      if (NSPLIT == 3) {
        unsigned char* pl = halo_lo + pos * 64;
        *reinterpret_cast<uint4*>(pl + (((2 * half) ^ sw) << 4)) = l0;
        *reinterpret_cast<uint4*>(pl + (((2 * half + 1) ^ sw) << 4)) = l1;
      }
    }
  };

  const int cpz = (a.nchunk + a.ksplit - 1) / a.ksplit;
  const int chunk_beg = blockIdx.z * cpz, chunk_end = min(a.nchunk, chunk_beg + cpz);
  for (int chunk = chunk_beg; chunk < chunk_end; ++chunk) {
    stage_halo(chunk);
    issue_b(0, chunk, 0);
    wait_b();
    __syncthreads();

    int kh = 0, kw = 0;
    for (int tap = 0; tap < a.ntaps; ++tap) {
      const int buf = tap & 1;
      if (tap + 1 < a.ntaps) issue_b(tap + 1, chunk, buf ^ 1);
      const int tapoff = kh * a.HW + kw;
      if (++kw == d.KW) { kw = 0; ++kh; }
      const unsigned char* bt = bbase + buf * BBUF;
      bf16x8 bh[NT], bl[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(bt + boff[j]);
        if (NSPLIT == 3) bl[j] = *reinterpret_cast<const bf16x8*>(bt + BN * 64 + boff[j]);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int pa = hb[i] + tapoff;
        const int off = pa * 64 + ((lc ^ sw16(pa)) << 4);
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(halo_hi + off);
        bf16x8 al;
        if (NSPLIT == 3) al = *reinterpret_cast<const bf16x8*>(halo_lo + off);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (NSPLIT == 3) {
            acc[i][j] = VCA_MFMA_16x16x32(bh[j], al, acc[i][j]);
            acc[i][j] = VCA_MFMA_16x16x32(bl[j], ah, acc[i][j]);
          }
          acc[i][j] = VCA_MFMA_16x16x32(bh[j], ah, acc[i][j]);
        }
      }
      wait_b();
      __syncthreads();
    }
  }

  // ---- epilogue: D[row][col]: col = lane&15, row = 4*(lane>>4) + r
  // d2s_h > 0: the GEMM's Cout columns are 4 groups (ry, rx) of Cout/4 channels and column (ry, rx, c) of output
  // pixel (oy, ox) is element c of pixel (2*oy + ry, 2*ox + rx) of a [N][d2s_h][d2s_w][Cout/4] tensor.
  // The output pixel of each of the 128 GEMM rows is worked out once, by the first 128 threads, into the (now dead)
  // halo image: .x = index of the pixel (of pixel (2*oy, 2*ox) for d2s), -1 for a row without output; .y = d2s only:
  // bit 0 / bit 1 set when row 2*oy + 1 / column 2*ox + 1 exists.
  int2* row_tab = reinterpret_cast<int2*>(smem);
  float* stat_red = reinterpret_cast<float*>(smem + BM * sizeof(int2));   // [WM][BN][2], behind the row table
  if (tid < BM) {
    const int img = small_div(tid, inv_tpix), rem = tid - img * tpix;
    const int ty = small_div(rem, inv_tw), tx = rem - ty * a.TW;
    int nn, oy0, ox0;
    const bool sv = slot_origin(sid0 + img, a.nsub, a.tiles_x, a.tiles_y, a.TH, a.TW, nn, oy0, ox0);
    const int oy = oy0 + ty, ox = ox0 + tx;
    int2 e = make_int2(-1, 0);
    if (img < a.NI && sv && oy < d.OH && ox < d.OW) {
      if (d.d2s_h) {
        e.x = (nn * d.d2s_h + 2 * oy) * d.d2s_w + 2 * ox;
        e.y = (2 * oy + 1 < d.d2s_h ? 1 : 0) | (2 * ox + 1 < d.d2s_w ? 2 : 0);
      } else {
        e.x = (nn * d.OH + oy) * d.OW + ox;
      }
    }
    row_tab[tid] = e;
  }
  __syncthreads();
  // The MFMAs take the weight fragment as the row operand, so an accumulator tile is D[channel][pixel]: a lane holds
  // 4 consecutive channels (4 * (lane >> 4) + r) of ONE pixel (lane & 15) and stores / adds them as one float4.
  const int cq = d.Cout >> 2;
  const int cmul = d.d2s_h ? cq : d.Cout;   // channels per pixel of the output tensor
  const bool vec = (cmul & 3) == 0 && a.vecY;   // groups of 4 channels stay inside a pixel (and a sub-pixel group)
  size_t rbase[MT];
  int rflag[MT];   // -1: no output
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int2 e = row_tab[wm * WTM + i * 16 + lr];
    rbase[i] = (size_t)max(e.x, 0) * cmul;
    rflag[i] = e.x < 0 ? -1 : e.y;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nb = n0 + wn * WTN + j * 16 + 4 * lc;   // first of this lane's 4 channels
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
      const bool nv = nb < d.Cout;
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (nv && a.bias && blockIdx.z == 0) b = *reinterpret_cast<const float4*>(a.bias + nb);
      const int grp = d.d2s_h ? nb / cq : 0;
      const int nc = nb - grp * cq, ry = grp >> 1, rx = grp & 1;
      const int need = d.d2s_h ? (ry | (rx << 1)) : 0;                 // neighbours the channels' sub-pixel needs
      const int coff = d.d2s_h ? (ry * d.d2s_w + rx) * cq + nc : nb;   // offset of the channels from the row's base
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (nv && rflag[i] >= 0) {
          float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          if (a.stats) {
            s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
            s2[0] += v.x * v.x; s2[1] += v.y * v.y; s2[2] += v.z * v.z; s2[3] += v.w * v.w;
          }
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
          if ((rflag[i] & need) != need) continue;
          const size_t o = rbase[i] + coff;
          if (a.ksplit > 1) {
            if (a.part) {
              *reinterpret_cast<float4*>(a.part + (size_t)blockIdx.z * a.part_stride + o) = v;
            } else {
              atomicAdd(a.y + o, v.x); atomicAdd(a.y + o + 1, v.y);
              atomicAdd(a.y + o + 2, v.z); atomicAdd(a.y + o + 3, v.w);
            }
          } else {
            if (a.res) {
              const float4 rr = *reinterpret_cast<const float4*>(a.res + o);
              v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            v.x *= d.out_scale; v.y *= d.out_scale; v.z *= d.out_scale; v.w *= d.out_scale;
            *reinterpret_cast<float4*>(a.y + o) = v;
          }
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb + r;
        const bool nv = n < d.Cout;
        const float b = (nv && a.bias && blockIdx.z == 0) ? a.bias[n] : 0.f;
        const int grp = d.d2s_h ? n / cq : 0;
        const int nc = n - grp * cq, ry = grp >> 1, rx = grp & 1;
        const int need = d.d2s_h ? (ry | (rx << 1)) : 0;
        const int coff = d.d2s_h ? (ry * d.d2s_w + rx) * cq + nc : n;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (nv && rflag[i] >= 0) {
            float v = acc[i][j][r];
            s1[r] += v;
            s2[r] += v * v;
            v += b;
            if ((rflag[i] & need) != need) continue;
            const size_t o = rbase[i] + coff;
            if (a.ksplit > 1) {
              if (a.part) a.part[(size_t)blockIdx.z * a.part_stride + o] = v;
              else atomicAdd(a.y + o, v);
            } else {
              if (a.res) v += a.res[o];
              a.y[o] = v * d.out_scale;
            }
          }
        }
      }
    }
    if (a.stats) {   // totals over the 16 pixels (lanes lr) of the tile rows, per channel
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int sh = 1; sh < 16; sh <<= 1) {
          s1[r] += __shfl_xor(s1[r], sh, 64);
          s2[r] += __shfl_xor(s2[r], sh, 64);
        }
      }
      if (lr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          stat_red[(wm * BN + wn * WTN + j * 16 + 4 * lc + r) * 2] = s1[r];
          stat_red[(wm * BN + wn * WTN + j * 16 + 4 * lc + r) * 2 + 1] = s2[r];
        }
      }
    }
  }
  // Column statistics: the WM waves that share a column meet in LDS, then one pair of atomics per column and
  // workgroup into one of VCA_CONV_STAT_SLOTS partial buffers (atomics on one address are resolved one after the
  // other, ~10-20 ns each: thousands of workgroups adding to the 2*Cout addresses of a single buffer cost more than
  // the convolution on the narrow layers)
  if (a.stats) {
    __syncthreads();
    if (tid < BN && n0 + tid < d.Cout) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < WM; ++k) {
        s1 += stat_red[(k * BN + tid) * 2];
        s2 += stat_red[(k * BN + tid) * 2 + 1];
      }
      float* dst = a.stats + (size_t)(blockIdx.x % VCA_CONV_STAT_SLOTS) * 2 * d.Cout;
      atomicAdd(dst + n0 + tid, s1);
      atomicAdd(dst + d.Cout + n0 + tid, s2);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient (split-bf16)
//
// dW[tap][ci][co] = sum over output pixels of xin[pixel (+) tap][ci] * gy[pixel][co].  A workgroup owns a
// 32 x 32 (ci x co) tile of EVERY tap and walks patches of 128 output pixels (the reduction): per patch it stages
// the input halo [position][32 ci] and the output-gradient tile [pixel][32 co] in LDS as hi/lo bf16, and the MFMA
// operands (which need 8 consecutive reduction indices per lane) are fetched with the transposing LDS read
// ds_read_b64_tr_b16: 4 rows (pixels) x 16 columns (channels) per 16-lane group, delivered column-major.  The
// ntaps*8 (tap, 16-pixel slice) units of a patch are dealt to the 8 waves in equal contiguous runs, so every wave
// issues the same number of MFMAs whatever the filter size, and keeps the accumulators of the (at most 5) taps its
// run touches in registers for the whole walk.  Partial tiles are added to dW with fp32 atomics (pixels are also
// split across workgroups).
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WArgs {
  vca_conv_desc d;
  const float* x;
  const float* gy;
  const float* in_scale;
  const float* in_shift;
  const float* in_alpha;
  float* dw;  // fp32 [tap][Cin][Cout4]
  float* bias_grad;  // optional [Cout]: += column sums of gy (the bias gradient), from the tiles the kernel stages anyway
  int Cout4, ntaps;
  int NI, TH, TW, HH, HW, NPOS;
  int tiles_x, tiles_y, nsub, ntiles, tiles_per_block, co_tiles;
  int vecA, vecB;
  int slot_off;  // byte offset of the per-patch slot table behind the staged images in LDS
};

__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* p0, const unsigned char* p1) {
  // two transposed reads: reduction rows kbase..kbase+3 and kbase+4..kbase+7 of this lane's column
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v;
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
  v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  return *reinterpret_cast<bf16x8*>(&v);
}

template <int NSPLIT, bool RR>
__global__ __launch_bounds__(512, 2) void conv_bf16_wgrad_kernel(WArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MAXSLOT = 5;  // distinct taps a wave's run of units can touch (<= 32 taps)
  constexpr int RRSLOT = 4;   // taps per wave when whole taps are dealt round-robin (17..32 taps)
  const vca_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int npos_total = a.NI * a.NPOS;
  unsigned char* halo_hi = smem;
  unsigned char* halo_lo = halo_hi + npos_total * 64;
  unsigned char* gy_hi = halo_lo + (NSPLIT == 3 ? npos_total * 64 : 0);
  unsigned char* gy_lo = gy_hi + BM * 64;

  const int cot = blockIdx.x % a.co_tiles, cit = blockIdx.x / a.co_tiles;
  const int ci_base = cit * 32, co_base = cot * 32;
  const int He = d.Hin << d.ups, We = d.Win << d.ups;
  const int tpix = a.TH * a.TW;
  // this wave's run of (tap, slice) units: u = tap*8 + slice
  const int u_first = wave * a.ntaps, u_last = u_first + a.ntaps - 1;
  const int tap_first = u_first >> 3, tap_last = u_last >> 3;

  // transposed-read geometry of this lane: column (channel) group and reduction-row offsets
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int col = 16 * (g & 1) + 4 * p;       // first of the 4 channels this lane addresses
  const int kq = 8 * (g >> 1) + q;            // reduction row this lane addresses within a 16-deep MFMA step
  const int colchunk = col >> 3, colbyte = (col & 7) * 2;
  int pk[16];                                 // halo position of reduction row k = (s>>1)*16 + kq + 4*(s&1)
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int k = (s >> 1) * 16 + kq + 4 * (s & 1);
    const int img = k / tpix, rem = k - img * tpix;
    const int ty = rem / a.TW, tx = rem - ty * a.TW;
    pk[s] = img < a.NI ? img * a.NPOS + ty * d.stride * a.HW + tx * d.stride : 0;
  }

  f32x16 acc[MAXSLOT];
#pragma unroll
  for (int t = 0; t < MAXSLOT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // Staging plan, the same for every patch: which (slot, halo row, halo column) each of this thread's halo items
  // covers and which output pixel its output-gradient item covers.  Only the slots' image / origin change from
  // patch to patch; they are worked out once per patch by the first NI threads (slot_info).
  int (*slot_info)[4] = reinterpret_cast<int (*)[4]>(smem + a.slot_off);   // [BM]: n, oy0, ox0, valid
  constexpr int NHI = (2 * MAXPOS + 511) / 512;
  int h_img[NHI], h_y[NHI], h_x[NHI];
#pragma unroll
  for (int k = 0; k < NHI; ++k) {
    const int pos = (tid + k * 512) >> 1;
    h_img[k] = pos / a.NPOS;
    const int rp = pos - h_img[k] * a.NPOS;
    h_y[k] = rp / a.HW;
    h_x[k] = rp - h_y[k] * a.HW;
  }
  const int g_m = tid >> 2, g_qt = tid & 3;
  const int g_img = g_m / tpix, g_ty = (g_m - g_img * tpix) / a.TW, g_tx = (g_m - g_img * tpix) - g_ty * a.TW;

  // bias gradient: the workgroups of the first input-channel tile also sum the output-gradient values they stage
  const bool do_bias = a.bias_grad != nullptr && cit == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // whole-tap schedule: halo offset of the (at most RRSLOT) taps wave, wave + 8, ... this wave owns
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int rr_tapoff[RRSLOT];
#pragma unroll
  for (int ts = 0; ts < RRSLOT; ++ts) {
    const int tap = min(wave_u + 8 * ts, a.ntaps - 1);
    const int kh = tap / d.KW;
    rr_tapoff[ts] = kh * a.HW + tap - kh * d.KW;
  }

  const int pt_beg = blockIdx.y * a.tiles_per_block;
  const int pt_end = min(a.ntiles, pt_beg + a.tiles_per_block);
  for (int pt = pt_beg; pt < pt_end; ++pt) {
    const int sid0 = pt * a.NI;
    if (tid < a.NI) {
      int n, oy0, ox0;
      const bool sv = slot_origin(sid0 + tid, a.nsub, a.tiles_x, a.tiles_y, a.TH, a.TW, n, oy0, ox0);
      slot_info[tid][0] = n; slot_info[tid][1] = oy0; slot_info[tid][2] = ox0; slot_info[tid][3] = sv;
    }
    __syncthreads();   // slot_info visible; the previous patch's fragments have been read
    // ---- input halo of this patch, channels ci_base .. ci_base+31
#pragma unroll
    for (int k = 0; k < NHI; ++k) {
      const int it = tid + k * 512;
      if (it >= npos_total * 2) break;
      const int pos = it >> 1, half = it & 1;
      const int img = h_img[k];
      const int n = slot_info[img][0], oy0 = slot_info[img][1], ox0 = slot_info[img][2];
      const bool sv = slot_info[img][3] != 0;
      int vy = oy0 * d.stride + h_y[k] - d.pad_h, vx = ox0 * d.stride + h_x[k] - d.pad_w;
      const bool ok = sv && vy >= 0 && vx >= 0 && vy < He && vx < We;
      vy >>= d.ups;
      vx >>= d.ups;
      const int ci0 = ci_base + half * 16;
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = 0.f;
      if (ok && ci0 < d.Cin) {
        const float* src = a.x + ((size_t)(n * d.Hin + vy) * d.Win + vx) * d.Cin + ci0;
        if (a.vecA && ci0 + 16 <= d.Cin) {
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const float4 t = *reinterpret_cast<const float4*>(src + 4 * qq);
            v[4 * qq] = t.x; v[4 * qq + 1] = t.y; v[4 * qq + 2] = t.z; v[4 * qq + 3] = t.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (ci0 + j < d.Cin) v[j] = src[j];
        }
        if (a.in_scale || d.act) {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int ci = ci0 + j;
            if (ci < d.Cin) {
              float z = v[j];
              if (a.in_scale) z = z * a.in_scale[ci] + a.in_shift[ci];
              if (d.act) z = vca_act(z, d.act, d.act == VCA_ACT_PRELU ? a.in_alpha[ci] : d.slope);
              v[j] = z;
            }
          }
        }
      }
      const int sw = (pos >> 2) & 3;
      uint4 h0, l0, h1, l1;
      split8(v, h0, l0);
      split8(v + 8, h1, l1);
      *reinterpret_cast<uint4*>(halo_hi + pos * 64 + (((2 * half) ^ sw) << 4)) = h0;
      *reinterpret_cast<uint4*>(halo_hi + pos * 64 + (((2 * half + 1) ^ sw) << 4)) = h1;
      if (NSPLIT == 3) {
        *reinterpret_cast<uint4*>(halo_lo + pos * 64 + (((2 * half) ^ sw) << 4)) = l0;
        *reinterpret_cast<uint4*>(halo_lo + pos * 64 + (((2 * half + 1) ^ sw) << 4)) = l1;
      }
    }
    // ---- output-gradient tile [128 pixels][32 co]: 4 threads per pixel, 8 channels each
    {
      const int m = g_m, qt = g_qt;
      const int img = min(g_img, BM - 1);
      const int n = slot_info[img][0];
      const bool sv = slot_info[img][3] != 0;
      const int oy = slot_info[img][1] + g_ty, ox = slot_info[img][2] + g_tx;
      const int c0 = co_base + qt * 8;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
      if (g_img < a.NI && sv && oy < d.OH && ox < d.OW && c0 < d.Cout) {
        const float* src = a.gy + ((size_t)(n * d.OH + oy) * d.OW + ox) * d.Cout + c0;
        if (a.vecB && c0 + 8 <= d.Cout) {
          const float4 t0 = *reinterpret_cast<const float4*>(src), t1 = *reinterpret_cast<const float4*>(src + 4);
          v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (c0 + j < d.Cout) v[j] = src[j];
        }
      }
      if (do_bias) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += v[j];
      }
      const int sw = (m >> 2) & 3;
      uint4 h0, l0;
      split8(v, h0, l0);
      *reinterpret_cast<uint4*>(gy_hi + m * 64 + ((qt ^ sw) << 4)) = h0;
      if (NSPLIT == 3) *reinterpret_cast<uint4*>(gy_lo + m * 64 + ((qt ^ sw) << 4)) = l0;
    }
    __syncthreads();

    if constexpr (RR) {
    // ---- two halves of the patch (64 pixels each): the B fragments (output gradient) of a half stay in
    //      registers while every tap of this wave sweeps it
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      bf16x8 bh[4], bl[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int k0 = (hf * 4 + kk) * 16 + kq, k1 = k0 + 4;
        const int o0 = k0 * 64 + ((colchunk ^ ((k0 >> 2) & 3)) << 4) + colbyte;
        const int o1 = k1 * 64 + ((colchunk ^ ((k1 >> 2) & 3)) << 4) + colbyte;
        bh[kk] = tr_read8(gy_hi + o0, gy_hi + o1);
        if (NSPLIT == 3) bl[kk] = tr_read8(gy_lo + o0, gy_lo + o1);
      }
      if constexpr (NSPLIT == 1) {
        // the input fragments of the next tap are requested before the MFMAs of the current one, so the LDS latency
        // runs under them (a wave has the matrix pipe of its SIMD to itself for most of this phase)
        auto load_a = [&](int ts, bf16x8 (&af)[4]) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int ks = hf * 4 + kk;
            const int p0 = pk[2 * ks] + rr_tapoff[ts], p1 = pk[2 * ks + 1] + rr_tapoff[ts];
            const int o0 = p0 * 64 + ((colchunk ^ ((p0 >> 2) & 3)) << 4) + colbyte;
            const int o1 = p1 * 64 + ((colchunk ^ ((p1 >> 2) & 3)) << 4) + colbyte;
            af[kk] = tr_read8(halo_hi + o0, halo_hi + o1);
          }
        };
        bf16x8 af[2][4];
        if (wave_u < a.ntaps) load_a(0, af[0]);
#pragma unroll
        for (int ts = 0; ts < RRSLOT; ++ts) {
          const int tap = wave_u + 8 * ts;
          if (ts + 1 < RRSLOT && tap + 8 < a.ntaps) load_a(ts + 1, af[(ts + 1) & 1]);
          if (tap < a.ntaps) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
              acc[ts] = VCA_MFMA_32x32x16(af[ts & 1][kk], bh[kk], acc[ts]);
          }
        }
      } else {
#pragma unroll
      for (int ts = 0; ts < RRSLOT; ++ts) {
        const int tap = wave_u + 8 * ts;
        if (tap < a.ntaps) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int ks = hf * 4 + kk;
            const int p0 = pk[2 * ks] + rr_tapoff[ts], p1 = pk[2 * ks + 1] + rr_tapoff[ts];
            const int o0 = p0 * 64 + ((colchunk ^ ((p0 >> 2) & 3)) << 4) + colbyte;
            const int o1 = p1 * 64 + ((colchunk ^ ((p1 >> 2) & 3)) << 4) + colbyte;
            const bf16x8 ah = tr_read8(halo_hi + o0, halo_hi + o1);
            const bf16x8 al = tr_read8(halo_lo + o0, halo_lo + o1);
            acc[ts] = VCA_MFMA_32x32x16(al, bh[kk], acc[ts]);
            acc[ts] = VCA_MFMA_32x32x16(ah, bl[kk], acc[ts]);
            acc[ts] = VCA_MFMA_32x32x16(ah, bh[kk], acc[ts]);
          }
        }
      }
      }
    }
    } else {
    // ---- MFMA work units (tap, 16-pixel slice): ntaps*8 of them, dealt in contiguous runs of ntaps per wave
#pragma unroll
    for (int sl = 0; sl < MAXSLOT; ++sl) {
      const int tap = tap_first + sl;
      if (tap <= tap_last) {
        const int kh = tap / d.KW, kw = tap - kh * d.KW;
        const int tapoff = kh * a.HW + kw;
        const int ks_lo = max(u_first - tap * 8, 0), ks_hi = min(u_last - tap * 8, 7);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks < ks_lo || ks > ks_hi) continue;
          const int k0 = ks * 16 + kq, k1 = k0 + 4;
          const int g0 = k0 * 64 + ((colchunk ^ ((k0 >> 2) & 3)) << 4) + colbyte;
          const int g1 = k1 * 64 + ((colchunk ^ ((k1 >> 2) & 3)) << 4) + colbyte;
          const int p0 = pk[2 * ks] + tapoff, p1 = pk[2 * ks + 1] + tapoff;
          const int o0 = p0 * 64 + ((colchunk ^ ((p0 >> 2) & 3)) << 4) + colbyte;
          const int o1 = p1 * 64 + ((colchunk ^ ((p1 >> 2) & 3)) << 4) + colbyte;
          const bf16x8 bh = tr_read8(gy_hi + g0, gy_hi + g1);
          const bf16x8 ah = tr_read8(halo_hi + o0, halo_hi + o1);
          if (NSPLIT == 3) {
            const bf16x8 bl = tr_read8(gy_lo + g0, gy_lo + g1);
            const bf16x8 al = tr_read8(halo_lo + o0, halo_lo + o1);
            acc[sl] = VCA_MFMA_32x32x16(al, bh, acc[sl]);
            acc[sl] = VCA_MFMA_32x32x16(ah, bl, acc[sl]);
          }
          acc[sl] = VCA_MFMA_32x32x16(ah, bh, acc[sl]);
        }
      }
    }
    }
  }

  if (do_bias) {   // fold the per-thread sums (thread t holds channels 8*(t&3).. of the 32-channel tile) through LDS
    float* bsh = reinterpret_cast<float*>(smem);   // the staged images are not needed any more
    __syncthreads();
    if (tid < 32) bsh[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&bsh[g_qt * 8 + j], bsum[j]);
    __syncthreads();
    if (tid < 32 && co_base + tid < d.Cout) atomicAdd(a.bias_grad + co_base + tid, bsh[tid]);
  }

  if constexpr (RR) {
  // ---- accumulate into dW: D[row = ci][col = co]
  const int lr = lane & 31, lh = lane >> 5;
  const int co = co_base + lr;
#pragma unroll
  for (int ts = 0; ts < RRSLOT; ++ts) {
    const int tap = wave + 8 * ts;
    if (tap < a.ntaps && co < d.Cout) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci_base + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ci < d.Cin) atomicAdd(a.dw + ((size_t)tap * d.Cin + ci) * a.Cout4 + co, acc[ts][r]);
      }
    }
  }
  } else {
  // ---- accumulate into dW: D[row = ci][col = co]
  const int lr = lane & 31, lh = lane >> 5;
  const int co = co_base + lr;
#pragma unroll
  for (int sl = 0; sl < MAXSLOT; ++sl) {
    const int tap = tap_first + sl;
    if (tap <= tap_last && co < d.Cout) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci_base + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ci < d.Cin) atomicAdd(a.dw + ((size_t)tap * d.Cin + ci) * a.Cout4 + co, acc[sl][r]);
      }
    }
  }
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient for filters of at most 9 taps (3x3, 1x1, short 1-D): a workgroup of 4 waves owns a 64 x 64
// (ci x co) tile of every tap; wave (h_ci, h_co) owns one 32 x 32 quadrant and keeps the accumulators of all taps
// in registers.  Per patch of 128 output pixels the input halo [position][64 ci] and the output-gradient tile
// [pixel][64 co] are staged once (as two 32-channel images each, in the layout of the kernel above) and serve
// four times the MFMA work the 32 x 32 workgroup tile gets from the same staging; the output-gradient fragments
// of half a patch stay in registers while the taps sweep it.
// ------------------------------------------------------------------------------------------------
template <int NSPLIT>
__global__ __launch_bounds__(256, 2) void conv_bf16_wgrad_quad_kernel(WArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MAXT = 9;
  const vca_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cih = wave >> 1, coh = wave & 1;
  const int npos_total = a.NI * a.NPOS;
  unsigned char* halo_hi = smem;                                   // [2 ci halves][position][32 ci]
  unsigned char* halo_lo = halo_hi + 2 * npos_total * 64;
  unsigned char* gy_hi = halo_lo + (NSPLIT == 3 ? 2 * npos_total * 64 : 0);   // [2 co halves][pixel][32 co]
  unsigned char* gy_lo = gy_hi + 2 * BM * 64;

  const int cot = blockIdx.x % a.co_tiles, cit = blockIdx.x / a.co_tiles;
  const int ci_base = cit * 64, co_base = cot * 64;
  const int He = d.Hin << d.ups, We = d.Win << d.ups;
  const int tpix = a.TH * a.TW;

  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;   // transposed-read geometry of this lane
  // v_mfma_f32_16x16x32_bf16: the 32 x 32 quadrant of a tap is 2 x 2 tiles of 16 x 16 and a step reduces 32 pixels; lane l
  // feeds row / column l%16 with the 8 pixels 8*(l/16).. of the step (two transposed reads of 4 pixels each).  7 %
  // faster than the 32x32x16 form here (the chip holds a higher clock on it); the 8-wave kernel above lost 8 % with it.
  const int kq16 = 8 * g + q, c16sel = p >> 1, c16byte = (p & 1) * 8;
  int pk16[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int k = (s >> 1) * 32 + kq16 + 4 * (s & 1);
    const int img = k / tpix, rem = k - img * tpix;
    const int ty = rem / a.TW, tx = rem - ty * a.TW;
    pk16[s] = img < a.NI ? img * a.NPOS + ty * d.stride * a.HW + tx * d.stride : 0;
  }
  f32x4 acc16[MAXT][2][2];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc16[t][i >> 1][i & 1][r] = 0.f;

  // bias gradient: as in the kernel above (thread t stages channels 8*(t&7).. of the 64-channel tile for every pixel)
  const bool do_bias = a.bias_grad != nullptr && cit == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const unsigned char* my_hh = halo_hi + cih * npos_total * 64;
  const unsigned char* my_hl = halo_lo + cih * npos_total * 64;
  const unsigned char* my_gh = gy_hi + coh * BM * 64;
  const unsigned char* my_gl = gy_lo + coh * BM * 64;

  // Staging plan, the same for every patch (integer divisions are the expensive part of staging: done per patch and
  // item they cost more VALU time than the patch's MFMAs): slot, halo row and column of each of this thread's halo
  // items, slot and pixel of its output-gradient items, packed as slot | a << 8 | b << 20.  Only the slots' image and
  // origin change from patch to patch; the first NI threads work them out once per patch (slot_info).
  int (*slot_info)[4] = reinterpret_cast<int (*)[4]>(smem + a.slot_off);   // [BM]: n, oy0, ox0, valid
  // (Single product only: the split-product variant has no registers left for the plan - it spilled 15 and lost 1.5 %
  // of a step - and recomputes the coordinates.)
  constexpr int NHQ = 3;   // up to 192 positions x 4 quarter slices over 256 threads
  constexpr bool PLAN = NSPLIT == 1;
  int plan_h[PLAN ? NHQ : 1], plan_g[PLAN ? 4 : 1];
#pragma unroll
  for (int k = 0; k < (PLAN ? NHQ : 0); ++k) {
    const int pos = (tid + k * 256) >> 2;
    const int img = pos / a.NPOS, rp = pos - img * a.NPOS;
    const int hy = rp / a.HW, hx = rp - hy * a.HW;
    plan_h[k] = (tid + k * 256 < npos_total * 4) ? (img | (hy << 8) | (hx << 20)) : -1;
  }
#pragma unroll
  for (int rep = 0; rep < (PLAN ? 4 : 0); ++rep) {
    const int m = (tid + rep * 256) >> 3;
    const int img = m / tpix, rem = m - img * tpix;
    const int ty = rem / a.TW, tx = rem - ty * a.TW;
    plan_g[rep] = img < a.NI ? (img | (ty << 8) | (tx << 20)) : -1;
  }

  const int pt_beg = blockIdx.y * a.tiles_per_block;
  const int pt_end = min(a.ntiles, pt_beg + a.tiles_per_block);
  for (int pt = pt_beg; pt < pt_end; ++pt) {
    const int sid0 = pt * a.NI;
    if (tid < a.NI) {
      int n, oy0, ox0;
      const bool sv = slot_origin(sid0 + tid, a.nsub, a.tiles_x, a.tiles_y, a.TH, a.TW, n, oy0, ox0);
      slot_info[tid][0] = n; slot_info[tid][1] = oy0; slot_info[tid][2] = ox0; slot_info[tid][3] = sv;
    }
    __syncthreads();   // slot_info visible; the previous patch's fragments have been read
    // ---- input halo, channels ci_base .. ci_base+63: 4 threads per position, 16 channels each
#pragma unroll
    for (int k = 0; k < NHQ; ++k) {
      const int it = tid + k * 256;
      if (it >= npos_total * 4) break;
      asm volatile("" ::: "memory");   // one item at a time in registers (the accumulators of 9 taps leave room for no more)
      const int pos = it >> 2, qtr = it & 3;
      int img, hy, hx;
      if constexpr (PLAN) {
        img = plan_h[k] & 255; hy = (plan_h[k] >> 8) & 4095; hx = plan_h[k] >> 20;
      } else {
        img = pos / a.NPOS;
        const int rp = pos - img * a.NPOS;
        hy = rp / a.HW; hx = rp - hy * a.HW;
      }
      const int4 si = *reinterpret_cast<const int4*>(slot_info[img]);
      const int n = si.x, oy0 = si.y, ox0 = si.z;
      const bool sv = si.w != 0;
      int vy = oy0 * d.stride + hy - d.pad_h, vx = ox0 * d.stride + hx - d.pad_w;
      const bool ok = sv && vy >= 0 && vx >= 0 && vy < He && vx < We;
      vy >>= d.ups;
      vx >>= d.ups;
      const int ci0 = ci_base + qtr * 16;
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = 0.f;
      if (ok && ci0 < d.Cin) {
        const float* src = a.x + ((size_t)(n * d.Hin + vy) * d.Win + vx) * d.Cin + ci0;
        if (a.vecA && ci0 + 16 <= d.Cin) {
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const float4 t = *reinterpret_cast<const float4*>(src + 4 * qq);
            v[4 * qq] = t.x; v[4 * qq + 1] = t.y; v[4 * qq + 2] = t.z; v[4 * qq + 3] = t.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < 16; ++j)
            if (ci0 + j < d.Cin) v[j] = src[j];
        }
        if (a.in_scale || d.act) {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int ci = ci0 + j;
            if (ci < d.Cin) {
              float z = v[j];
              if (a.in_scale) z = z * a.in_scale[ci] + a.in_shift[ci];
              if (d.act) z = vca_act(z, d.act, d.act == VCA_ACT_PRELU ? a.in_alpha[ci] : d.slope);
              v[j] = z;
            }
          }
        }
      }
      const int sw = (pos >> 2) & 3, sub = qtr & 1;
      const int base = (qtr >> 1) * npos_total * 64 + pos * 64;
      uint4 h0, l0, h1, l1;
      split8(v, h0, l0);
      split8(v + 8, h1, l1);
      *reinterpret_cast<uint4*>(halo_hi + base + (((2 * sub) ^ sw) << 4)) = h0;
      *reinterpret_cast<uint4*>(halo_hi + base + (((2 * sub + 1) ^ sw) << 4)) = h1;
      if (NSPLIT == 3) {
        *reinterpret_cast<uint4*>(halo_lo + base + (((2 * sub) ^ sw) << 4)) = l0;
        *reinterpret_cast<uint4*>(halo_lo + base + (((2 * sub + 1) ^ sw) << 4)) = l1;
      }
    }
    // ---- output-gradient tile [128 pixels][64 co]: 8 threads per pixel, 8 channels each
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      const int it = tid + rep * 256;
      const int m = it >> 3, oc = it & 7;
      int pg;
      if constexpr (PLAN) {
        pg = plan_g[rep];
      } else {
        const int img = m / tpix, rem = m - img * tpix;
        const int ty = rem / a.TW, tx = rem - ty * a.TW;
        pg = img < a.NI ? (img | (ty << 8) | (tx << 20)) : -1;
      }
      const int4 si = *reinterpret_cast<const int4*>(slot_info[pg < 0 ? 0 : (pg & 255)]);
      const int n = si.x;
      const bool sv = si.w != 0;
      const int oy = si.y + ((pg >> 8) & 4095), ox = si.z + (pg >> 20);
      const int c0 = co_base + oc * 8;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
      if (pg >= 0 && sv && oy < d.OH && ox < d.OW && c0 < d.Cout) {
        const float* src = a.gy + ((size_t)(n * d.OH + oy) * d.OW + ox) * d.Cout + c0;
        if (a.vecB && c0 + 8 <= d.Cout) {
          const float4 t0 = *reinterpret_cast<const float4*>(src), t1 = *reinterpret_cast<const float4*>(src + 4);
          v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (c0 + j < d.Cout) v[j] = src[j];
        }
      }
      if (do_bias) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += v[j];
      }
      const int sw = (m >> 2) & 3, qt = oc & 3;
      const int base = (oc >> 2) * BM * 64 + m * 64 + ((qt ^ sw) << 4);
      uint4 h0, l0;
      split8(v, h0, l0);
      *reinterpret_cast<uint4*>(gy_hi + base) = h0;
      if (NSPLIT == 3) *reinterpret_cast<uint4*>(gy_lo + base) = l0;
    }
    __syncthreads();

    {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      bf16x8 bh[2][2], bl[2][2];              // [32-pixel step][co half of the quadrant]
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const int k0 = (hf * 2 + st) * 32 + kq16, k1 = k0 + 4;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          const int o0 = k0 * 64 + (((2 * ch + c16sel) ^ ((k0 >> 2) & 3)) << 4) + c16byte;
          const int o1 = k1 * 64 + (((2 * ch + c16sel) ^ ((k1 >> 2) & 3)) << 4) + c16byte;
          bh[st][ch] = tr_read8(my_gh + o0, my_gh + o1);
          if (NSPLIT == 3) bl[st][ch] = tr_read8(my_gl + o0, my_gl + o1);
        }
      }
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        if (t < a.ntaps) {
          const int kh = t / d.KW, kw = t - kh * d.KW;
          int tapoff = kh * a.HW + kw;
          asm volatile("" : "+s"(tapoff));
#pragma unroll
          for (int st = 0; st < 2; ++st) {
            const int s32 = hf * 2 + st;
            const int p0 = pk16[2 * s32] + tapoff, p1 = pk16[2 * s32 + 1] + tapoff;
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) {
              const int o0 = p0 * 64 + (((2 * ih + c16sel) ^ ((p0 >> 2) & 3)) << 4) + c16byte;
              const int o1 = p1 * 64 + (((2 * ih + c16sel) ^ ((p1 >> 2) & 3)) << 4) + c16byte;
              const bf16x8 ah = tr_read8(my_hh + o0, my_hh + o1);
              bf16x8 al;
              if (NSPLIT == 3) al = tr_read8(my_hl + o0, my_hl + o1);
#pragma unroll
              for (int ch = 0; ch < 2; ++ch) {
                if (NSPLIT == 3) {
                  acc16[t][ih][ch] = VCA_MFMA_16x16x32(al, bh[st][ch], acc16[t][ih][ch]);
                  acc16[t][ih][ch] = VCA_MFMA_16x16x32(ah, bl[st][ch], acc16[t][ih][ch]);
                }
                acc16[t][ih][ch] = VCA_MFMA_16x16x32(ah, bh[st][ch], acc16[t][ih][ch]);
              }
            }
          }
        }
      }
    }
    }
  }

  if (do_bias) {
    float* bsh = reinterpret_cast<float*>(smem);   // the staged images are not needed any more
    __syncthreads();
    if (tid < 64) bsh[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&bsh[(tid & 7) * 8 + j], bsum[j]);
    __syncthreads();
    if (tid < 64 && co_base + tid < d.Cout) atomicAdd(a.bias_grad + co_base + tid, bsh[tid]);
  }

  {
  // ---- accumulate into dW: 16 x 16 tiles, col = lane&15, row = 4*(lane>>4) + r
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t < a.ntaps) {
#pragma unroll
      for (int ih = 0; ih < 2; ++ih)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          const int co = co_base + coh * 32 + ch * 16 + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ci = ci_base + cih * 32 + ih * 16 + 4 * g + r;
            if (ci < d.Cin && co < d.Cout)
              atomicAdd(a.dw + ((size_t)t * d.Cin + ci) * a.Cout4 + co, acc16[t][ih][ch][r]);
          }
        }
    }
  }
  }
}

// ------------------------------------------------------------------------------------------------
// 8-wave form of the quadrant kernel (single product).  On the large maps a launch of the 4-wave kernel is the sum
// of its phases: the global loads of a patch run at the HBM's streaming rate for as long as they last (44 % of the
// launch), then conversion, then the MFMAs, and its 144 accumulator registers leave no room to fetch a patch ahead.
// Here two groups of four waves share the quadrants and split the taps (at most 5 each: 80 accumulator registers),
// 512 threads halve the staging items per thread, and the raw fp32 values of the NEXT patch are fetched into
// registers right before the MFMA phase of the current one, so the loads run under it; half as many workgroups also
// issue half as many final atomics.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void conv_bf16_wgrad_quad8_kernel(WArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MAXT8 = 5;
  const vca_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int quadrant = wave & 3, tg = __builtin_amdgcn_readfirstlane(wave >> 2);   // tg: wave-uniform, kept scalar
  const int cih = quadrant >> 1, coh = quadrant & 1;
  const int npos_total = a.NI * a.NPOS;
  unsigned char* halo_hi = smem;                                   // [2 ci halves][position][32 ci]
  unsigned char* gy_hi = halo_hi + 2 * npos_total * 64;            // [2 co halves][pixel][32 co]
  int (*slot_info)[4] = reinterpret_cast<int (*)[4]>(smem + a.slot_off);   // two tables of [BM]: n, oy0, ox0, valid

  const int cot = blockIdx.x % a.co_tiles, cit = blockIdx.x / a.co_tiles;
  const int ci_base = cit * 64, co_base = cot * 64;
  const int He = d.Hin << d.ups, We = d.Win << d.ups;
  const int tpix = a.TH * a.TW;
  const int t_half = (a.ntaps + 1) >> 1;
  const int t_beg = tg ? t_half : 0, t_end = tg ? a.ntaps : t_half;

  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;   // transposed-read geometry of this lane
  const int kq16 = 8 * g + q, c16sel = p >> 1, c16byte = (p & 1) * 8;
  int pk16[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int k = (s >> 1) * 32 + kq16 + 4 * (s & 1);
    const int img = k / tpix, rem = k - img * tpix;
    const int ty = rem / a.TW, tx = rem - ty * a.TW;
    pk16[s] = img < a.NI ? img * a.NPOS + ty * d.stride * a.HW + tx * d.stride : 0;
  }
  f32x4 acc16[MAXT8][2][2];
#pragma unroll
  for (int t = 0; t < MAXT8; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc16[t][i >> 1][i & 1][r] = 0.f;

  const bool do_bias = a.bias_grad != nullptr && cit == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const unsigned char* my_hh = halo_hi + cih * npos_total * 64;
  const unsigned char* my_gh = gy_hi + coh * BM * 64;

  // staging plan (see the 4-wave kernel): 2 halo items (position, 16-channel quarter) and 2 output-gradient items
  // (pixel, 8-channel eighth) per thread, packed as slot | a << 8 | b << 20
  int plan_h[2], plan_g[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int pos = (tid + k * 512) >> 2;
    const int img = pos / a.NPOS, rp = pos - img * a.NPOS;
    const int hy = rp / a.HW, hx = rp - hy * a.HW;
    plan_h[k] = (tid + k * 512 < npos_total * 4) ? (img | (hy << 8) | (hx << 20)) : -1;
    const int m = (tid + k * 512) >> 3;
    const int gi = m / tpix, rem = m - gi * tpix;
    const int ty = rem / a.TW, tx = rem - ty * a.TW;
    plan_g[k] = gi < a.NI ? (gi | (ty << 8) | (tx << 20)) : -1;
  }

  float4 ph[2][4], pg[2][2];
  auto write_slots = [&](int pt, int tbl) {
    if (tid < a.NI) {
      int n, oy0, ox0;
      const bool sv = slot_origin(pt * a.NI + tid, a.nsub, a.tiles_x, a.tiles_y, a.TH, a.TW, n, oy0, ox0);
      int* si = slot_info[tbl * BM + tid];
      si[0] = n; si[1] = oy0; si[2] = ox0; si[3] = sv;
    }
  };
  auto fetch = [&](int tbl) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) ph[k][qq] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (plan_h[k] >= 0) {
        const int qtr = (tid + k * 512) & 3;
        const int img = plan_h[k] & 255, hy = (plan_h[k] >> 8) & 4095, hx = plan_h[k] >> 20;
        const int4 si = *reinterpret_cast<const int4*>(slot_info[tbl * BM + img]);
        int vy = si.y * d.stride + hy - d.pad_h, vx = si.z * d.stride + hx - d.pad_w;
        const bool ok = si.w != 0 && vy >= 0 && vx >= 0 && vy < He && vx < We;
        vy >>= d.ups;
        vx >>= d.ups;
        const int ci0 = ci_base + qtr * 16;
        if (ok && ci0 < d.Cin) {
          const float* src = a.x + ((size_t)(si.x * d.Hin + vy) * d.Win + vx) * d.Cin + ci0;
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) ph[k][qq] = *reinterpret_cast<const float4*>(src + 4 * qq);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      pg[k][0] = make_float4(0.f, 0.f, 0.f, 0.f);
      pg[k][1] = pg[k][0];
      const int oc = (tid + k * 512) & 7;
      const int pl = plan_g[k];
      if (pl >= 0) {
        const int4 si = *reinterpret_cast<const int4*>(slot_info[tbl * BM + (pl & 255)]);
        const int oy = si.y + ((pl >> 8) & 4095), ox = si.z + (pl >> 20);
        const int c0 = co_base + oc * 8;
        if (si.w != 0 && oy < d.OH && ox < d.OW && c0 < d.Cout) {
          const float* src = a.gy + ((size_t)(si.x * d.OH + oy) * d.OW + ox) * d.Cout + c0;
          pg[k][0] = *reinterpret_cast<const float4*>(src);
          pg[k][1] = *reinterpret_cast<const float4*>(src + 4);
        }
      }
    }
  };

  const int pt_beg = blockIdx.y * a.tiles_per_block;
  const int pt_end = min(a.ntiles, pt_beg + a.tiles_per_block);
  if (pt_beg < pt_end) {
    write_slots(pt_beg, 0);
    __syncthreads();
    fetch(0);
  }
  for (int pt = pt_beg; pt < pt_end; ++pt) {
    const int cur = (pt - pt_beg) & 1;
    if (pt + 1 < pt_end) write_slots(pt + 1, cur ^ 1);
    __syncthreads();   // the previous patch's fragments have been read; the next patch's slot table is visible
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (plan_h[k] < 0) break;
      const int it = tid + k * 512;
      const int pos = it >> 2, qtr = it & 3;
      float v[16];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        v[4 * qq] = ph[k][qq].x; v[4 * qq + 1] = ph[k][qq].y; v[4 * qq + 2] = ph[k][qq].z; v[4 * qq + 3] = ph[k][qq].w;
      }
      const int sw = (pos >> 2) & 3, sub = qtr & 1;
      const int base = (qtr >> 1) * npos_total * 64 + pos * 64;
      uint4 h0, l0, h1, l1;
      split8(v, h0, l0);
      split8(v + 8, h1, l1);
      *reinterpret_cast<uint4*>(halo_hi + base + (((2 * sub) ^ sw) << 4)) = h0;
      *reinterpret_cast<uint4*>(halo_hi + base + (((2 * sub + 1) ^ sw) << 4)) = h1;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int it = tid + k * 512;
      const int m = it >> 3, oc = it & 7;
      const float v[8] = {pg[k][0].x, pg[k][0].y, pg[k][0].z, pg[k][0].w, pg[k][1].x, pg[k][1].y, pg[k][1].z, pg[k][1].w};
      if (do_bias) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += v[j];
      }
      const int sw = (m >> 2) & 3, qt = oc & 3;
      const int base = (oc >> 2) * BM * 64 + m * 64 + ((qt ^ sw) << 4);
      uint4 h0, l0;
      split8(v, h0, l0);
      *reinterpret_cast<uint4*>(gy_hi + base) = h0;
    }
    __syncthreads();
    if (pt + 1 < pt_end) fetch(cur ^ 1);   // in flight under the MFMAs below

#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      bf16x8 bh[2][2];              // [32-pixel step][co half of the quadrant]
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const int k0 = (hf * 2 + st) * 32 + kq16, k1 = k0 + 4;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          const int o0 = k0 * 64 + (((2 * ch + c16sel) ^ ((k0 >> 2) & 3)) << 4) + c16byte;
          const int o1 = k1 * 64 + (((2 * ch + c16sel) ^ ((k1 >> 2) & 3)) << 4) + c16byte;
          bh[st][ch] = tr_read8(my_gh + o0, my_gh + o1);
        }
      }
#pragma unroll
      for (int tt = 0; tt < MAXT8; ++tt) {
        const int t = t_beg + tt;
        if (t < t_end) {
          const int kh = t / d.KW, kw = t - kh * d.KW;
          int tapoff = kh * a.HW + kw;
          asm volatile("" : "+s"(tapoff));
#pragma unroll
          for (int st = 0; st < 2; ++st) {
            const int s32 = hf * 2 + st;
            const int p0 = pk16[2 * s32] + tapoff, p1 = pk16[2 * s32 + 1] + tapoff;
#pragma unroll
            for (int ih = 0; ih < 2; ++ih) {
              const int o0 = p0 * 64 + (((2 * ih + c16sel) ^ ((p0 >> 2) & 3)) << 4) + c16byte;
              const int o1 = p1 * 64 + (((2 * ih + c16sel) ^ ((p1 >> 2) & 3)) << 4) + c16byte;
              const bf16x8 ah = tr_read8(my_hh + o0, my_hh + o1);
#pragma unroll
              for (int ch = 0; ch < 2; ++ch)
                acc16[tt][ih][ch] = VCA_MFMA_16x16x32(ah, bh[st][ch], acc16[tt][ih][ch]);
            }
          }
        }
      }
    }
  }

  if (do_bias) {
    float* bsh = reinterpret_cast<float*>(smem);   // the staged images are not needed any more
    __syncthreads();
    if (tid < 64) bsh[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&bsh[(tid & 7) * 8 + j], bsum[j]);
    __syncthreads();
    if (tid < 64 && co_base + tid < d.Cout) atomicAdd(a.bias_grad + co_base + tid, bsh[tid]);
  }

  // ---- accumulate into dW: 16 x 16 tiles, col = lane&15, row = 4*(lane>>4) + r
#pragma unroll
  for (int tt = 0; tt < MAXT8; ++tt) {
    const int t = t_beg + tt;
    if (t < t_end) {
#pragma unroll
      for (int ih = 0; ih < 2; ++ih)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          const int co = co_base + coh * 32 + ch * 16 + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ci = ci_base + cih * 32 + ih * 16 + 4 * g + r;
            if (ci < d.Cin && co < d.Cout)
              atomicAdd(a.dw + ((size_t)t * d.Cin + ci) * a.Cout4 + co, acc16[tt][ih][ch][r]);
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight packing: fp32 [Cout][Cin][taps] -> bf16 [tap][chunk][npad][hi 32 | lo 32]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float s2_weight(const float* __restrict__ src, int Cin, int tap, int n, int co);

__global__ void pack_weight_bf16_kernel(int mode, int Cout, int Cin, int taps, int N, int K, int npad, int nchunk,
                                        const float* __restrict__ src, vca_e16* __restrict__ dst) {
  const size_t total = (size_t)taps * nchunk * npad * 32;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = i % 32;
  size_t r = i / 32;
  const int n = r % npad; r /= npad;
  const int chunk = r % nchunk;
  const int tap = r / nchunk;
  const int k = chunk * 32 + j;
  float v = 0.f;
  if (n < N && k < K) {
    if (mode == 0) v = src[((size_t)n * Cin + k) * taps + tap];                    // N = Cout, K = Cin
    else if (mode == 1) v = src[((size_t)k * Cin + n) * taps + (taps - 1 - tap)];  // N = Cin,  K = Cout
    else v = s2_weight(src, Cin, tap, n, k);
  }
  const vca_e16 h = (vca_e16)v;
  const vca_e16 l = (vca_e16)(v - (float)h);
  const size_t row = ((size_t)(tap * nchunk + chunk) * npad + n) * 64;
  dst[row + j] = h;
  dst[row + 32 + j] = l;
}

// Mode 2 (input gradient of a 3x3, stride-2, padding-1 convolution in sub-pixel form): the gradient of input pixel
// (2a + ry, 2b + rx) only involves the filter taps of matching parity, gx[2a+r] = sum over taps k = (r+1)%2 + 2m of
// w[k] * gy[a + (r+1)/2 - m], so the four parities are four convolutions of the UNdilated output gradient over a
// 2 x 2 window (offsets dy, dx in {0, 1}), run as one convolution with 4*Cin output columns (ry, rx, ci) whose
// epilogue interleaves them: 16 multiply-adds per output-gradient pixel instead of the 36 of the zero-dilated form.
// dst taps: (dy, dx); N = 4*Cin; K = Cout.
__device__ __forceinline__ float s2_weight(const float* __restrict__ src, int Cin, int tap, int n, int co) {
  const int grp = n / Cin, ci = n - grp * Cin;
  const int ry = grp >> 1, rx = grp & 1, dy = tap >> 1, dx = tap & 1;
  // r = 0: offset 0 <- tap 1;  r = 1: offset 1 <- tap 0, offset 0 <- tap 2
  const int ky = ry ? (dy ? 0 : 2) : (dy ? -1 : 1);
  const int kx = rx ? (dx ? 0 : 2) : (dx ? -1 : 1);
  if (ky < 0 || kx < 0) return 0.f;
  return src[((size_t)co * Cin + ci) * 9 + ky * 3 + kx];
}

// Same packing for filters of at most 32 taps with coalesced traffic on both sides: a workgroup stages the
// 32 k x 8 n x taps source values of one (channel slice, 8 output rows) group in LDS (the source is contiguous
// over (k, tap) for a fixed n in mode 0 and over (n, tap) for a fixed k in mode 1) and writes whole 128-byte rows.
__device__ __forceinline__ void pack_tile_bf16(int mode, int Cin, int taps, int N, int K, int npad, int nchunk,
                                               const float* __restrict__ src, vca_e16* __restrict__ dst, int chunk,
                                               int n0, float* tl /* [8 n][32 k][taps] */) {
  const int k0 = chunk * 32;
  const int tid = threadIdx.x;
  const int per = 32 * taps;
  if (mode == 2) {   // 4 destination taps, gathered straight from the source (small filters, a few KB per tile)
    for (int i = tid; i < 4 * 32; i += 256) {
      const int jq = i & 3, nl = (i >> 2) & 7, t = i >> 5;
      const int n = n0 + nl;
      if (n < npad) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + jq * 8 + u;
          v[u] = (n < N && k < K) ? s2_weight(src, Cin, t, n, k) : 0.f;
        }
        uint4 h, l;
        split8(v, h, l);
        unsigned char* row = reinterpret_cast<unsigned char*>(dst) + ((size_t)(t * nchunk + chunk) * npad + n) * 128;
        *reinterpret_cast<uint4*>(row + jq * 16) = h;
        *reinterpret_cast<uint4*>(row + 64 + jq * 16) = l;
      }
    }
    return;
  }
  if (mode == 0) {
    const int rmax = min(K - k0, 32) * taps;          // entries of this slice that exist in the source
    for (int r = tid; r < per; r += 256) {
      float v[8];
#pragma unroll
      for (int nl = 0; nl < 8; ++nl)
        v[nl] = (n0 + nl < N && r < rmax) ? src[((size_t)(n0 + nl) * Cin + k0) * taps + r] : 0.f;
#pragma unroll
      for (int nl = 0; nl < 8; ++nl) tl[nl * per + r] = v[nl];
    }
  } else {
    const int nl = tid / taps, tt = tid - nl * taps;  // 8*taps <= 256: one source element per thread and k
    const int n = n0 + nl;
    if (nl < 8) {
      float v[32];
#pragma unroll
      for (int kl = 0; kl < 32; ++kl) {     // all 32 loads in flight before the first LDS write
        const int k = k0 + kl;
        v[kl] = (n < N && k < K) ? src[((size_t)k * Cin + n) * taps + tt] : 0.f;
      }
#pragma unroll
      for (int kl = 0; kl < 32; ++kl) tl[(nl * 32 + kl) * taps + (taps - 1 - tt)] = v[kl];
    }
  }
  __syncthreads();
  for (int i = tid; i < taps * 32; i += 256) {       // (tap, n, quarter row): 8 k -> one 16-byte hi and lo store
    const int jq = i & 3, nl = (i >> 2) & 7, t = i >> 5;
    const int n = n0 + nl;
    if (n < npad) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = tl[(nl * 32 + jq * 8 + u) * taps + t];
      uint4 h, l;
      split8(v, h, l);
      unsigned char* row = reinterpret_cast<unsigned char*>(dst) + ((size_t)(t * nchunk + chunk) * npad + n) * 128;
      *reinterpret_cast<uint4*>(row + jq * 16) = h;
      *reinterpret_cast<uint4*>(row + 64 + jq * 16) = l;
    }
  }
}

__global__ __launch_bounds__(256) void pack_weight_bf16_tiled_kernel(int mode, int Cin, int taps, int N, int K,
                                                                     int npad, int nchunk,
                                                                     const float* __restrict__ src,
                                                                     vca_e16* __restrict__ dst) {
  extern __shared__ float tl[];   // [8 n][32 k][taps]
  pack_tile_bf16(mode, Cin, taps, N, K, npad, nchunk, src, dst, blockIdx.x, blockIdx.y * 8, tl);
}

// Every packed copy of a network's weights in one launch (after its optimizer step): the descriptors list the
// (parameter, role) pairs with the first tile of each in a running count; a workgroup finds its descriptor by
// bisection and packs one (channel slice, 8 rows) tile exactly as the single-weight kernel does.
__global__ __launch_bounds__(256) void pack_weight_bf16_batch_kernel(const vca_pack_desc* __restrict__ table,
                                                                     int ndesc) {
  extern __shared__ float tl[];
  const int tile = blockIdx.x;
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].tile_begin <= tile) lo = mid;
    else hi = mid - 1;
  }
  const vca_pack_desc d = table[lo];
  const int local = tile - d.tile_begin;
  const int K = d.mode == 0 ? d.Cin : d.Cout, N = d.mode == 0 ? d.Cout : (d.mode == 1 ? d.Cin : 4 * d.Cin);
  const int npad = (N + 31) / 32 * 32, nchunk = (K + 31) / 32;
  if (local >= nchunk * (npad / 8)) return;
  pack_tile_bf16(d.mode, d.Cin, d.taps, N, K, npad, nchunk, d.src, reinterpret_cast<vca_e16*>(d.dst), local % nchunk,
                 (local / nchunk) * 8, tl);
}

// y[i] = sum over the S slabs of part[s * stride + i]: folds the partial sums of a split-reduction launch.
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ part, unsigned stride, int S,
                                                        float* __restrict__ y, unsigned n) {
  const unsigned i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    float4 acc = *reinterpret_cast<const float4*>(part + i);
    for (int s = 1; s < S; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(part + (size_t)s * stride + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if ((reinterpret_cast<size_t>(y) & 15) == 0) {
      *reinterpret_cast<float4*>(y + i) = acc;
    } else {
      y[i] = acc.x; y[i + 1] = acc.y; y[i + 2] = acc.z; y[i + 3] = acc.w;
    }
  } else {
    for (unsigned k = i; k < n; ++k) {
      float acc = part[k];
      for (int s = 1; s < S; ++s) acc += part[(size_t)s * stride + k];
      y[k] = acc;
    }
  }
}

struct TileChoice {
  int NI, TH, TW;
};

// Pitches of the halo image for the 16x16x32 kernel.  Its fragment reads are conflict-free when the 16 rows of a
// fragment are 16 consecutive halo positions; a fragment that wraps from one patch row to the next (or to the next
// slot) jumps by pitch - TW (resp. a slot-dependent amount) positions, and the pattern survives exactly when that
// jump is a multiple of 8 positions.  With `padded` the row pitch is TW + 8k and the slot pitch matches; the padding
// is never staged and never read.  Measured per layer: -3...-5 % on the 5x5 layers with >= 128 output channels (the
// split-product launches that sit at the LDS / MFMA limit); a loss wherever the padding shrinks the patch or the
// number of images per workgroup (small maps: up to 2x), on 3x3 filters (6 padded positions per row for 2 taps of
// reach) and on the narrow-channel kernels, so only the first group gets it.
inline void halo_pitch(int TH, int TW, int KH, int KW, int stride, bool padded, int& hw, int& npos) {
  const int hh = (TH - 1) * stride + KH;
  hw = (TW - 1) * stride + KW;
  npos = hh * hw;
  if (!padded || stride != 1 || KW == 1) return;
  hw = TW + (KW - 1 + 7) / 8 * 8;
  npos = hh * hw;
  // first position of the next slot minus (last position of this slot's tile + 1) must be 0 mod 8
  const int want = ((TH - 1) * hw + TW) & 7;
  npos += (want - npos) & 7;
}

// Patch shape maximising the fraction of the 128 GEMM rows that are real output pixels, subject to the halo
// fitting in LDS.  Cached per geometry.
TileChoice choose_tile(int N, int OH, int OW, int KH, int KW, int stride, int Cout, int maxpos = MAXPOS,
                       double stage_cost = 6.0, bool padded = false) {
  static std::mutex mu;
  static std::unordered_map<unsigned long long, TileChoice> cache;
  const unsigned long long key = ((unsigned long long)N << 44) ^ ((unsigned long long)OH << 32) ^
                                 ((unsigned long long)OW << 20) ^ (KH << 12) ^ (KW << 4) ^ stride ^
                                 ((unsigned long long)(Cout > 96 ? 3 : (Cout > 32 ? 2 : 1)) << 58) ^
                                 ((unsigned long long)maxpos << 8) ^ ((unsigned long long)(stage_cost * 4) << 24) ^
                                 ((unsigned long long)padded << 63);
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
  }
  // cost model per workgroup and 32-channel slice: MFMA cycles of all taps + cycles to stage the halo positions
  TileChoice best{1, 1, 1};
  double best_cost = 1e300;
  const int bn32 = Cout > 96 ? 4 : (Cout > 32 ? 2 : 1);
  const double mfma = (double)KH * KW * 192.0 * bn32;
  for (int th = 1; th <= OH && th <= BM; ++th) {
    for (int tw = 1; tw <= OW && th * tw <= BM; ++tw) {
      const int hh = (th - 1) * stride + KH, hw = (tw - 1) * stride + KW;
      int slot = hh * hw, pitch = hw;
      halo_pitch(th, tw, KH, KW, stride, padded, pitch, slot);
      if (slot > maxpos) continue;
      const long long nsub = (long long)N * ((OH + th - 1) / th) * ((OW + tw - 1) / tw);
      int ni = BM / (th * tw);
      if (ni > nsub) ni = (int)nsub;
      while (ni > 1 && ni * slot > maxpos) --ni;
      const long long groups = (nsub + ni - 1) / ni;
      double cost = (double)groups * (mfma + stage_cost * ni * hh * hw);
      cost *= 1.0 + 0.02 / tw;   // tie-break: wider patches (longer contiguous runs, fewer LDS bank conflicts)
      if (cost < best_cost) {
        best_cost = cost;
        best = TileChoice{ni, th, tw};
      }
    }
  }
  std::lock_guard<std::mutex> g(mu);
  cache[key] = best;
  return best;
}

// Number of channel-slice ranges a launch of `blocks` output tiles is split into (1 = no split): few output tiles
// but a deep reduction (the discriminators' 5x5 layers on 5x18 and 10x37 maps) would leave most of the chip idle.
inline int choose_ksplit(long long blocks, int nchunk) {
  if (blocks >= 224 || nchunk < 4) return 1;
  int ks = (int)((448 + blocks - 1) / blocks);
  if (ks > nchunk) ks = nchunk;               // down to one 32-channel slice per workgroup
  const int cpz = (nchunk + ks - 1) / ks;
  return (nchunk + cpz - 1) / cpz;            // no empty ranges
}

inline size_t out_floats(const vca_conv_desc& d) {
  return d.d2s_h ? (size_t)d.N * d.d2s_h * d.d2s_w * (d.Cout / 4) : (size_t)d.N * d.OH * d.OW * d.Cout;
}

template <int BN, int WM, int WN, int NSPLIT>
int launch_halo(HArgs a, int gx, int gy, size_t halo, float* ws, size_t ws_floats, hipStream_t s) {
  auto kern = conv_bf16_halo_kernel<BN, WM, WN, NSPLIT>;
  const size_t lds = halo + 2 * (size_t)BN * (NSPLIT == 3 ? 128 : 64);
  VCA_MAX_LDS(kern);
  // split reduction: with a workspace every range stores its partial sums in a slab of its own and a second kernel
  // folds the slabs into y; without one the partial sums meet in a zero-filled y through fp32 atomics (device-scope
  // atomics are resolved beyond the XCDs' L2s: 15-35 us per launch against 3-5 us for the fold)
  a.ksplit = 1;
  a.part = nullptr;
  a.part_stride = 0;
  const size_t total = out_floats(a.d);
  if (!a.res && !a.stats && a.d.out_scale == 1.0f) a.ksplit = choose_ksplit((long long)gx * gy, a.nchunk);
  if (a.ksplit > 1) {
    const size_t stride = (total + 63) / 64 * 64;
    if (ws && ws_floats >= stride * a.ksplit && stride < (1ull << 32)) {
      a.part = ws;
      a.part_stride = (unsigned)stride;
    } else {
      vca_zero_async(a.y, total, s);
    }
  }
  hipLaunchKernelGGL(kern, dim3(gx, gy, a.ksplit), dim3(256), lds, s, a);
  if (a.part)
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, s, a.part,
                       a.part_stride, a.ksplit, a.y, (unsigned)total);
  return vca_launch_status();
}

}  // namespace

extern "C" int vca_conv_wgrad_bf16(const vca_conv_desc* d, int nsplit, const float* x, const float* gy,
                                   const float* in_scale, const float* in_shift, const float* in_alpha,
                                   float* dwpack, float* bias_grad, void* stream) {
  VCA_CHECK_ARG(d && x && gy && dwpack && (nsplit == 1 || nsplit == 3));
  VCA_CHECK_ARG(d->N > 0 && d->D == 1 && d->OD == 1 && d->KD == 1 && d->pad_d == 0 && d->Cin > 1 && d->Cout > 0);
  VCA_CHECK_ARG(d->Hin > 0 && d->Win > 0 && d->OH > 0 && d->OW > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0);
  VCA_CHECK_ARG(d->pad_h >= 0 && d->pad_w >= 0 && d->dil == 1 && (d->ups == 0 || d->ups == 1));
  VCA_CHECK_ARG(d->act >= 0 && d->act <= 3 && (in_scale == nullptr) == (in_shift == nullptr));
  VCA_CHECK_ARG(d->act != VCA_ACT_PRELU || in_alpha);
  VCA_CHECK_ARG(d->KH * d->KW <= 32);
  WArgs a;
  a.d = *d;
  a.x = x; a.gy = gy; a.in_scale = in_scale; a.in_shift = in_shift; a.in_alpha = in_alpha; a.dw = dwpack;
  a.bias_grad = bias_grad;
  a.Cout4 = (d->Cout + 3) & ~3;
  a.ntaps = d->KH * d->KW;
  // at most 9 taps: 64 x 64 tiles, one quadrant per wave
  const bool quad = a.ntaps <= 9;
  TileChoice t;
  if (quad) {
    // split product: 184 halo positions x 256 B + 32 KiB output-gradient tile + 2 KiB slot table = 80 KiB, two
    // workgroups per CU; the single product needs half of that and takes the 192 positions three items per thread cover
    t = choose_tile(d->N, d->OH, d->OW, d->KH, d->KW, d->stride, 128, nsplit == 3 ? 184 : 192, 12.0);
  } else {
    t = choose_tile(d->N, d->OH, d->OW, d->KH, d->KW, d->stride, d->Cout);
  }
  a.NI = t.NI; a.TH = t.TH; a.TW = t.TW;
  a.HH = (t.TH - 1) * d->stride + d->KH;
  a.HW = (t.TW - 1) * d->stride + d->KW;
  a.NPOS = a.HH * a.HW;
  a.tiles_x = (d->OW + t.TW - 1) / t.TW;
  a.tiles_y = (d->OH + t.TH - 1) / t.TH;
  a.nsub = d->N * a.tiles_x * a.tiles_y;
  a.ntiles = (a.nsub + t.NI - 1) / t.NI;
  const int tile_c = quad ? 64 : 32;
  a.co_tiles = (d->Cout + tile_c - 1) / tile_c;
  const int ci_tiles = (d->Cin + tile_c - 1) / tile_c;
  const int out_tiles = a.co_tiles * ci_tiles;
  // Pixel split: every workgroup ends with one atomic add per accumulator, which costs about as much as `fin`
  // patches of MFMA work, and the launch runs in ceil(workgroups / resident slots) rounds.  Take the split that
  // minimises rounds * (patches per workgroup + fin).
  // 8-wave form of the quadrant kernel: single product, whole 16-byte pieces, no prologue on the input
  const bool quad8 = quad && nsplit == 1 && d->Cin % 16 == 0 && d->Cout % 8 == 0 && !in_scale && d->act == 0;
  const int slots = (quad && !quad8) ? 512 : 256;   // 2 x 4-wave or 1 x 8-wave workgroups per CU
  const double fin = quad ? 2.0 : 1.0;
  int split = 1;
  double best_cost = 1e300;
  for (int sp = 1; sp <= a.ntiles && sp <= 512; ++sp) {
    const int ppb = (a.ntiles + sp - 1) / sp;
    const int sp_eff = (a.ntiles + ppb - 1) / ppb;
    const long long blocks = (long long)out_tiles * sp_eff;
    const double cost = (double)((blocks + slots - 1) / slots) * (ppb + fin);
    if (cost < best_cost) {
      best_cost = cost;
      split = sp_eff;
    }
  }
  a.tiles_per_block = (a.ntiles + split - 1) / split;
  split = (a.ntiles + a.tiles_per_block - 1) / a.tiles_per_block;
  a.vecA = (d->Cin % 4 == 0) ? 1 : 0;
  a.vecB = (d->Cout % 4 == 0) ? 1 : 0;
  a.slot_off = (int)(((size_t)a.NI * a.NPOS * 64 + BM * 64) * (nsplit == 3 ? 2 : 1) * (quad ? 2 : 1));
  const size_t lds = (size_t)a.slot_off + BM * 4 * sizeof(int) * (quad8 ? 2 : 1);   // + the per-patch slot table(s)
  hipStream_t s = (hipStream_t)stream;
  VCA_MAX_LDS(conv_bf16_wgrad_quad8_kernel);
  VCA_MAX_LDS(conv_bf16_wgrad_quad_kernel<3>);
  VCA_MAX_LDS(conv_bf16_wgrad_quad_kernel<1>);
  VCA_MAX_LDS((conv_bf16_wgrad_kernel<3, true>));
  VCA_MAX_LDS((conv_bf16_wgrad_kernel<3, false>));
  VCA_MAX_LDS((conv_bf16_wgrad_kernel<1, true>));
  VCA_MAX_LDS((conv_bf16_wgrad_kernel<1, false>));
  // many taps: whole taps per wave (output-gradient fragments stay in registers); few taps: balanced unit runs
  const bool rr = a.ntaps > 16;
  const dim3 grid(out_tiles, split), blk(512);
  if (quad8) {
    hipLaunchKernelGGL(conv_bf16_wgrad_quad8_kernel, grid, dim3(512), lds, s, a);
  } else if (quad) {
    if (nsplit == 3) hipLaunchKernelGGL((conv_bf16_wgrad_quad_kernel<3>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv_bf16_wgrad_quad_kernel<1>), grid, dim3(256), lds, s, a);
  } else if (nsplit == 3) {
    if (rr) hipLaunchKernelGGL((conv_bf16_wgrad_kernel<3, true>), grid, blk, lds, s, a);
    else hipLaunchKernelGGL((conv_bf16_wgrad_kernel<3, false>), grid, blk, lds, s, a);
  } else {
    if (rr) hipLaunchKernelGGL((conv_bf16_wgrad_kernel<1, true>), grid, blk, lds, s, a);
    else hipLaunchKernelGGL((conv_bf16_wgrad_kernel<1, false>), grid, blk, lds, s, a);
  }
  return vca_launch_status();
}

extern "C" long long vca_pack_weight_bf16_bytes(int mode, int Cout, int Cin, int taps) {
  const int N = mode == 0 ? Cout : (mode == 1 ? Cin : 4 * Cin), K = mode == 0 ? Cin : Cout;
  const long long npad = (N + 31) / 32 * 32, nchunk = (K + 31) / 32;
  return (long long)(mode == 2 ? 4 : taps) * nchunk * npad * 128;
}

extern "C" int vca_pack_weight_bf16(int mode, int Cout, int Cin, int taps, const float* src, void* dst, void* stream) {
  VCA_CHECK_ARG(mode >= 0 && mode <= 2 && Cout > 0 && Cin > 0 && taps > 0 && src && dst);
  VCA_CHECK_ARG(mode != 2 || taps == 9);
  const int N = mode == 0 ? Cout : (mode == 1 ? Cin : 4 * Cin), K = mode == 0 ? Cin : Cout;
  const int npad = (N + 31) / 32 * 32, nchunk = (K + 31) / 32;
  if (taps <= 32) {
    const dim3 grid(nchunk, npad / 8);
    hipLaunchKernelGGL(pack_weight_bf16_tiled_kernel, grid, dim3(256), (size_t)8 * 32 * taps * sizeof(float),
                       (hipStream_t)stream, mode, Cin, taps, N, K, npad, nchunk, src, reinterpret_cast<vca_e16*>(dst));
    return vca_launch_status();
  }
  const size_t total = (size_t)taps * nchunk * npad * 32;
  hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     mode, Cout, Cin, taps, N, K, npad, nchunk, src, reinterpret_cast<vca_e16*>(dst));
  return vca_launch_status();
}

extern "C" int vca_pack_weight_bf16_batch(const vca_pack_desc* table, int ndesc, int total_tiles, int max_taps,
                                          void* stream) {
  VCA_CHECK_ARG(table && ndesc > 0 && total_tiles > 0 && max_taps > 0 && max_taps <= 32);
  hipLaunchKernelGGL(pack_weight_bf16_batch_kernel, dim3(total_tiles), dim3(256),
                     (size_t)8 * 32 * max_taps * sizeof(float), (hipStream_t)stream, table, ndesc);
  return vca_launch_status();
}

extern "C" int vca_conv_fwd_bf16_ws(const vca_conv_desc* d, int nsplit, const float* x, const void* wpack,
                                    const float* bias, const float* in_scale, const float* in_shift,
                                    const float* in_alpha, const float* res, float* y, float* stats, float* ws,
                                    long long ws_floats, void* stream) {
  VCA_CHECK_ARG(d && x && wpack && y && (nsplit == 1 || nsplit == 3));
  VCA_CHECK_ARG(d->N > 0 && d->D == 1 && d->OD == 1 && d->KD == 1 && d->pad_d == 0 && d->Cin > 1 && d->Cout > 0);
  VCA_CHECK_ARG(d->Hin > 0 && d->Win > 0 && d->OH > 0 && d->OW > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0);
  VCA_CHECK_ARG(d->pad_h >= 0 && d->pad_w >= 0 && d->dil >= 1 && (d->ups == 0 || d->ups == 1));
  VCA_CHECK_ARG(d->act >= 0 && d->act <= 3 && (in_scale == nullptr) == (in_shift == nullptr));
  VCA_CHECK_ARG(d->act != VCA_ACT_PRELU || in_alpha);
  VCA_CHECK_ARG((long long)d->N * d->OH * d->OW < (1ll << 31));
  VCA_CHECK_ARG(d->KH * d->KW <= MAXPOS);
  VCA_CHECK_ARG(d->d2s_h >= 0 && d->d2s_w >= 0 && (d->d2s_h > 0) == (d->d2s_w > 0));
  VCA_CHECK_ARG(d->d2s_h == 0 || (d->Cout % 4 == 0 && !stats && !bias && d->d2s_h <= 2 * d->OH && d->d2s_w <= 2 * d->OW));
  HArgs a;
  a.d = *d;
  a.x = x; a.w = reinterpret_cast<const unsigned char*>(wpack); a.bias = bias;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_alpha = in_alpha; a.res = res; a.y = y; a.stats = stats;
  a.npad = (d->Cout + 31) / 32 * 32;
  a.nchunk = (d->Cin + 31) / 32;
  a.ntaps = d->KH * d->KW;
  TileChoice t = choose_tile(d->N, d->OH, d->OW, d->KH, d->KW, d->stride, d->Cout);
  bool padded = false;
  if (d->stride == 1 && d->KW >= 5 && d->Cout > 64) {   // see halo_pitch: only where the padding costs no patch area
    const TileChoice tp = choose_tile(d->N, d->OH, d->OW, d->KH, d->KW, d->stride, d->Cout, MAXPOS, 6.0, true);
    padded = tp.NI == t.NI && tp.TH == t.TH && tp.TW == t.TW;
  }
  a.NI = t.NI; a.TH = t.TH; a.TW = t.TW;
  a.HH = (t.TH - 1) * d->stride + d->KH;
  a.HWv = (t.TW - 1) * d->stride + d->KW;
  halo_pitch(t.TH, t.TW, d->KH, d->KW, d->stride, padded, a.HW, a.NPOS);
  a.tiles_x = (d->OW + t.TW - 1) / t.TW;
  a.tiles_y = (d->OH + t.TH - 1) / t.TH;
  a.vecA = (d->Cin % 4 == 0) ? 1 : 0;
  a.vecY = ((reinterpret_cast<size_t>(y) | reinterpret_cast<size_t>(res) | reinterpret_cast<size_t>(bias) |
             reinterpret_cast<size_t>(ws)) & 15) == 0 ? 1 : 0;
  a.nsub = d->N * a.tiles_x * a.tiles_y;
  const int gx = (a.nsub + t.NI - 1) / t.NI;
  hipStream_t s = (hipStream_t)stream;
  const size_t wsf = ws && ws_floats > 0 ? (size_t)ws_floats : 0;
  const size_t halo = (size_t)a.NI * a.NPOS * 64 * (nsplit == 3 ? 2 : 1);
#define LAUNCH_H(BN_, WM_, WN_)                                                                   \
  do {                                                                                             \
    const int gy = (d->Cout + BN_ - 1) / BN_;                                                      \
    return nsplit == 3 ? launch_halo<BN_, WM_, WN_, 3>(a, gx, gy, halo, ws, wsf, s)                         \
                       : launch_halo<BN_, WM_, WN_, 1>(a, gx, gy, halo, ws, wsf, s);                        \
  } while (0)
  if (d->Cout > 64) LAUNCH_H(128, 2, 2);
  else if (d->Cout > 32) LAUNCH_H(64, 2, 2);
  else LAUNCH_H(32, 4, 1);
#undef LAUNCH_H
}

extern "C" int vca_conv_fwd_bf16(const vca_conv_desc* d, int nsplit, const float* x, const void* wpack,
                                 const float* bias, const float* in_scale, const float* in_shift,
                                 const float* in_alpha, const float* res, float* y, float* stats, void* stream) {
  return vca_conv_fwd_bf16_ws(d, nsplit, x, wpack, bias, in_scale, in_shift, in_alpha, res, y, stats, nullptr, 0,
                              stream);
}

extern "C" long long vca_conv_fwd_bf16_ws_floats(const vca_conv_desc* d) {
  if (!d || d->N <= 0 || d->OH <= 0 || d->OW <= 0 || d->Cout <= 0 || d->Cin <= 1 || d->KH <= 0 || d->KW <= 0 ||
      d->stride <= 0 || d->out_scale != 1.0f)
    return 0;
  const TileChoice t = choose_tile(d->N, d->OH, d->OW, d->KH, d->KW, d->stride, d->Cout);
  const long long nsub = (long long)d->N * ((d->OW + t.TW - 1) / t.TW) * ((d->OH + t.TH - 1) / t.TH);
  const long long gx = (nsub + t.NI - 1) / t.NI;
  const int bn = d->Cout > 64 ? 128 : (d->Cout > 32 ? 64 : 32);
  const int ks = choose_ksplit(gx * ((d->Cout + bn - 1) / bn), (d->Cin + 31) / 32);
  return ks > 1 ? (long long)((out_floats(*d) + 63) / 64 * 64) * ks : 0;
}
