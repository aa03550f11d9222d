// Weight gradient of a 2-D convolution whose input has ONE channel (the first layer of the mel discriminators,
// generator.py:291: Conv2d(1, dim_in, 5, 1, 2); of the sync discriminator, generator.py:346: Conv2d(1, 128, 3, 2, 1)):
//   dw[tap][co] = sum over output pixels p of gy[p][co] * x[p (+) tap].
// There is no channel reduction to feed the matrix cores with: the work is one pass over gy (HBM-bound) with KH*KW
// multiply-adds per element against a few input rows held in LDS.  A workgroup walks over groups of up to four
// consecutive output rows; a thread owns one output channel and every filter tap (registers) for one of eight pixel
// lanes and requests eight output-gradient values ahead of their use.  Workgroups store their partial
// sums in slabs of a caller-provided workspace (plain stores, fixed summation order); a second kernel folds the slabs
// into the [tap][1][Cout4] accumulation buffer the other weight-gradient kernels use.  The column sums of gy (the
// gradient of the convolution's bias) ride along as one more row of every slab.
#include "common.h"

namespace {

constexpr int CG = 32;        // output channels per workgroup (one per lane of a half-wave)
constexpr int PL = 8;         // pixel lanes per workgroup
constexpr int MAXW = 640;     // widest padded input row staged in LDS
constexpr int MAXWG = 1280;   // workgroups (slabs) per channel group: five per CU, all resident at once
constexpr int RC = 4;         // consecutive output rows per work unit (their input rows are staged together)
constexpr int PF = 8;         // output-gradient values a thread requests before it uses the first

struct C1Args {
  int N, Hin, Win, OH, OW, Cout, Cout4, stride, pad_h, pad_w, nwg;
  int rc, chunks;  // output rows per work unit, work units per image
  const float* x;
  const float* gy;
  float* slabs;   // [cgroups][nwg][KH*KW + 1][CG]; the last row: column sums of gy
  float* dw;      // [KH*KW][Cout4]
  float* bias;    // [Cout] or null
};

template <int KH, int KW>
__global__ __launch_bounds__(CG* PL) void wgrad_cin1_partial_kernel(C1Args a) {
  constexpr int TAPS = KH * KW;
  constexpr int XROWS = (RC - 1) * 2 + KH;                       // stride <= 2
  constexpr int XS = XROWS * (MAXW + 8), RED = PL * (TAPS + 1) * CG;
  __shared__ float smem[XS > RED ? XS : RED];
  float(*xs)[MAXW + 8] = reinterpret_cast<float(*)[MAXW + 8]>(smem);
  float(*red)[TAPS + 1][CG] = reinterpret_cast<float(*)[TAPS + 1][CG]>(smem);   // after the last unit
  const int tid = threadIdx.x, co = tid & (CG - 1), pl = tid / CG;
  const int cg0 = blockIdx.y * CG;
  const bool cok = cg0 + co < a.Cout;
  const int LW = a.Win + 2 * a.pad_w;            // staged row: pad_w zeros, the row, pad_w zeros
  float acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) acc[t] = 0.f;
  float accb = 0.f;
  const int units = a.N * a.chunks;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int n = u / a.chunks, oy0 = (u - n * a.chunks) * a.rc;
    const int nr = min(a.rc, a.OH - oy0);
    const int xrows = (nr - 1) * a.stride + KH;
    __syncthreads();
    for (int i = tid; i < xrows * LW; i += CG * PL) {
      const int xr = i / LW, c = i - xr * LW;
      const int iy = oy0 * a.stride + xr - a.pad_h, ix = c - a.pad_w;
      float v = 0.f;
      if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) v = a.x[((size_t)n * a.Hin + iy) * a.Win + ix];
      xs[xr][c] = v;
    }
    __syncthreads();
    for (int rr = 0; rr < nr; ++rr) {
      const float* g = a.gy + ((size_t)(n * a.OH + oy0 + rr) * a.OW) * a.Cout + cg0 + co;
      const int xr0 = rr * a.stride;
      for (int ox0 = pl; ox0 < a.OW; ox0 += PL * PF) {
        float v[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) {
          const int ox = ox0 + j * PL;
          v[j] = (cok && ox < a.OW) ? g[(size_t)ox * a.Cout] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < PF; ++j) {
          const int c0 = min(ox0 + j * PL, a.OW - 1) * a.stride;   // past the row: v = 0, any staged column will do
          accb += v[j];
#pragma unroll
          for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) acc[kh * KW + kw] += v[j] * xs[xr0 + kh][c0 + kw];
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < TAPS; ++t) red[pl][t][co] = acc[t];
  red[pl][TAPS][co] = accb;
  __syncthreads();
  float* slab = a.slabs + ((size_t)blockIdx.y * a.nwg + blockIdx.x) * (TAPS + 1) * CG;
  for (int i = tid; i < (TAPS + 1) * CG; i += CG * PL) {
    const int t = i / CG, c = i - t * CG;
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < PL; ++p) s += red[p][t][c];
    slab[i] = s;
  }
}

// dw[tap][cg*CG + c] += sum over the nwg slabs of channel group cg (row `taps` of a slab goes to bias[]).  A
// workgroup owns 64 consecutive (tap, c) entries of one channel group; 16 thread rows stride over the slabs.
__global__ __launch_bounds__(1024) void wgrad_cin1_fold_kernel(int taps, C1Args a) {
  __shared__ float red[16][64];
  const int c = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int cg = blockIdx.y, e = blockIdx.x * 64 + c, n = (taps + 1) * CG;
  float s = 0.f;
  if (e < n) {
    const float* p = a.slabs + (size_t)cg * a.nwg * n + e;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = r;
    for (; w + 48 < a.nwg; w += 64) {          // four independent chains of loads
      s += p[(size_t)w * n];
      s1 += p[(size_t)(w + 16) * n];
      s2 += p[(size_t)(w + 32) * n];
      s3 += p[(size_t)(w + 48) * n];
    }
    for (; w < a.nwg; w += 16) s += p[(size_t)w * n];
    s += s1 + s2 + s3;
  }
  red[r][c] = s;
  __syncthreads();
  if (r == 0 && e < n) {
#pragma unroll
    for (int i = 1; i < 16; ++i) s += red[i][c];
    const int t = e / CG, ch = cg * CG + (e - t * CG);
    if (ch < a.Cout) {
      if (t < taps) atomicAdd(a.dw + (size_t)t * a.Cout4 + ch, s);
      else if (a.bias) atomicAdd(a.bias + ch, s);
    }
  }
}

bool supported(const vca_conv_desc* d) {
  if (!d || d->Cin != 1 || d->KD != 1 || d->D != 1 || d->OD != 1 || d->pad_d != 0 || d->dil != 1 || d->ups != 0 ||
      d->act != 0 || d->stride < 1 || d->stride > 2)
    return false;
  if (!((d->KH == 5 && d->KW == 5) || (d->KH == 3 && d->KW == 3))) return false;
  if (d->Win + 2 * d->pad_w > MAXW || d->N <= 0 || d->OH <= 0 || d->OW <= 0 || d->Cout <= 0) return false;
  // the last tap of the last output pixel stays inside the staged row, and rows are addressed by oy*stride + kh
  if ((d->OW - 1) * d->stride + d->KW > d->Win + 2 * d->pad_w) return false;
  return (long long)d->N * d->OH * d->OW * d->Cout < (1ll << 31);
}

// Output rows per work unit: as many as keep every CU supplied with a few workgroups.
int rows_per_unit(const vca_conv_desc* d) {
  const long long rows = (long long)d->N * d->OH;
  int rc = 1;
  while (rc < RC && rows > (long long)MAXWG * rc) rc *= 2;     // the fewest rows per unit that fit one unit per workgroup
  return rc;
}

int units_per_image(const vca_conv_desc* d) {
  const int rc = rows_per_unit(d);
  return (d->OH + rc - 1) / rc;
}

int workgroups(const vca_conv_desc* d) {
  const long long units = (long long)d->N * units_per_image(d);
  return (int)(units < MAXWG ? units : MAXWG);
}

}  // namespace

extern "C" long long vca_conv_wgrad_cin1_ws_floats(const vca_conv_desc* d) {
  if (!supported(d)) return 0;
  const long long cgroups = (d->Cout + CG - 1) / CG;
  return cgroups * workgroups(d) * (d->KH * d->KW + 1) * CG;
}

extern "C" int vca_conv_wgrad_cin1_f32(const vca_conv_desc* d, const float* x, const float* gy, float* ws,
                                       long long ws_floats, float* dwpack, float* bias_grad, void* stream) {
  VCA_CHECK_ARG(supported(d) && x && gy && ws && dwpack);
  VCA_CHECK_ARG(ws_floats >= vca_conv_wgrad_cin1_ws_floats(d));
  C1Args a;
  a.N = d->N; a.Hin = d->Hin; a.Win = d->Win; a.OH = d->OH; a.OW = d->OW; a.Cout = d->Cout;
  a.Cout4 = (d->Cout + 3) & ~3;
  a.stride = d->stride; a.pad_h = d->pad_h; a.pad_w = d->pad_w;
  a.nwg = workgroups(d);
  a.rc = rows_per_unit(d);
  a.chunks = units_per_image(d);
  a.x = x; a.gy = gy; a.slabs = ws; a.dw = dwpack; a.bias = bias_grad;
  const int cgroups = (d->Cout + CG - 1) / CG;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(a.nwg, cgroups);
  const int taps = d->KH * d->KW;
  if (d->KH == 5)
    hipLaunchKernelGGL((wgrad_cin1_partial_kernel<5, 5>), grid, dim3(CG * PL), 0, s, a);
  else
    hipLaunchKernelGGL((wgrad_cin1_partial_kernel<3, 3>), grid, dim3(CG * PL), 0, s, a);
  hipLaunchKernelGGL(wgrad_cin1_fold_kernel, dim3(((taps + 1) * CG + 63) / 64, cgroups), dim3(1024), 0, s, taps, a);
  return vca_launch_status();
}
