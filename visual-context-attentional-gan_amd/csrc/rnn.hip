// GRU cell kernels (the gate GEMMs run on the convolution/GEMM kernels; these are the pointwise parts).
// torch.nn.GRU gate order in weight_ih / weight_hh rows: [reset | update | new]  (visual_front.py:20).
#include "common.h"

namespace {

constexpr int THR = 256;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// h = (1-z)*n + z*h_prev,  r = s(gi_r+gh_r), z = s(gi_z+gh_z), n = tanh(gi_n + r*gh_n)
__global__ __launch_bounds__(THR) void gru_cell_fwd_kernel(int B, int H, const float* __restrict__ gi,
                                                           const float* __restrict__ gh,
                                                           const float* __restrict__ h_prev,
                                                           float* __restrict__ h_new, float* __restrict__ out,
                                                           int out_stride, float* __restrict__ save) {
  const int i = blockIdx.x * THR + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, j = i - b * H;
  const float* gib = gi + (size_t)b * 3 * H;
  const float* ghb = gh + (size_t)b * 3 * H;
  const float r = sigmoidf_(gib[j] + ghb[j]);
  const float z = sigmoidf_(gib[H + j] + ghb[H + j]);
  const float hn = ghb[2 * H + j];
  const float n = tanhf(gib[2 * H + j] + r * hn);
  const float hp = h_prev ? h_prev[i] : 0.f;
  const float h = (1.f - z) * n + z * hp;
  h_new[i] = h;
  out[(size_t)b * out_stride + j] = h;
  float* sv = save + (size_t)b * 4 * H;
  sv[j] = r; sv[H + j] = z; sv[2 * H + j] = n; sv[3 * H + j] = hn;
}

// dh = dout + dh_a + dh_b;  d_gi = [dr, dz, dn], d_gh = [dr, dz, dn*r], dh_direct = dh*z
__global__ __launch_bounds__(THR) void gru_cell_bwd_kernel(int B, int H, const float* __restrict__ dout,
                                                           int dout_stride, const float* __restrict__ dh_a,
                                                           const float* __restrict__ dh_b,
                                                           const float* __restrict__ save,
                                                           const float* __restrict__ h_prev,
                                                           float* __restrict__ d_gi, float* __restrict__ d_gh,
                                                           float* __restrict__ dh_direct) {
  const int i = blockIdx.x * THR + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, j = i - b * H;
  float dh = dout[(size_t)b * dout_stride + j];
  if (dh_a) dh += dh_a[i];
  if (dh_b) dh += dh_b[i];
  const float* sv = save + (size_t)b * 4 * H;
  const float r = sv[j], z = sv[H + j], n = sv[2 * H + j], hn = sv[3 * H + j];
  const float hp = h_prev ? h_prev[i] : 0.f;
  const float dn = dh * (1.f - z) * (1.f - n * n);
  const float dz = dh * (hp - n) * z * (1.f - z);
  const float dr = dn * hn * r * (1.f - r);
  float* gi = d_gi + (size_t)b * 3 * H;
  float* gh = d_gh + (size_t)b * 3 * H;
  gi[j] = dr; gi[H + j] = dz; gi[2 * H + j] = dn;
  gh[j] = dr; gh[H + j] = dz; gh[2 * H + j] = dn * r;
  dh_direct[i] = dh * z;
}

// Both directions of a bidirectional layer in one launch (blockIdx.y = direction): the recurrence is a chain of
// small dependent launches, so each launch saved shortens the step.
struct CellFwd2 {
  const float* gi[2]; const float* gh[2]; const float* h_prev[2];
  float* h_new[2]; float* out[2]; float* save[2];
};

__global__ __launch_bounds__(THR) void gru_cell_fwd2_kernel(int B, int H, int out_stride, CellFwd2 a) {
  const int i = blockIdx.x * THR + threadIdx.x, d = blockIdx.y;
  if (i >= B * H) return;
  const int b = i / H, j = i - b * H;
  const float* gib = a.gi[d] + (size_t)b * 3 * H;
  const float* ghb = a.gh[d] + (size_t)b * 3 * H;
  const float r = sigmoidf_(gib[j] + ghb[j]);
  const float z = sigmoidf_(gib[H + j] + ghb[H + j]);
  const float hn = ghb[2 * H + j];
  const float n = tanhf(gib[2 * H + j] + r * hn);
  const float hp = a.h_prev[d] ? a.h_prev[d][i] : 0.f;
  const float h = (1.f - z) * n + z * hp;
  a.h_new[d][i] = h;
  a.out[d][(size_t)b * out_stride + j] = h;
  float* sv = a.save[d] + (size_t)b * 4 * H;
  sv[j] = r; sv[H + j] = z; sv[2 * H + j] = n; sv[3 * H + j] = hn;
}

struct CellBwd2 {
  const float* dout[2]; const float* dh_a[2]; const float* dh_b[2]; const float* save[2]; const float* h_prev[2];
  float* d_gi[2]; float* d_gh[2]; float* dh_direct[2];
};

__global__ __launch_bounds__(THR) void gru_cell_bwd2_kernel(int B, int H, int dout_stride, CellBwd2 a) {
  const int i = blockIdx.x * THR + threadIdx.x, d = blockIdx.y;
  if (i >= B * H) return;
  const int b = i / H, j = i - b * H;
  float dh = a.dout[d][(size_t)b * dout_stride + j];
  if (a.dh_a[d]) dh += a.dh_a[d][i];
  if (a.dh_b[d]) dh += a.dh_b[d][i];
  const float* sv = a.save[d] + (size_t)b * 4 * H;
  const float r = sv[j], z = sv[H + j], n = sv[2 * H + j], hn = sv[3 * H + j];
  const float hp = a.h_prev[d] ? a.h_prev[d][i] : 0.f;
  const float dn = dh * (1.f - z) * (1.f - n * n);
  const float dz = dh * (hp - n) * z * (1.f - z);
  const float dr = dn * hn * r * (1.f - r);
  float* gi = a.d_gi[d] + (size_t)b * 3 * H;
  float* gh = a.d_gh[d] + (size_t)b * 3 * H;
  gi[j] = dr; gi[H + j] = dz; gi[2 * H + j] = dn;
  gh[j] = dr; gh[H + j] = dz; gh[2 * H + j] = dn * r;
  a.dh_direct[d][i] = dh * z;
}

}  // namespace

extern "C" int vca_gru_cell_fwd2_f32(int B, int H, int out_stride, const float* gi0, const float* gi1,
                                     const float* gh0, const float* gh1, const float* hp0, const float* hp1,
                                     float* hn0, float* hn1, float* out0, float* out1, float* save0, float* save1,
                                     void* stream) {
  VCA_CHECK_ARG(B > 0 && H > 0 && gi0 && gi1 && gh0 && gh1 && hn0 && hn1 && out0 && out1 && save0 && save1 &&
                out_stride >= H);
  CellFwd2 a{{gi0, gi1}, {gh0, gh1}, {hp0, hp1}, {hn0, hn1}, {out0, out1}, {save0, save1}};
  hipLaunchKernelGGL(gru_cell_fwd2_kernel, dim3((B * H + THR - 1) / THR, 2), dim3(THR), 0, (hipStream_t)stream, B, H,
                     out_stride, a);
  return vca_launch_status();
}

extern "C" int vca_gru_cell_bwd2_f32(int B, int H, int dout_stride, const float* dout0, const float* dout1,
                                     const float* dha0, const float* dha1, const float* dhb0, const float* dhb1,
                                     const float* save0, const float* save1, const float* hp0, const float* hp1,
                                     float* dgi0, float* dgi1, float* dgh0, float* dgh1, float* dhd0, float* dhd1,
                                     void* stream) {
  VCA_CHECK_ARG(B > 0 && H > 0 && dout0 && dout1 && save0 && save1 && dgi0 && dgi1 && dgh0 && dgh1 && dhd0 && dhd1 &&
                dout_stride >= H);
  CellBwd2 a{{dout0, dout1}, {dha0, dha1}, {dhb0, dhb1}, {save0, save1}, {hp0, hp1},
             {dgi0, dgi1}, {dgh0, dgh1}, {dhd0, dhd1}};
  hipLaunchKernelGGL(gru_cell_bwd2_kernel, dim3((B * H + THR - 1) / THR, 2), dim3(THR), 0, (hipStream_t)stream, B, H,
                     dout_stride, a);
  return vca_launch_status();
}

extern "C" int vca_gru_cell_fwd_f32(int B, int H, const float* gi, const float* gh, const float* h_prev, float* h_new,
                                    float* out, int out_stride, float* save, void* stream) {
  VCA_CHECK_ARG(B > 0 && H > 0 && gi && gh && h_new && out && save && out_stride >= H);
  hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3((B * H + THR - 1) / THR), dim3(THR), 0, (hipStream_t)stream, B, H, gi,
                     gh, h_prev, h_new, out, out_stride, save);
  return vca_launch_status();
}

extern "C" int vca_gru_cell_bwd_f32(int B, int H, const float* dout, int dout_stride, const float* dh_a,
                                    const float* dh_b, const float* save, const float* h_prev, float* d_gi,
                                    float* d_gh, float* dh_direct, void* stream) {
  VCA_CHECK_ARG(B > 0 && H > 0 && dout && save && d_gi && d_gh && dh_direct && dout_stride >= H);
  hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3((B * H + THR - 1) / THR), dim3(THR), 0, (hipStream_t)stream, B, H, dout,
                     dout_stride, dh_a, dh_b, save, h_prev, d_gi, d_gh, dh_direct);
  return vca_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Skinny GEMM for the recurrent step: y[M<=32*gy][N] = x[M][K] . W^T (+ bias), split-bf16, for up to two
// independent problems per launch (the two directions of a bidirectional layer: blockIdx.z).
// A 75-step recurrence is latency-bound, so the kernel is built for the shortest dependent chain rather than for
// throughput: no LDS staging and no barriers in the main loop - each wave takes a quarter of the reduction, loads
// its A rows straight from global memory (fp32 -> hi/lo bf16 in registers) and its B fragments straight from the
// L2-resident packed weights ([32-deep slice][row][hi 32 | lo 32] bf16, the layout of vca_pack_weight_bf16); the
// four partial 32x32 tiles meet in LDS once at the end.
// ------------------------------------------------------------------------------------------------
namespace {


struct SkinnyArgs {
  const float* x[2];
  const unsigned char* w[2];
  const float* bias[2];
  float* y[2];
  int M, N, K, npad, nsplit;
  // optional epilogue (EP): the product is dh of a GRU backward step (N = H hidden units) and the gate arithmetic
  // of that step (gru_cell_bwd2_kernel) is applied to it in place of the store
  const float* dout[2]; const float* dh_b[2]; const float* save[2]; const float* h_prev[2];
  float* d_gi[2]; float* d_gh[2]; float* dh_direct[2];
  int dout_stride;
};

// NCOL: output columns per workgroup (32, or 16 = half of the MFMA tile left unused: twice the workgroups, half the
// weight bytes through each CU, for the launches whose duration is a dependent chain of memory round trips).
template <bool EP, int NW, int NCOL = 32>
__global__ __launch_bounds__(64 * NW) void skinny_gemm_kernel(SkinnyArgs a) {
  __shared__ float red[NW][32][NCOL + 1];   // NW waves split the reduction (8 or 12 for the deep K = 3H of the backward step)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int z = blockIdx.z, n0 = blockIdx.x * NCOL, m0 = blockIdx.y * 32;
  const int brow = min(n0 + (lr < NCOL ? lr : 0), a.npad - 1);
  const float* x = a.x[z];
  const unsigned char* w = a.w[z];
  const int nks = (a.K + 15) / 16;
  const int per = (nks + NW - 1) / NW;
  const int ks_beg = wave * per, ks_end = min(nks, ks_beg + per);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int row = m0 + lr;
  const bool rv = row < a.M;
  const bool vec = (a.K % 4) == 0;
  // The reduction is walked in chunks of CH k-steps; all global loads of a chunk (A rows and weight fragments) are
  // issued before the first MFMA, so a wave pays the memory latency once per chunk instead of once per k-step.
  constexpr int CH = 8;
  for (int base = ks_beg; base < ks_end; base += CH) {
    float4 ap[CH], aq[CH];
    bf16x8 bh[CH], bl[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int ks = base + j;
      const int k0 = ks * 16 + lh * 8;
      ap[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      aq[j] = ap[j];
      const bool kv = ks < ks_end;
      if (rv && kv) {
        const float* src = x + (size_t)row * a.K + k0;
        if (vec && k0 + 8 <= a.K) {
          ap[j] = *reinterpret_cast<const float4*>(src);
          aq[j] = *reinterpret_cast<const float4*>(src + 4);
        } else {
          float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (k0 + e < a.K) v[e] = src[e];
          ap[j] = make_float4(v[0], v[1], v[2], v[3]);
          aq[j] = make_float4(v[4], v[5], v[6], v[7]);
        }
      }
      // weight row n0+lr of slice ks>>1: 128 B = [hi 32 | lo 32]; this lane's 8 values start at (ks&1)*16 + lh*8
      const int ksc = kv ? ks : ks_beg;        // past the end: any valid address, the A operand is zero
      const unsigned char* wp = w + ((size_t)(ksc >> 1) * a.npad + brow) * 128 + ((ksc & 1) * 16 + lh * 8) * 2;
      bh[j] = *reinterpret_cast<const bf16x8*>(wp);
      if (a.nsplit == 3) bl[j] = *reinterpret_cast<const bf16x8*>(wp + 64);
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float v[8] = {ap[j].x, ap[j].y, ap[j].z, ap[j].w, aq[j].x, aq[j].y, aq[j].z, aq[j].w};
      bf16x8 ah, al;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ah[e] = (vca_e16)v[e];
        al[e] = (vca_e16)(v[e] - (float)ah[e]);
      }
      if (a.nsplit == 3) {
        acc = VCA_MFMA_32x32x16(al, bh[j], acc);
        acc = VCA_MFMA_32x32x16(ah, bl[j], acc);
      }
      acc = VCA_MFMA_32x32x16(ah, bh[j], acc);
    }
  }
  if (lr < NCOL) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][lr] = acc[r];
  }
  __syncthreads();
  for (int i = tid; i < 32 * NCOL; i += 64 * NW) {
    const int r = i / NCOL, c = i - r * NCOL;
    const int m = m0 + r, n = n0 + c;
    if (m < a.M && n < a.N) {
      float s = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < NW; ++w8) s += red[w8][r][c];
      if (!EP) {
        if (a.bias[z]) s += a.bias[z][n];
        a.y[z][(size_t)m * a.N + n] = s;
      } else {
        const int H = a.N;
        const size_t e = (size_t)m * H + n;
        float dh = a.dout[z][(size_t)m * a.dout_stride + n] + s;
        if (a.dh_b[z]) dh += a.dh_b[z][e];
        const float* sv = a.save[z] + (size_t)m * 4 * H;
        const float rg = sv[n], zg = sv[H + n], ng = sv[2 * H + n], hn = sv[3 * H + n];
        const float hp = a.h_prev[z] ? a.h_prev[z][e] : 0.f;
        const float dn = dh * (1.f - zg) * (1.f - ng * ng);
        const float dz = dh * (hp - ng) * zg * (1.f - zg);
        const float dr = dn * hn * rg * (1.f - rg);
        float* gi = a.d_gi[z] + (size_t)m * 3 * H;
        float* gh = a.d_gh[z] + (size_t)m * 3 * H;
        gi[n] = dr; gi[H + n] = dz; gi[2 * H + n] = dn;
        gh[n] = dr; gh[H + n] = dz; gh[2 * H + n] = dn * rg;
        a.dh_direct[z][e] = dh * zg;
      }
    }
  }
}

// One forward time step of both directions of a GRU layer in one launch: the recurrent product gh = h_prev . W_hh^T
// + b_hh and the gate arithmetic of gru_cell_fwd2_kernel in its epilogue.  A workgroup owns 8 hidden units of up to 32
// sequences and computes the three gate columns of those units (24 of the 32 columns of one MFMA tile: rows
// g*H + unit of the packed weight, picked per lane), so the cell needs nothing from another workgroup; the NW waves
// split the reduction as in skinny_gemm_kernel.  2*H/8 workgroups of ~50 KB of weights each keep the dependent chain
// (75 steps per layer) short.
struct GruFwdArgs {
  const float* hp[2]; const unsigned char* w[2]; const float* bhh[2]; const float* gi[2];
  float* h_new[2]; float* out[2]; float* save[2];
  int B, H, npad, nsplit, out_stride;
};

template <int NW>
__global__ __launch_bounds__(64 * NW) void gru_fwd_step_kernel(GruFwdArgs a) {
  __shared__ float red[NW][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const int z = blockIdx.z, u0 = blockIdx.x * 8, m0 = blockIdx.y * 32;
  const int H = a.H;
  const float* x = a.hp[z];
  const unsigned char* w = a.w[z];
  const int nks = (H + 15) / 16;
  const int per = (nks + NW - 1) / NW;
  const int ks_beg = wave * per, ks_end = min(nks, ks_beg + per);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int row = m0 + lr;
  const bool rv = row < a.B;
  const int gate = lr >> 3, unit = u0 + (lr & 7);
  const int brow = (gate < 3 && unit < H) ? gate * H + unit : 0;   // columns 24..31 are never read back
  constexpr int CH = 8;
  if (x) {
    for (int base = ks_beg; base < ks_end; base += CH) {
      float4 ap[CH], aq[CH];
      bf16x8 bh[CH], bl[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int ks = base + j;
        const int k0 = ks * 16 + lh * 8;
        ap[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        aq[j] = ap[j];
        const bool kv = ks < ks_end;
        if (rv && kv) {
          const float* src = x + (size_t)row * H + k0;
          ap[j] = *reinterpret_cast<const float4*>(src);
          aq[j] = *reinterpret_cast<const float4*>(src + 4);
        }
        const int ksc = kv ? ks : ks_beg;
        const unsigned char* wp = w + ((size_t)(ksc >> 1) * a.npad + brow) * 128 + ((ksc & 1) * 16 + lh * 8) * 2;
        bh[j] = *reinterpret_cast<const bf16x8*>(wp);
        if (a.nsplit == 3) bl[j] = *reinterpret_cast<const bf16x8*>(wp + 64);
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const float v[8] = {ap[j].x, ap[j].y, ap[j].z, ap[j].w, aq[j].x, aq[j].y, aq[j].z, aq[j].w};
        bf16x8 ah, al;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          ah[e] = (vca_e16)v[e];
          al[e] = (vca_e16)(v[e] - (float)ah[e]);
        }
        if (a.nsplit == 3) {
          acc = VCA_MFMA_32x32x16(al, bh[j], acc);
          acc = VCA_MFMA_32x32x16(ah, bl[j], acc);
        }
        acc = VCA_MFMA_32x32x16(ah, bh[j], acc);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][lr] = acc[r];
  __syncthreads();
  for (int i = tid; i < 32 * 8; i += 64 * NW) {
    const int m = i >> 3, u = i & 7;
    const int b = m0 + m, j = u0 + u;
    if (b >= a.B || j >= H) continue;
    float gh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      float t = a.bhh[z] ? a.bhh[z][g * H + j] : 0.f;
#pragma unroll
      for (int w8 = 0; w8 < NW; ++w8) t += red[w8][m][g * 8 + u];
      gh[g] = t;
    }
    const float* gib = a.gi[z] + (size_t)b * 3 * H;
    const float r = sigmoidf_(gib[j] + gh[0]);
    const float zg = sigmoidf_(gib[H + j] + gh[1]);
    const float n = tanhf(gib[2 * H + j] + r * gh[2]);
    const float hp = x ? x[(size_t)b * H + j] : 0.f;
    const float h = (1.f - zg) * n + zg * hp;
    a.h_new[z][(size_t)b * H + j] = h;
    a.out[z][(size_t)b * a.out_stride + j] = h;
    float* sv = a.save[z] + (size_t)b * 4 * H;
    sv[j] = r; sv[H + j] = zg; sv[2 * H + j] = n; sv[3 * H + j] = gh[2];
  }
}

}  // namespace

extern "C" int vca_gru_fwd_step_bf16(int nsplit, int B, int H, const float* hp0, const float* hp1, const void* w0,
                                     const void* w1, const float* bhh0, const float* bhh1, const float* gi0,
                                     const float* gi1, float* hn0, float* hn1, float* out0, float* out1, int out_stride,
                                     float* save0, float* save1, void* stream) {
  // the kernel reads h_prev in whole 16-wide reduction slices (two float4 per lane): H must be a multiple of 16
  VCA_CHECK_ARG((nsplit == 1 || nsplit == 3) && B > 0 && H > 0 && (H % 16) == 0 && w0 && w1 && gi0 && gi1);
  VCA_CHECK_ARG(hn0 && hn1 && out0 && out1 && save0 && save1 && out_stride >= H && (hp0 == nullptr) == (hp1 == nullptr));
  GruFwdArgs a;
  a.hp[0] = hp0; a.hp[1] = hp1;
  a.w[0] = reinterpret_cast<const unsigned char*>(w0); a.w[1] = reinterpret_cast<const unsigned char*>(w1);
  a.bhh[0] = bhh0; a.bhh[1] = bhh1; a.gi[0] = gi0; a.gi[1] = gi1;
  a.h_new[0] = hn0; a.h_new[1] = hn1; a.out[0] = out0; a.out[1] = out1; a.save[0] = save0; a.save[1] = save1;
  a.B = B; a.H = H; a.npad = (3 * H + 31) / 32 * 32; a.nsplit = nsplit; a.out_stride = out_stride;
  hipLaunchKernelGGL((gru_fwd_step_kernel<4>), dim3(H / 8, (B + 31) / 32, 2), dim3(256), 0, (hipStream_t)stream, a);
  return vca_launch_status();
}

extern "C" int vca_skinny_gemm_bf16(int nsplit, int nprob, int M, int N, int K, const float* x0, const float* x1,
                                    const void* w0, const void* w1, const float* b0, const float* b1, float* y0,
                                    float* y1, void* stream) {
  VCA_CHECK_ARG((nsplit == 1 || nsplit == 3) && (nprob == 1 || nprob == 2) && M > 0 && N > 0 && K > 0);
  VCA_CHECK_ARG(x0 && w0 && y0 && (nprob == 1 || (x1 && w1 && y1)));
  SkinnyArgs a;
  a.x[0] = x0; a.x[1] = x1;
  a.w[0] = reinterpret_cast<const unsigned char*>(w0); a.w[1] = reinterpret_cast<const unsigned char*>(w1);
  a.bias[0] = b0; a.bias[1] = b1;
  a.y[0] = y0; a.y[1] = y1;
  a.M = M; a.N = N; a.K = K; a.npad = (N + 31) / 32 * 32; a.nsplit = nsplit;
  if (K >= 1024)
    hipLaunchKernelGGL((skinny_gemm_kernel<false, 8>), dim3((N + 31) / 32, (M + 31) / 32, nprob), dim3(512), 0,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((skinny_gemm_kernel<false, 4>), dim3((N + 31) / 32, (M + 31) / 32, nprob), dim3(256), 0,
                       (hipStream_t)stream, a);
  return vca_launch_status();
}

extern "C" int vca_gru_bwd_step_bf16(int nsplit, int B, int H, const float* dgh_prev0, const float* dgh_prev1,
                                     const void* w0, const void* w1, int dout_stride, const float* dout0,
                                     const float* dout1, const float* dhb0, const float* dhb1, const float* save0,
                                     const float* save1, const float* hp0, const float* hp1, float* dgi0, float* dgi1,
                                     float* dgh0, float* dgh1, float* dhd0, float* dhd1, void* stream) {
  VCA_CHECK_ARG((nsplit == 1 || nsplit == 3) && B > 0 && H > 0 && dgh_prev0 && dgh_prev1 && w0 && w1);
  VCA_CHECK_ARG(dout0 && dout1 && save0 && save1 && dgi0 && dgi1 && dgh0 && dgh1 && dhd0 && dhd1 && dout_stride >= H);
  SkinnyArgs a;
  a.x[0] = dgh_prev0; a.x[1] = dgh_prev1;
  a.w[0] = reinterpret_cast<const unsigned char*>(w0); a.w[1] = reinterpret_cast<const unsigned char*>(w1);
  a.bias[0] = a.bias[1] = nullptr;
  a.y[0] = a.y[1] = nullptr;
  a.M = B; a.N = H; a.K = 3 * H; a.npad = (H + 31) / 32 * 32; a.nsplit = nsplit;
  a.dout[0] = dout0; a.dout[1] = dout1; a.dh_b[0] = dhb0; a.dh_b[1] = dhb1; a.save[0] = save0; a.save[1] = save1;
  a.h_prev[0] = hp0; a.h_prev[1] = hp1; a.d_gi[0] = dgi0; a.d_gi[1] = dgi1; a.d_gh[0] = dgh0; a.d_gh[1] = dgh1;
  a.dh_direct[0] = dhd0; a.dh_direct[1] = dhd1; a.dout_stride = dout_stride;
  // K = 3H: with 12 waves a wave's share of a 1536-deep reduction is one 8-step chunk, i.e. one round of loads
  if (3 * H <= 12 * 8 * 16 && 3 * H > 8 * 8 * 16)
    hipLaunchKernelGGL((skinny_gemm_kernel<true, 12, 16>), dim3((H + 15) / 16, (B + 31) / 32, 2), dim3(768), 0,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((skinny_gemm_kernel<true, 8>), dim3((H + 31) / 32, (B + 31) / 32, 2), dim3(512), 0,
                       (hipStream_t)stream, a);
  return vca_launch_status();
}
