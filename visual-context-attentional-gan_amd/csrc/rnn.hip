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

}  // namespace

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
