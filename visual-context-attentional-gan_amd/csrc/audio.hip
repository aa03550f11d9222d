// Waveform-side kernels of the validation path (src/data/stft.py, audio_processing.py, vid_aud_grid.py:190-232):
// everything around the two dense contractions of the STFT pair, which run on the fp32 implicit-GEMM convolution
// kernels (the hop divides the window, so a frame is KT = win/hop consecutive rows of a [rows][hop] view of the
// signal: the analysis transform is a KT-tap convolution over rows and the synthesis overlap-add is its input
// gradient).
#include "common.h"

namespace {

constexpr int THR = 256;

int grid_for(size_t n) {
  size_t g = (n + THR - 1) / THR;
  if (g > 4096) g = 4096;
  return g < 1 ? 1 : (int)g;
}

// y[b][0..Lrow) = reflect-pad of x[b][0..L) by `pad` on both sides (F.pad(mode='reflect'), stft.py:75-79), cut or
// zero-filled to the row pitch Lrow (a multiple of the hop: samples behind the last whole frame are not used).
__global__ __launch_bounds__(THR) void reflect_pad_kernel(int B, int L, int pad, int Lrow, const float* __restrict__ x,
                                                          float* __restrict__ y) {
  const size_t total = (size_t)B * Lrow;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int n = (int)(i % Lrow), b = (int)(i / Lrow);
    float v = 0.f;
    if (n < L + 2 * pad) {
      int s = n - pad;
      if (s < 0) s = -s;
      if (s >= L) s = 2 * (L - 1) - s;
      v = x[(size_t)b * L + s];
    }
    y[i] = v;
  }
}

// spec[b][t][0..F) real, [F..2F) imaginary (channels-last rows of the transform) ->
//   mag[b][f][t] = |z|, phase[b][f][t] = atan2(im, re)   (stft.py:88-94; outputs in the reference's (B, F, T) layout)
__global__ __launch_bounds__(THR) void mag_phase_kernel(int B, int T, int F, const float* __restrict__ spec,
                                                        float* __restrict__ mag, float* __restrict__ phase) {
  const size_t total = (size_t)B * F * T;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int t = (int)(i % T);
    size_t r = i / T;
    const int f = (int)(r % F), b = (int)(r / F);
    const float* row = spec + ((size_t)b * T + t) * 2 * F;
    const float re = row[f], im = row[F + f];
    mag[i] = sqrtf(re * re + im * im);
    if (phase) phase[i] = atan2f(im, re);
  }
}

// rect[b][t][0..F) = mag*cos(phase), [F..2F) = mag*sin(phase) from (B, F, T) inputs (stft.py:99-100)
__global__ __launch_bounds__(THR) void polar_to_rect_kernel(int B, int T, int F, const float* __restrict__ mag,
                                                            const float* __restrict__ phase,
                                                            float* __restrict__ rect) {
  const size_t total = (size_t)B * T * F;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int f = (int)(i % F);
    size_t r = i / F;
    const int t = (int)(r % T), b = (int)(r / T);
    const size_t src = ((size_t)b * F + f) * T + t;
    const float m = mag[src], p = phase[src];
    float* row = rect + ((size_t)b * T + t) * 2 * F;
    row[f] = m * cosf(p);
    row[F + f] = m * sinf(p);
  }
}

// One Griffin-Lim projection (audio_processing.py:62-64): keep the phase of the current transform, impose the target
// magnitude.  cos(atan2(im,re)) = re/|z| and sin = im/|z| (atan2(0,0) = 0), so no angle is ever formed.
__global__ __launch_bounds__(THR) void gl_project_kernel(int B, int T, int F, const float* __restrict__ spec,
                                                         const float* __restrict__ mag, float* __restrict__ rect) {
  const size_t total = (size_t)B * T * F;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int f = (int)(i % F);
    size_t r = i / F;
    const int t = (int)(r % T), b = (int)(r / T);
    const float* srow = spec + ((size_t)b * T + t) * 2 * F;
    const float re = srow[f], im = srow[F + f];
    const float z = sqrtf(re * re + im * im);
    const float m = mag[((size_t)b * F + f) * T + t];
    float* row = rect + ((size_t)b * T + t) * 2 * F;
    row[f] = z > 0.f ? m * (re / z) : m;
    row[F + f] = z > 0.f ? m * (im / z) : 0.f;
  }
}

// Tail of STFT.inverse (stft.py:108-127): divide the overlap-added signal by the window sum-square envelope where it
// is not tiny, scale by win/hop, drop `pad` samples on both sides.  ola is [B][Lrow], y is [B][Lout].
__global__ __launch_bounds__(THR) void istft_finish_kernel(int B, int Lrow, int Lout, int pad, float scale, float tiny,
                                                           const float* __restrict__ ola,
                                                           const float* __restrict__ wsum, float* __restrict__ y) {
  const size_t total = (size_t)B * Lout;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int n = (int)(i % Lout), b = (int)(i / Lout);
    float v = ola[(size_t)b * Lrow + n + pad];
    const float w = wsum[n + pad];
    if (w > tiny) v /= w;
    y[i] = v * scale;
  }
}

// scipy.signal.lfilter([1], [1, -c], x) per row, then clip to [-1, 1] (vid_aud_grid.py:203-210,230-232): a first-order
// recursion y[n] = x[n] + c*y[n-1].  One wave per signal: each lane scans a contiguous segment with a zero carry-in,
// the segment carries are combined with a wave prefix (carry-in of lane l = sum_j<l c^(len*(l-j)) * last_j), then
// each lane adds carry * c^(k+1) to its segment.
__global__ __launch_bounds__(64) void deemphasis_kernel(int L, float c, const float* __restrict__ x,
                                                        float* __restrict__ y) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int seg = (L + 63) / 64;
  const int beg = lane * seg, end = min(L, beg + seg);
  const float* xb = x + (size_t)b * L;
  float* yb = y + (size_t)b * L;
  float acc = 0.f;
  for (int n = beg; n < end; ++n) {
    acc = xb[n] + c * acc;
    yb[n] = acc;
  }
  const float cseg = powf(c, (float)seg);
  // inclusive scan of (last value, decay) pairs: carry_out(l) = last_l + cseg * carry_out(l-1)
  float carry = acc;   // value at the end of this lane's segment assuming zero carry-in
  float decay = cseg;
  for (int o = 1; o < 64; o <<= 1) {
    const float pc = __shfl_up(carry, o, 64), pd = __shfl_up(decay, o, 64);
    if (lane >= o) {
      carry += decay * pc;
      decay *= pd;
    }
  }
  float cin = __shfl_up(carry, 1, 64);   // true value just before this lane's segment
  if (lane == 0) cin = 0.f;
  float f = c;
  for (int n = beg; n < end; ++n) {
    float v = yb[n] + cin * f;
    f *= c;
    yb[n] = fminf(1.f, fmaxf(-1.f, v));
  }
}

}  // namespace

extern "C" int vca_reflect_pad_f32(int B, int L, int pad, int Lrow, const float* x, float* y, void* stream) {
  VCA_CHECK_ARG(B > 0 && L > pad && pad >= 0 && Lrow > 0 && x && y);   // a shorter pitch truncates the padded signal
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(grid_for((size_t)B * Lrow)), dim3(THR), 0, (hipStream_t)stream, B, L, pad,
                     Lrow, x, y);
  return vca_launch_status();
}

extern "C" int vca_mag_phase_f32(int B, int T, int F, const float* spec, float* mag, float* phase, void* stream) {
  VCA_CHECK_ARG(B > 0 && T > 0 && F > 0 && spec && mag);
  hipLaunchKernelGGL(mag_phase_kernel, dim3(grid_for((size_t)B * T * F)), dim3(THR), 0, (hipStream_t)stream, B, T, F,
                     spec, mag, phase);
  return vca_launch_status();
}

extern "C" int vca_polar_to_rect_f32(int B, int T, int F, const float* mag, const float* phase, float* rect,
                                     void* stream) {
  VCA_CHECK_ARG(B > 0 && T > 0 && F > 0 && mag && phase && rect);
  hipLaunchKernelGGL(polar_to_rect_kernel, dim3(grid_for((size_t)B * T * F)), dim3(THR), 0, (hipStream_t)stream, B, T,
                     F, mag, phase, rect);
  return vca_launch_status();
}

extern "C" int vca_gl_project_f32(int B, int T, int F, const float* spec, const float* mag, float* rect, void* stream) {
  VCA_CHECK_ARG(B > 0 && T > 0 && F > 0 && spec && mag && rect);
  hipLaunchKernelGGL(gl_project_kernel, dim3(grid_for((size_t)B * T * F)), dim3(THR), 0, (hipStream_t)stream, B, T, F,
                     spec, mag, rect);
  return vca_launch_status();
}

extern "C" int vca_istft_finish_f32(int B, int Lrow, int Lout, int pad, float scale, float tiny, const float* ola,
                                    const float* wsum, float* y, void* stream) {
  VCA_CHECK_ARG(B > 0 && Lout > 0 && pad >= 0 && Lrow >= Lout + pad && ola && wsum && y);
  hipLaunchKernelGGL(istft_finish_kernel, dim3(grid_for((size_t)B * Lout)), dim3(THR), 0, (hipStream_t)stream, B, Lrow,
                     Lout, pad, scale, tiny, ola, wsum, y);
  return vca_launch_status();
}

extern "C" int vca_deemphasis_clip_f32(int B, int L, float coef, const float* x, float* y, void* stream) {
  VCA_CHECK_ARG(B > 0 && L > 0 && x && y && coef > -1.f && coef < 1.f);
  hipLaunchKernelGGL(deemphasis_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, L, coef, x, y);
  return vca_launch_status();
}
