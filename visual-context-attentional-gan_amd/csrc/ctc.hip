// Connectionist temporal classification loss over the ASR scorer's frame logits (the auxiliary text loss of the
// LRS3 multi-task configuration; alphabet and blank of ASR_model/GRID/src/data/vid_aud_GRID_test.py:20-33: 28 classes,
// '_' = index 0 = blank).  log-softmax over the classes is part of the kernels, so the input is raw logits and the
// gradient is the gradient with respect to them.
//
// One workgroup per sample: the alpha / beta recursions run over time with one thread per extended-label state
// (blank, l1, blank, l2, ..., blank: S = 2L+1 states) in log space, two LDS rows ping-ponging, one barrier per frame.
// The forward kernel keeps log alpha of every frame in a caller-provided workspace; the backward kernel runs the beta
// recursion and turns alpha + beta into the gradient frame by frame:
//   d nll / d logit[t][v] = softmax[t][v] - sum_{s: l'_s = v} exp(alpha_t(s) + beta_t(s) - logp[t][v] + nll)
// (both alpha and beta include frame t's emission, hence the single division).  The class sums are taken by a fixed
// group of lanes per class in a fixed order: results are reproducible bit for bit.
//
// The recursions run in fp64: a log-space recursion drifts by |log alpha| * eps per frame, and with log-likelihoods of
// a few hundred over 150-300 frames fp32 leaves a few 1e-4 of the largest gradient entry (measured 4e-4 at T = 150;
// torch's own float path is in the same place); the alpha workspace holds doubles for the same reason.  The work
// is T * S exp/log evaluations per sample, a few hundred microseconds per batch either way.
#include "common.h"

#include <math.h>

namespace {

constexpr int CT = 256;          // threads per workgroup
constexpr int MAXT = 2048;       // frames per sample (per-frame log-sum-exp kept in LDS)
constexpr int MAXS = 1023;       // extended states per sample (L <= 511)
#define NEG_INF (-INFINITY)

__device__ __forceinline__ double lse2(double a, double b) {
  const double m = fmax(a, b);
  if (m == NEG_INF) return NEG_INF;
  return m + log(exp(a - m) + exp(b - m));
}
__device__ __forceinline__ double lse3(double a, double b, double c) {
  const double m = fmax(fmax(a, b), c);
  if (m == NEG_INF) return NEG_INF;
  return m + log(exp(a - m) + exp(b - m) + exp(c - m));
}

// log-sum-exp over the classes of every valid frame of sample b into lse[t]
__device__ void frame_lse(const float* __restrict__ lg, int Tb, int V, double* lse) {
  for (int t = threadIdx.x; t < Tb; t += CT) {
    const float* r = lg + (size_t)t * V;
    float m = NEG_INF;
    for (int v = 0; v < V; ++v) m = fmaxf(m, r[v]);
    double s = 0.0;
    for (int v = 0; v < V; ++v) s += exp((double)r[v] - (double)m);
    lse[t] = (double)m + log(s);
  }
}

struct CtcArgs {
  int B, T, V, Lmax, blank, zero_inf;
  const float* logits;      // [B][T][V]
  const int* targets;       // [B][Lmax]
  const int* in_len;        // [B]
  const int* tgt_len;       // [B]
  double* alpha;            // [B][T][2*Lmax+1] log alpha
  float* nll;               // [B]
  const float* gnll;        // [B]   (backward)
  float* glogits;           // [B][T][V] (backward)
};

// extended label sequence blank, l1, blank, l2, ..., blank (labels clamped into the alphabet: they index the logits)
__device__ void fill_ext(const CtcArgs& a, int b, int S, int* ext) {
  for (int s = threadIdx.x; s < S; s += CT)
    ext[s] = (s & 1) ? min(max(a.targets[(size_t)b * a.Lmax + (s >> 1)], 0), a.V - 1) : a.blank;
}

__global__ __launch_bounds__(CT) void ctc_alpha_kernel(CtcArgs a) {
  __shared__ double lse[MAXT];
  __shared__ double row[2][MAXS + 1];
  __shared__ int ext[MAXS + 1];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Smax = 2 * a.Lmax + 1;
  const int Tb = min(max(a.in_len[b], 0), a.T), L = min(max(a.tgt_len[b], 0), a.Lmax), S = 2 * L + 1;
  const float* lg = a.logits + (size_t)b * a.T * a.V;
  double* al = a.alpha + (size_t)b * a.T * Smax;
  fill_ext(a, b, S, ext);
  frame_lse(lg, Tb, a.V, lse);
  __syncthreads();
  if (Tb == 0) {                       // no frames: only the empty labelling has probability one
    if (tid == 0) a.nll[b] = (L == 0) ? 0.f : (a.zero_inf ? 0.f : INFINITY);
    return;
  }
  for (int s = tid; s < S; s += CT) {
    const double v = (s < 2) ? (double)lg[ext[s]] - lse[0] : (double)NEG_INF;
    row[0][s] = v;
    al[s] = v;
  }
  __syncthreads();
  for (int t = 1; t < Tb; ++t) {
    const double* prev = row[(t - 1) & 1];
    double* cur = row[t & 1];
    const float* r = lg + (size_t)t * a.V;
    const double ls = lse[t];
    for (int s = tid; s < S; s += CT) {
      const int c = ext[s];
      const double p0 = prev[s], p1 = s >= 1 ? prev[s - 1] : (double)NEG_INF;
      const double p2 = (s >= 2 && c != a.blank && c != ext[s - 2]) ? prev[s - 2] : (double)NEG_INF;
      const double v = lse3(p0, p1, p2) + ((double)r[c] - ls);
      cur[s] = v;
      al[(size_t)t * Smax + s] = v;
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double* last = row[(Tb - 1) & 1];
    const double ll = lse2(last[S - 1], S >= 2 ? last[S - 2] : (double)NEG_INF);
    float n = (float)-ll;
    if (a.zero_inf && !(n < INFINITY)) n = 0.f;
    a.nll[b] = n;
  }
}

template <int LPC>   // lanes per class
__global__ __launch_bounds__(CT) void ctc_beta_grad_kernel(CtcArgs a) {
  __shared__ double lse[MAXT];
  __shared__ double row[2][MAXS + 1];
  __shared__ float wgt[MAXS + 1];     // exp(alpha + beta - logp(label) - log p(l|x)) of the current frame
  __shared__ int ext[MAXS + 1];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Smax = 2 * a.Lmax + 1;
  const int Tb = min(max(a.in_len[b], 0), a.T), L = min(max(a.tgt_len[b], 0), a.Lmax), S = 2 * L + 1;
  const float* lg = a.logits + (size_t)b * a.T * a.V;
  const double* al = a.alpha + (size_t)b * a.T * Smax;
  float* gl = a.glogits + (size_t)b * a.T * a.V;
  // frames behind the sample's length take no gradient
  for (size_t i = (size_t)Tb * a.V + tid; i < (size_t)a.T * a.V; i += CT) gl[i] = 0.f;
  if (Tb == 0) return;
  fill_ext(a, b, S, ext);
  frame_lse(lg, Tb, a.V, lse);
  __syncthreads();
  // log p(l|x) from the last alpha row (the forward kernel may have replaced an infinite loss by zero)
  const double* alast = al + (size_t)(Tb - 1) * Smax;
  const double ll = lse2(alast[S - 1], S >= 2 ? alast[S - 2] : (double)NEG_INF);
  const bool dead = !(ll > NEG_INF);            // no valid alignment: zero gradient (zero_infinity) or NaN
  const float g = a.gnll[b];
  for (int t = Tb - 1; t >= 0; --t) {
    const double* nxt = row[(t + 1) & 1];
    double* cur = row[t & 1];
    const float* r = lg + (size_t)t * a.V;
    const double ls = lse[t];
    for (int s = tid; s < S; s += CT) {
      const int c = ext[s];
      const double lp = (double)r[c] - ls;
      double v;
      if (t == Tb - 1) {
        v = (s >= S - 2) ? lp : (double)NEG_INF;
      } else {
        const double p0 = nxt[s], p1 = s + 1 < S ? nxt[s + 1] : (double)NEG_INF;
        const double p2 = (s + 2 < S && c != a.blank && c != ext[s + 2]) ? nxt[s + 2] : (double)NEG_INF;
        v = lse3(p0, p1, p2) + lp;
      }
      cur[s] = v;
      const double e = al[(size_t)t * Smax + s] + v - lp - ll;
      wgt[s] = (e > (double)NEG_INF && !dead) ? (float)exp(e) : 0.f;
    }
    __syncthreads();
    // class sums: LPC consecutive lanes per class, each over a strided share of the states, then a shuffle tree
    const int v = tid / LPC, j = tid % LPC;
    float acc = 0.f;
    if (v < a.V)
      for (int s = j; s < S; s += LPC)
        if (ext[s] == v) acc += wgt[s];
#pragma unroll
    for (int o = LPC >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (v < a.V && j == 0) {
      const float sm = (float)exp((double)r[v] - ls);
      float o = g * (sm - acc);
      if (dead) o = a.zero_inf ? 0.f : NAN;
      gl[(size_t)t * a.V + v] = o;
    }
    __syncthreads();
  }
}

int check_args(const CtcArgs& a) {
  VCA_CHECK_ARG(a.B > 0 && a.T > 0 && a.T <= MAXT && a.V > 0 && a.V <= 64 && a.Lmax > 0 && 2 * a.Lmax + 1 <= MAXS);
  VCA_CHECK_ARG(a.blank >= 0 && a.blank < a.V && a.logits && a.targets && a.in_len && a.tgt_len && a.alpha && a.nll);
  return 0;
}

}  // namespace

extern "C" int vca_ctc_loss_fwd_f32(int B, int T, int V, int Lmax, const float* logits, const int* targets,
                                    const int* in_len, const int* tgt_len, int blank, int zero_infinity,
                                    double* alpha_ws, float* nll, void* stream) {
  CtcArgs a{B, T, V, Lmax, blank, zero_infinity, logits, targets, in_len, tgt_len, alpha_ws, nll, nullptr, nullptr};
  if (int e = check_args(a)) return e;
  hipLaunchKernelGGL(ctc_alpha_kernel, dim3(B), dim3(CT), 0, (hipStream_t)stream, a);
  return vca_launch_status();
}

extern "C" int vca_ctc_loss_bwd_f32(int B, int T, int V, int Lmax, const float* logits, const int* targets,
                                    const int* in_len, const int* tgt_len, int blank, int zero_infinity,
                                    const double* alpha_ws, const float* nll, const float* gnll, float* glogits,
                                    void* stream) {
  CtcArgs a{B, T, V, Lmax, blank, zero_infinity, logits, targets, in_len, tgt_len, const_cast<double*>(alpha_ws),
            const_cast<float*>(nll), gnll, glogits};
  if (int e = check_args(a)) return e;
  VCA_CHECK_ARG(gnll && glogits);
  if (V <= 32) hipLaunchKernelGGL(ctc_beta_grad_kernel<8>, dim3(B), dim3(CT), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(ctc_beta_grad_kernel<4>, dim3(B), dim3(CT), 0, (hipStream_t)stream, a);
  return vca_launch_status();
}
