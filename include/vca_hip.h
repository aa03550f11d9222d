/* C ABI of libvca_hip.so: hand-written gfx950 kernels for the VCA-GAN generator/discriminator
 * forward/backward path.
 *
 * The reference (ms-dot-k/Visual-Context-Attentional-GAN) has no FFI layer: its hot path is a set
 * of torch.nn modules (src/models/{visual_front,resnet,generator}.py) whose work is done by ATen
 * convolution / batch-norm / activation / GRU / bmm kernels.  Each entry point below replaces the
 * ATen call(s) behind the cited reference lines.  The Python host code (torch.autograd.Function
 * wrappers in the package) binds these symbols with ctypes and passes raw device pointers of
 * torch-allocated tensors plus the current HIP stream.
 *
 * Conventions
 *   - every pointer is a device pointer unless named h_*; the library allocates nothing and keeps
 *     no state; all work is enqueued on `stream` (a hipStream_t passed as void*) and the call
 *     returns without synchronising;
 *   - activations are fp32, channels-last: a 2-D feature map is [N][H][W][C], a sequence is
 *     [N][T][C]; a matrix view [P][C] (P = all pixels) is used by the per-channel kernels;
 *   - return value: 0 on success, -22 for an argument the kernel does not support, otherwise the
 *     hipError_t of the failed launch.
 */
#ifndef VCA_HIP_H
#define VCA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* Library/ABI version (bumped when a signature changes). */
int vca_abi_version(void);

/* Element type of the 16-bit matrix operands of this build of the library: 0 = bf16 (libvca_hip.so), 1 = IEEE half
 * (libvca_hip_f16.so, compiled from the same sources with -DVCA_E16_FP16).  Every "_bf16" entry point below computes
 * in that type; packed weights are not interchangeable between the two builds.  The reference has no reduced-precision
 * path (train.py runs fp32 throughout); the half build exists for BASELINE configs[4] "fp16 with loss-scaling". */
int vca_elem_type(void);

/* ------------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on the matrix cores (exact fp32: v_mfma_f32_32x32x2_f32).
 * Replaces nn.Conv2d / nn.Conv1d / nn.Linear / nn.Conv3d forward, input-gradient and
 * weight-gradient at: visual_front.py:11 (stem, KD=5), resnet.py:5-14 (3x3, 1x1 s2),
 * generator.py:19-25,62-68,103-108 (5x5 / 1x1 residual blocks), :147-152 (attention projections),
 * :178-184 (Postnet 1-D convs: H=1), :204-206 (attconv), :211,217,223 (to_mel), :272,290-303
 * (discriminator heads), :324-336 (sync discriminator), visual_front.py:21 (fc).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int N, D, Hin, Win, Cin; /* stored input tensor [N][D][Hin][Win][Cin] (D = 1 for 2-D maps)      */
  int OD, OH, OW, Cout;    /* output tensor [N][OD][OH][OW][Cout]                                  */
  int KD, KH, KW;          /* filter taps (KD = 1 for 2-D)                                         */
  int stride;              /* spatial stride (H and W alike); depth stride is 1                    */
  int pad_h, pad_w, pad_d; /* zero padding                                                         */
  int dil;                 /* >1: the stored input is read as if zero-dilated by `dil` (this is the
                              input-gradient of a stride-`dil` convolution)                        */
  int ups;                 /* 1: the stored input is read through a nearest-neighbour x2 upsample
                              (generator.py:112,121)                                               */
  int act;                 /* prologue on the input: 0 none, 1 LeakyReLU(slope), 2 PReLU(in_alpha),
                              3 ReLU; applied after the optional per-channel affine in_scale/shift */
  float slope;
  float out_scale;         /* epilogue: y = (conv + bias + res) * out_scale                        */
  int d2s_h, d2s_w;        /* > 0 (vca_conv_fwd_bf16 only): depth-to-space epilogue.  The Cout columns
                              are 4 groups (ry, rx) of Cout/4 channels; column (ry, rx, c) of output
                              pixel (oy, ox) is stored as channel c of pixel (2*oy+ry, 2*ox+rx) of a
                              [N][d2s_h][d2s_w][Cout/4] tensor (pixels outside it are dropped; res has
                              that shape too).  With the mode-2 weight pack this is the input gradient
                              of a 3x3 stride-2 convolution (resnet.py:5-9) without multiplying the
                              zeros of a dilated output gradient.                                   */
} vca_conv_desc;

/* y = conv(prologue(x), w) + bias (+ res), * out_scale.
 * wpack: [KD*KH*KW][Cin][Cout4] with Cout4 = Cout rounded up to 4 (see vca_pack_weight_f32).
 * bias/in_scale/in_shift/in_alpha/res/stats may be NULL.  stats (zeroed by the caller) receives the
 * per-output-channel sum and sum of squares of (y - bias) for a following train-mode BatchNorm, as
 * VCA_CONV_STAT_SLOTS partial buffers of 2*Cout floats each (a workgroup adds to one of them; this fp32
 * kernel and the stem only use the first, vca_conv_fwd_bf16 all); vca_bn_finalize_f32 totals them. */
#define VCA_CONV_STAT_SLOTS 32
int vca_conv_fwd_f32(const vca_conv_desc* d, const float* x, const float* wpack, const float* bias,
                     const float* in_scale, const float* in_shift, const float* in_alpha,
                     const float* res, float* y, float* stats, void* stream);

/* dwpack[tap][ci][co4] += sum over output pixels of prologue(x)[pixel (+) tap][ci] * gy[pixel][co].
 * dwpack must be zeroed by the caller; the kernel accumulates with fp32 atomics (split over
 * pixels).  The descriptor is the forward convolution's. */
int vca_conv_wgrad_f32(const vca_conv_desc* d, const float* x, const float* gy,
                       const float* in_scale, const float* in_shift, const float* in_alpha,
                       float* dwpack, void* stream);

/* Weight re-layout between the reference's parameter layout [Cout][Cin][taps] (nn.Conv*d /
 * nn.Linear .weight) and the kernels' [taps][Cin][Cout4].
 *   mode 0: forward pack                wpack[t][ci][co] = w[co][ci][t]
 *   mode 1: input-gradient pack (channels swapped, taps reversed)
 *                                       wpack[t][co][ci] = w[co][ci][taps-1-t], row length Cin4
 *   mode 2: unpack-accumulate a weight gradient: w[co][ci][t] = beta*w[co][ci][t] + wpack[t][ci][co] */
int vca_pack_weight_f32(int mode, int Cout, int Cin, int taps, const float* src, float* dst, float beta,
                        void* stream);
/* Mode 2 of vca_pack_weight_f32 with coalesced traffic on both sides; with clear != 0 the workspace dwpack is
 * zeroed behind the read, so a workspace that is only ever filled by vca_conv_wgrad_* and drained by this call
 * stays zeroed between uses (the gradient that autograd accumulates into nn.Parameter.grad, train.py:209,235). */
int vca_unpack_wgrad_f32(int Cout, int Cin, int taps, float* dwpack, float* w_grad, float beta, int clear,
                         void* stream);
/* vca_unpack_wgrad_f32 (beta = 1, clear = 1) for many workspaces in one launch: the weight gradients of a whole
 * backward pass are left in per-parameter workspaces and added to the parameters' gradients together.  table: ndesc
 * records in DEVICE memory sorted by tile_begin = running sum of ceil(Cin*taps/64) * ceil(roundup(Cout,4)/32);
 * total_tiles = that sum over all records. */
typedef struct vca_unpack_desc {
  float* src;    /* [taps][Cin][Cout4] workspace */
  float* dst;    /* [Cout][Cin][taps] gradient */
  int Cout, Cin, taps, tile_begin;
} vca_unpack_desc;
int vca_unpack_wgrad_batch_f32(const vca_unpack_desc* table, int ndesc, int total_tiles, void* stream);

/* Split-bf16 variant of vca_conv_fwd_f32 on v_mfma_f32_32x32x16_bf16 (2-D maps, Cin > 1; used for the
 * forward pass and, with the input-gradient descriptor, for the input gradient).  nsplit = 3: every fp32
 * operand is split into bf16 hi + lo and hi*hi + hi*lo + lo*hi is accumulated in fp32; nsplit = 1: plain
 * bf16 operands.  wpack is the layout written by vca_pack_weight_bf16 (mode 0 forward, 1 input gradient),
 * vca_pack_weight_bf16_bytes gives its size. */
int vca_conv_fwd_bf16(const vca_conv_desc* d, int nsplit, const float* x, const void* wpack, const float* bias,
                      const float* in_scale, const float* in_shift, const float* in_alpha, const float* res,
                      float* y, float* stats, void* stream);
/* vca_conv_fwd_bf16 with a caller-provided workspace.  Launches with few output tiles and a deep reduction (the
 * discriminators' 5x5 layers on 5x18 / 10x37 maps, generator.py:286-290) split the channel slices over several
 * workgroups per tile; with ws (ws_floats >= vca_conv_fwd_bf16_ws_floats(d), fp32, contents irrelevant before and
 * after) every range stores its partial sums in a slab of ws and a second kernel folds the slabs into y; without it
 * (ws NULL or too small) the partial sums are added to a zero-filled y with fp32 atomics, as vca_conv_fwd_bf16 does.
 * vca_conv_fwd_bf16_ws_floats returns 0 for a descriptor whose launch does not split. */
int vca_conv_fwd_bf16_ws(const vca_conv_desc* d, int nsplit, const float* x, const void* wpack, const float* bias,
                         const float* in_scale, const float* in_shift, const float* in_alpha, const float* res,
                         float* y, float* stats, float* ws, long long ws_floats, void* stream);
long long vca_conv_fwd_bf16_ws_floats(const vca_conv_desc* d);
/* Split-bf16 variant of vca_conv_wgrad_f32 (same fp32 [tap][Cin][Cout4] output, accumulated with atomics into a
 * zeroed buffer; at most 32 taps).  bias_grad (may be NULL): [Cout] vector to which the column sums of gy, i.e. the
 * gradient of the convolution's bias, are ADDED (fp32, from the output-gradient tiles the kernel stages anyway). */
int vca_conv_wgrad_bf16(const vca_conv_desc* d, int nsplit, const float* x, const float* gy, const float* in_scale,
                        const float* in_shift, const float* in_alpha, float* dwpack, float* bias_grad, void* stream);
/* Weight gradient of a 2-D convolution with ONE input channel and a 5x5 or 3x3 filter (first layer of the mel
 * discriminators, generator.py:291, and of the sync discriminator, generator.py:346): one streaming pass over gy on
 * the vector units, input rows in LDS, partial sums per workgroup in slabs of ws (fixed summation order), folded and
 * ADDED to dwpack[tap][1][Cout4] like vca_conv_wgrad_f32 does.  bias_grad (may be NULL): [Cout] vector to which the
 * column sums of gy are ADDED, from the same pass.  vca_conv_wgrad_cin1_ws_floats returns the workspace size in
 * floats, 0 for a descriptor this kernel does not cover (the caller then uses vca_conv_wgrad_f32). */
long long vca_conv_wgrad_cin1_ws_floats(const vca_conv_desc* d);
int vca_conv_wgrad_cin1_f32(const vca_conv_desc* d, const float* x, const float* gy, float* ws, long long ws_floats,
                            float* dwpack, float* bias_grad, void* stream);
/* mode 2 (taps = 9): the input gradient of a 3x3 / stride 2 / padding 1 convolution in sub-pixel form: 4 taps
 * (dy, dx) over the output gradient, N = 4*Cin columns (ry, rx, ci), K = Cout; for a descriptor with KH = KW = 2,
 * stride 1, padding 0, OH x OW = the output gradient's size and d2s_h x d2s_w = the input's. */
long long vca_pack_weight_bf16_bytes(int mode, int Cout, int Cin, int taps);
int vca_pack_weight_bf16(int mode, int Cout, int Cin, int taps, const float* src, void* dst, void* stream);
/* vca_pack_weight_bf16 for many weights in one launch (all packed copies of one network after its optimizer step;
 * the reference has no counterpart: torch re-reads nn.Parameter storage directly).  table: ndesc records in DEVICE
 * memory, sorted by tile_begin = running sum of ceil(K/32) * (roundup(N,32)/8) over the preceding records (N, K as
 * in vca_pack_weight_bf16); total_tiles = that sum over all records; every record needs taps <= 32;
 * max_taps = the largest taps in the table. */
typedef struct vca_pack_desc {
  const float* src;   /* [Cout][Cin][taps] */
  void* dst;          /* vca_pack_weight_bf16_bytes(mode, Cout, Cin, taps) bytes */
  int mode, Cout, Cin, taps;
  int tile_begin, reserved;
} vca_pack_desc;
int vca_pack_weight_bf16_batch(const vca_pack_desc* table, int ndesc, int total_tiles, int max_taps, void* stream);

/* Linear layers and 1x1 / stride-1 convolutions in split-bf16 (nn.Linear: generator.py:145-148,299,302,335,
 * visual_front.py:21, the input products of nn.GRU visual_front.py:20; pointwise nn.Conv2d / nn.Conv1d:
 * generator.py:25,68,108,184,212-224), forward and - with the input-gradient weight pack - input gradient:
 *   C[m][n] = sum_k A[m][k] * W[n][k] (+ bias[n]) (+ res[m][n]),   m < M, n < N, k < K (K a multiple of 8).
 * wpack: vca_pack_weight_bf16 mode 0 of the (N, K) weight for the forward pass, mode 1 of the (K, N) weight for the
 * input gradient (taps = 1).  A and C (and res, laid out like C) may be strided views, which is how the attention
 * projections read and write channels-last maps without the view / permute copies of generator.py:159,169:
 *   A[m][k] at a + (m / a_tm) * a_bs + (m % a_tm) * a_ts + (k / a_ck) * a_ks + (k % a_ck)
 *   C[m][n] at c + (m / c_tm) * c_bs + (m % c_tm) * c_ts + (n / c_cn) * c_ns + (n % c_cn)        (in floats)
 * with a_ck a multiple of 32 (or >= K), every A stride a multiple of 4 and a 16-byte aligned; a plain row-major
 * matrix is a_tm = M, a_ts = row stride, a_ck = K.  Descriptors that do not meet this return -22. */
typedef struct vca_lin_desc {
  int M, N, K;
  int a_tm, a_ck;
  long long a_bs, a_ts, a_ks;
  int c_tm, c_cn;
  long long c_bs, c_ts, c_ns;
} vca_lin_desc;
int vca_linear_bf16(const vca_lin_desc* d, int nsplit, const float* a, const void* wpack, const float* bias,
                    const float* res, float* c, void* stream);
/* The same product for shapes of a few GFLOP (attention projections generator.py:145-148,167, GRU input products
 * visual_front.py:20, the sync discriminator's Linear generator.py:335): 128 x 128 tiles through a three-stage LDS
 * ring.  With few tiles the reduction is split over workgroups and the partial tiles go through ws
 * (ws_floats >= vca_gemm_bf16_ws_floats(d) fp32 values, 16-byte aligned, contents irrelevant before and after; the
 * query returns 0 for a descriptor that does not split, and ws may then be NULL). */
long long vca_gemm_bf16_ws_floats(const vca_lin_desc* d);
int vca_gemm_bf16(const vca_lin_desc* d, int nsplit, const float* a, const void* wpack, const float* bias,
                  const float* res, float* c, float* ws, long long ws_floats, void* stream);

/* Conv3d stem of the visual front-end (nn.Conv3d(1, 64, (5,7,7), (1,2,2), (2,3,3), bias=False),
 * visual_front.py:11) in split-bf16: x is the single-channel video [B][T][H][W], y the channels-last feature map
 * [B][T][OH][OW][64] with OH = (H-1)/2+1.  wpack (vca_stem_pack_bytes() bytes, 16-byte aligned: the hi/lo bf16
 * register fragments of every reduction step) is written by vca_stem_pack_weight from the parameter's
 * [64][1][5][7][7] layout.  stats ([stat_slots][2][64] floats, zeroed by the caller, may be NULL) receives partial
 * per-channel sums and sums of squares of y for the BatchNorm3d that follows (visual_front.py:12): the statistic is
 * the total over the slots (vca_bn_finalize_f32 takes the slot count). */
long long vca_stem_pack_bytes(void);
int vca_stem_pack_weight(const float* w, void* wpack, void* stream);
int vca_stem_fwd_bf16(int nsplit, int B, int T, int H, int W, const float* x, const void* wpack, float* y,
                      float* stats, int stat_slots, void* stream);
/* Patch form of the clip for the stem's weight gradient (visual_front.py:11 backward): xp is
 * [B][T][OH*OW][64] with channel kh*8+kw = clip[b][t][2*oy+kh-3][2*ox+kw-3] (zero outside the clip and for kh or
 * kw = 7), over which the 5x7x7 stride-(1,2,2) convolution is a 5-tap convolution along t with 64 input channels,
 * so that vca_conv_wgrad_bf16 (N=B, H=T, W=OH*OW, KH=5, KW=1, pad_h=2) computes the stem's weight gradient. */
int vca_stem_patches_f32(int B, int T, int H, int W, const float* x, float* xp, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-channel kernels on a channels-last matrix view x[P][C] (C % 4 == 0).
 * Replace nn.BatchNorm1d/2d/3d (train and eval), nn.PReLU / nn.LeakyReLU / nn.ReLU / nn.Tanh, the
 * residual adds and the 1/sqrt(2) scaling at: visual_front.py:12-13, resnet.py:34-64,
 * generator.py:22-49,77-92,105-131,179-180,208-225,325-333.
 * ---------------------------------------------------------------------------------------------- */
/* sums[0..C) = sum_p (x[p][c] - x[0][c]);  sums[C..2C) = sum_p (x[p][c] - x[0][c])^2.  sums holds VCA_BN_SLOTS*2*C
 * floats, zeroed by the caller: the part behind the first 2*C is scratch for per-workgroup partial sums. */
#define VCA_BN_SLOTS 16
int vca_colstats_f32(int P, int C, const float* x, float* sums, void* stream);

/* Train-mode statistics -> mean, 1/sqrt(var+eps), scale = gamma*invstd, shift = beta - mean*scale; updates the
 * running statistics in place (momentum, unbiased variance) when running_mean/var are given.
 * `sums`: nslots buffers of 2*C floats whose element-wise total is (sum, sum of squares) per channel: nslots = 1
 * for the first 2*C floats written by vca_colstats_f32, VCA_CONV_STAT_SLOTS for the `stats` of a convolution
 * epilogue; `pivot` is what was subtracted (or NULL). */
int vca_bn_finalize_f32(int C, float count, const float* sums, int nslots, const float* pivot, const float* gamma,
                        const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                        float* mean, float* invstd, float* scale, float* shift, void* stream);

/* Eval-mode coefficients from the running statistics. */
int vca_bn_eval_coeff_f32(int C, const float* running_mean, const float* running_var, const float* gamma,
                          const float* beta, float eps, float* mean, float* invstd, float* scale, float* shift,
                          void* stream);

/* y = act(x*scale + shift + res) * oscale.  act: 0 none, 1 LeakyReLU(slope), 2 PReLU(alpha[c]), 3 ReLU, 4 tanh;
 * scale/shift, alpha and res may be NULL. */
int vca_affine_act_f32(int P, int C, const float* x, const float* scale, const float* shift, const float* alpha,
                       const float* res, int act, float slope, float oscale, float* y, void* stream);

/* Backward of vca_affine_act_f32.  With gz = gy*oscale*act'(z):
 *   sums (VCA_BN_SLOTS*3*C floats, zeroed by the caller; may be NULL when train == 0 and act != PReLU): the
 *     first 3*C receive sum gz, sum gz*xhat (xhat = (x-mean)*invstd) and sum gy*oscale*min(z,0)
 *     = d beta, d gamma, d alpha; the rest is scratch for per-workgroup partial sums;
 *   gx   = scale*(gz - mean(gz) - xhat*mean(gz*xhat)) when train != 0 (batch statistics), scale*gz otherwise;
 *   gres = gz.   gx / gres may be NULL. */
int vca_affine_act_bwd_f32(int P, int C, const float* gy, const float* x, const float* scale, const float* shift,
                           const float* alpha, const float* res, const float* mean, const float* invstd, int act,
                           float slope, float oscale, int train, float* sums, float* gx, float* gres, void* stream);

/* The same backward for a training step that accumulates in place (the reference's autograd adds the two branch
 * gradients of a residual block, generator.py:128-130, and the parameter gradients, as separate passes):
 *   gadd (may be NULL; needs gx): gx += gadd, the gradient arriving at x from its other consumer (identity shortcut);
 *   gbeta_acc / ggamma_acc / galpha_acc (each may be NULL; need sums): the C-element d beta / d gamma / d alpha
 *     totals are also added into these buffers (the parameters' gradient storage). */
int vca_affine_act_bwd2_f32(int P, int C, const float* gy, const float* x, const float* scale, const float* shift,
                            const float* alpha, const float* res, const float* mean, const float* invstd, int act,
                            float slope, float oscale, int train, float* sums, float* gx, float* gres,
                            const float* gadd, float* gbeta_acc, float* ggamma_acc, float* galpha_acc, void* stream);

/* BatchNorm3d -> PReLU -> MaxPool3d((1,3,3),(1,2,2),(0,1,1)) of the front-end (visual_front.py:12-14) without the
 * activated full-resolution map: writes the pooled map [N][OH][OW][C] and the argmax taps (0..8) straight from
 * x [N][H][W][C] and the batch-norm coefficients.  Backward: vca_maxpool3s2_bwd_f32, then vca_affine_act_bwd_f32. */
int vca_affine_act_pool3_fwd_f32(int N, int H, int W, int C, const float* x, const float* scale, const float* shift,
                                 const float* alpha, int act, float slope, float* y, unsigned char* idx, void* stream);

/* 2x2 pooling family on [N][H][W][C] (F.avg_pool2d(x,2) generator.py:74,83; nearest x2 upsample :112,121).
 * (Hs,Ws) is the small map, (Hl,Wl) the large one, Hs = Hl/2, Ws = Wl/2.
 * mode 0: average large->small; 1: its adjoint small->large; 2: block sum large->small; 3: nearest upsample. */
int vca_pool2_f32(int mode, int N, int Hs, int Ws, int Hl, int Wl, int C, const float* x, float* y, void* stream);

/* Entry of a down-sampling discriminator block (generator.py:71-83: the block input goes through LeakyReLU into the
 * residual branch and through F.avg_pool2d(x, 2) into the shortcut) in one pass: y_act = LeakyReLU(x, slope) on
 * [N][H][W][C], y_pool = 2x2 average on [N][H/2][W/2][C] (floor).  Backward in one pass as well:
 * gx = g_act * LeakyReLU'(x) + g_pool[h/2][w/2] / 4 (either gradient may be NULL). */
int vca_leaky_pool2_fwd_f32(int N, int H, int W, int C, float slope, const float* x, float* y_act, float* y_pool,
                            void* stream);
int vca_leaky_pool2_bwd_f32(int N, int H, int W, int C, float slope, const float* g_act, const float* g_pool,
                            const float* x, float* gx, void* stream);

/* MaxPool (3x3, stride 2, pad 1) of visual_front.py:14 (per frame), forward (+argmax tap) and backward. */
int vca_maxpool3s2_fwd_f32(int N, int H, int W, int C, const float* x, float* y, unsigned char* idx, void* stream);
int vca_maxpool3s2_bwd_f32(int N, int H, int W, int C, const float* gy, const unsigned char* idx, float* gx,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * Visual Context Attention core (AVAttention.forward, generator.py:161-167): per sample
 *   ctx = softmax(q k^T * scale with keys >= lens[b] masked to -inf) v
 * q, ctx: (B, Tq, D); k, v: (B, S, D); D = 256, S <= 160; lens: B int32 on the device.  probs (B, Tq, S) receives
 * the attention weights for the backward call, which returns dq and the (internally zeroed, atomically
 * accumulated) dk, dv.  The q/k/v/output projections (generator.py:158-159,166,168) run on the convolution
 * kernels as 1x1 convolutions.
 * ---------------------------------------------------------------------------------------------- */
int vca_attention_fwd_f32(int B, int Tq, int S, int D, const float* q, const float* k, const float* v,
                          const int* lens, float scale, float* ctx, float* probs, void* stream);
int vca_attention_bwd_f32(int B, int Tq, int S, int D, const float* dctx, const float* q, const float* k,
                          const float* v, const float* probs, float scale, float* dq, float* dk, float* dv,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * GRU cell (nn.GRU(512,512,2,bidirectional) of visual_front.py:20,33-34).  The input and recurrent gate products
 * x W_ih^T + b_ih and h W_hh^T + b_hh run on the convolution kernels as 1x1 "convolutions"; these two kernels are
 * the pointwise gate arithmetic of one time step and its backward.  gi/gh: (B, 3H) rows [reset | update | new].
 *   forward:  h_new (B,H) contiguous, and the same values to out[b*out_stride + j] (the time step's slot of the
 *             layer output); save (B,4H) keeps r, z, n and the recurrent new-gate pre-activation for backward.
 *   backward: dh = dout[b*dout_stride + j] + dh_a + dh_b (either may be NULL) -> d_gi, d_gh (B,3H) and
 *             dh_direct = dh*z, the part of d h_prev that does not pass through W_hh.
 * ---------------------------------------------------------------------------------------------- */
int vca_gru_cell_fwd_f32(int B, int H, const float* gi, const float* gh, const float* h_prev, float* h_new,
                         float* out, int out_stride, float* save, void* stream);
int vca_gru_cell_bwd_f32(int B, int H, const float* dout, int dout_stride, const float* dh_a, const float* dh_b,
                         const float* save, const float* h_prev, float* d_gi, float* d_gh, float* dh_direct,
                         void* stream);
/* The two directions of a bidirectional layer (visual_front.py:20, bidirectional=True) in one launch each: same
 * arithmetic as vca_gru_cell_fwd_f32 / vca_gru_cell_bwd_f32 with every pointer given once per direction (h_prev,
 * dh_a, dh_b may be NULL per direction). */
int vca_gru_cell_fwd2_f32(int B, int H, int out_stride, const float* gi0, const float* gi1, const float* gh0,
                          const float* gh1, const float* hp0, const float* hp1, float* hn0, float* hn1, float* out0,
                          float* out1, float* save0, float* save1, void* stream);
int vca_gru_cell_bwd2_f32(int B, int H, int dout_stride, const float* dout0, const float* dout1, const float* dha0,
                          const float* dha1, const float* dhb0, const float* dhb1, const float* save0,
                          const float* save1, const float* hp0, const float* hp1, float* dgi0, float* dgi1,
                          float* dgh0, float* dgh1, float* dhd0, float* dhd1, void* stream);
/* One forward time step of both directions of a bidirectional GRU layer (nn.GRU, visual_front.py:20,33-34) in one
 * launch: gh = h_prev . W_hh^T + b_hh on the matrix cores (W packed by vca_pack_weight_bf16 mode 0, [3H][H]) with the
 * gate arithmetic of vca_gru_cell_fwd2_f32 in the epilogue.  hp0/hp1 NULL = zero initial state.  gi: [B][3H] input
 * gates of this step; writes h_new [B][H], out (row stride out_stride) and save [B][4H] = (r, z, n, gh_n).
 * H must be a multiple of 16 (the reduction is read in 16-wide slices); other sizes return -22. */
int vca_gru_fwd_step_bf16(int nsplit, int B, int H, const float* hp0, const float* hp1, const void* w0, const void* w1,
                          const float* bhh0, const float* bhh1, const float* gi0, const float* gi1, float* hn0,
                          float* hn1, float* out0, float* out1, int out_stride, float* save0, float* save1,
                          void* stream);
/* One step of the GRU backward recurrence (visual_front.py:20 backward) for both directions in one launch: the
 * recurrent product dh = d_gh(previous step) . W_hh (w_i: vca_pack_weight_bf16 mode 1 of the (3H, H) weight) with
 * the gate arithmetic of vca_gru_cell_bwd2_f32 applied in its epilogue (dh_a is the product itself). */
int vca_gru_bwd_step_bf16(int nsplit, int B, int H, const float* dgh_prev0, const float* dgh_prev1, const void* w0,
                          const void* w1, int dout_stride, const float* dout0, const float* dout1, const float* dhb0,
                          const float* dhb1, const float* save0, const float* save1, const float* hp0,
                          const float* hp1, float* dgi0, float* dgi1, float* dgh0, float* dgh1, float* dhd0,
                          float* dhd1, void* stream);

/* Latency-oriented GEMM for the recurrent products: y_i[M][N] = x_i[M][K] . W_i^T (+ bias_i) for nprob = 1 or 2
 * independent problems of the same shape in one launch (the two directions of a bidirectional layer).
 * w_i: the split-bf16 layout of vca_pack_weight_bf16 for a (N, K) [mode 0] or (K, N) [mode 1] weight with 1 tap. */
int vca_skinny_gemm_bf16(int nsplit, int nprob, int M, int N, int K, const float* x0, const float* x1, const void* w0,
                         const void* w1, const float* b0, const float* b1, float* y0, float* y1, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Losses and optimiser.
 * ---------------------------------------------------------------------------------------------- */
/* out[0] += scale * sum softplus(sign*x)  and its backward (gan_loss, generator.py:363-366). */
int vca_softplus_sum_f32(int n, const float* x, float sign, float scale, float* out, void* stream);
int vca_softplus_bwd_f32(int n, const float* x, float sign, float scale, const float* gout, float* gx, void* stream);
/* out[0] += scale * sum |a - b|  and d/da (nn.L1Loss, train.py:150,226-229). */
int vca_l1_sum_f32(long long n, const float* a, const float* b, float scale, float* out, void* stream);
int vca_l1_bwd_f32(long long n, const float* a, const float* b, float scale, const float* gout, float* ga,
                   void* stream);
/* Connectionist temporal classification loss on frame logits [B][T][V] (log-softmax over V inside), the auxiliary
 * text loss of the LRS3 multi-task configuration.  The reference ships the ASR scorer only for inference
 * (ASR_model/GRID/test.py:160-193 decodes its logits greedily; alphabet and blank '_' = 0 from
 * ASR_model/GRID/src/data/vid_aud_GRID_test.py:20-33), so the loss follows torch.nn.functional.ctc_loss
 * (reduction 'none'), which the tests compare it with.  targets [B][Lmax] int32 class ids, in_len / tgt_len [B] int32.
 * fwd: nll[b] = -log p(targets_b | logits_b); alpha_ws receives B*T*(2*Lmax+1) doubles (log alpha) for the backward
 * call.  zero_infinity != 0: an impossible alignment gives loss 0 and gradient 0 instead of inf / NaN.
 * bwd: glogits[b][t][v] = gnll[b] * d nll[b] / d logits[b][t][v], zero for t >= in_len[b].
 * Limits: T <= 2048, V <= 64, Lmax <= 511. */
int vca_ctc_loss_fwd_f32(int B, int T, int V, int Lmax, const float* logits, const int* targets, const int* in_len,
                         const int* tgt_len, int blank, int zero_infinity, double* alpha_ws, float* nll, void* stream);
int vca_ctc_loss_bwd_f32(int B, int T, int V, int Lmax, const float* logits, const int* targets, const int* in_len,
                         const int* tgt_len, int blank, int zero_infinity, const double* alpha_ws, const float* nll,
                         const float* gnll, float* glogits, void* stream);

/* All loss heads of one phase of the step in one launch (train.py:200-205 discriminator terms incl. the R1 squared
 * norms of :188-194, :222-231 generator terms).  Head e computes v_e = scale * sum_i f(x_i[, y_i]) with
 * f = softplus(sign*x) [kind 0, generator.py:363-366], |x - y| [1, nn.L1Loss], x^2 [2], x [3] and adds weight[k] * v_e
 * into out[slot[k]] for each slot[k] >= 0 (slots < VCA_MAX_LOSS_SLOTS); `out` must be zero on entry.  The head table
 * is host memory and is copied into the launch (nothing has to stay alive; a captured launch keeps its copy).  Head e
 * owns the workgroups [block_begin_e, block_begin_e + ceil(n_e / vca_loss_head_chunk())); total_blocks is their sum.
 * The backward launch writes gx_e[i] = (sum_k gout[slot[k]] * weight[k]) * scale * f'(x_i) for every head with
 * gx != NULL. */
#define VCA_MAX_LOSS_HEADS 24
#define VCA_MAX_LOSS_SLOTS 16
typedef struct vca_loss_head {
  const float* x;
  const float* y;
  float* gx;
  long long n;
  int kind;
  int block_begin;
  float sign;
  float scale;
  int slot[3];
  float weight[3];
} vca_loss_head;
typedef struct vca_loss_heads {
  int nhead;
  int total_blocks;
  vca_loss_head h[VCA_MAX_LOSS_HEADS];
} vca_loss_heads;
int vca_loss_head_chunk(void);
int vca_loss_heads_fwd_f32(const vca_loss_heads* heads, float* out, void* stream);
int vca_loss_heads_bwd_f32(const vca_loss_heads* heads, const float* gout, void* stream);
/* ------------------------------------------------------------------------------------------------
 * Data movement on channels-last maps.
 * ---------------------------------------------------------------------------------------------- */
/* torch.cat along channels with broadcasting sources (generator.py:249-250 noise | phonemes tiled over frequency,
 * :257,262 attention merges, :312-314 conditioning vector tiled over the map): y[n][h][w] = [x1[row1] | x2[row2]],
 * row_k = n*skn + h*skh + w*skw in rows of Ck floats (stride 0 broadcasts).  The backward call extracts channels
 * [c0, c0+Cs) of gy into gx with the same row mapping, adding atomically when `accumulate` (broadcast source,
 * gx zeroed by the caller). */
int vca_concat2_f32(int N, int H, int W, int C1, int C2, const float* x1, long long s1n, long long s1h, long long s1w,
                    const float* x2, long long s2n, long long s2h, long long s2w, float* y, void* stream);
int vca_concat2_bwd_f32(int N, int H, int W, int C, int c0, int Cs, const float* gy, long long sn, long long sh,
                        long long sw, int accumulate, float* gx, void* stream);
/* Decoder input without the tiled copy (generator.py:249-250 repeats the phoneme features over the 20 frequency rows
 * and concatenates noise; decode[0] then runs BN, a 5x5 and a 1x1 convolution over it, generator.py:94-131).  For the
 * repeated channels a KxK convolution is K 1xK convolutions of the single row (kernel row kh stacked in the output
 * channels, p: (B, W, K*C)); output row h sums the kernel rows that stay inside the map:
 *   expand: y[b][h][w][c] = base[b][h][w][c] (optional) + sum_{0 <= h+kh-pad < H} p[b][w][kh*C + c]
 *   reduce: gp[b][w][kh*C + c] = sum_{0 <= h+kh-pad < H} gy[b][h][w][c]   (the adjoint, for the backward pass) */
int vca_hband_expand_f32(int B, int H, int W, int C, int K, int pad, const float* p, const float* base, float* y,
                         void* stream);
int vca_hband_reduce_f32(int B, int H, int W, int C, int K, int pad, const float* gy, float* gp, void* stream);
/* dst[c] += sum_p x[p][c]: bias gradient of nn.Conv*d / nn.Linear from the output gradient, accumulated
 * into the parameter's gradient (what autograd's sum + AccumulateGrad do for the generator.py / visual_front.py
 * biases).  partial is scratch of vca_colsum_groups(P, C) * C floats (need not be initialised). */
int vca_colsum_groups(int P, int C);
int vca_colsum_add_f32(int P, int C, const float* x, float* partial, float* dst, void* stream);
/* vca_colsum_add_f32 for callers that cannot allocate device memory or bring a workspace (code under HIP graph
 * capture, serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The partial sums live
 * in a fixed 4 MiB device array inside the library (one copy per GPU, held until the process ends), so the call
 * allocates nothing; sizes with vca_colsum_groups(P, C) * C > vca_colsum_noalloc_max_floats() are refused with
 * VCA_E_SCRATCH_LIMIT.  All calls on one GPU share that array: calls that overlap on one GPU (different streams
 * without ordering) race; calls on one stream are safe back to back.  vca_noalloc_init() resolves the array's address
 * for the CURRENT device; call it once per device at set-up (the entry point does it on its first call on a device
 * otherwise, which must then not happen under stream capture). */
#define VCA_E_SCRATCH_LIMIT (-27)
int vca_noalloc_init(void);
long long vca_colsum_noalloc_max_floats(void);
int vca_colsum_add_noalloc_f32(int P, int C, const float* x, float* dst, void* stream);
/* y[n][c] = mean of x[n][0:wh][0:ww][c] (nn.AvgPool2d(4) with a 1x1 output, resnet.py:82; .mean([2,3]) of the
 * discriminator heads, generator.py:310,316) and its adjoint. */
int vca_window_mean_f32(int N, int H, int W, int C, int wh, int ww, const float* x, float* y, void* stream);
int vca_window_mean_bwd_f32(int N, int H, int W, int C, int wh, int ww, const float* gy, float* gx, void* stream);
/* nn.Dropout / nn.GRU(dropout=) in training mode (visual_front.py:18,20,29): counter-based generator keyed by
 * (seed, *step_dev, element index); the byte mask is kept for the backward call. */
int vca_dropout_f32(long long n, const float* x, float p, unsigned long long seed, const int* step_dev, float* y,
                    unsigned char* mask, void* stream);
int vca_dropout_bwd_f32(long long n, const float* gy, const unsigned char* mask, float p, float* gx, void* stream);
/* n standard-normal draws from the same counter-based generator (the decoder noise of generator.py:248, drawn on the
 * device instead of on the CPU generator + host-to-device copy); step_dev as in vca_dropout_f32. */
int vca_randn_f32(long long n, unsigned long long seed, const int* step_dev, float* y, void* stream);
/* F.interpolate(mel, scale_factor=1/factor, mode='bilinear') for factor 2 or 4 on [N][H][W] maps (train.py:170-171). */
int vca_mel_down_f32(int N, int H, int W, int factor, const float* x, float* y, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Waveform <-> spectrogram path of validation (src/data/stft.py, audio_processing.py, vid_aud_grid.py:190-232).
 * The windowed-DFT contractions themselves are vca_conv_fwd_f32 calls (see audio.py); these are the kernels around
 * them.  spec / rect rows are [real F | imaginary F] per frame; mag / phase are (B, F, T) like the reference.
 * ---------------------------------------------------------------------------------------------- */
/* F.pad(x, (pad, pad), mode='reflect') into rows of pitch Lrow (zero-filled tail), stft.py:75-79 */
int vca_reflect_pad_f32(int B, int L, int pad, int Lrow, const float* x, float* y, void* stream);
/* magnitude = sqrt(re^2 + im^2), phase = atan2(im, re) (phase may be NULL), stft.py:88-94 */
int vca_mag_phase_f32(int B, int T, int F, const float* spec, float* mag, float* phase, void* stream);
/* [mag cos(phase) | mag sin(phase)], stft.py:99-100 */
int vca_polar_to_rect_f32(int B, int T, int F, const float* mag, const float* phase, float* rect, void* stream);
/* one Griffin-Lim projection: the phase of `spec` with the magnitude `mag`, audio_processing.py:62-64 */
int vca_gl_project_f32(int B, int T, int F, const float* spec, const float* mag, float* rect, void* stream);
/* y = ola[pad : pad+Lout] / window_sum (where > tiny) * scale, stft.py:108-127 */
int vca_istft_finish_f32(int B, int Lrow, int Lout, int pad, float scale, float tiny, const float* ola,
                         const float* wsum, float* y, void* stream);
/* y[n] = x[n] + coef*y[n-1], clipped to [-1, 1]: lfilter([1], [1, -coef]) + np.clip, vid_aud_grid.py:203-210,230-232 */
int vca_deemphasis_clip_f32(int B, int L, float coef, const float* x, float* y, void* stream);

/* Dataset video transform, vid_aud_grid.py:94-121 (build_tensor), on decoded RGB frames [B][Tin][H0][W0][3]:
 * crop at (crop_l, crop_u), Pillow-exact antialiased bilinear resize to OH x OW with the fixed-point coefficient
 * tables (xmin/xcnt/xk per output column, ymin/ycnt/yk per output row, K taps each, 22 fractional bits), optional
 * horizontal flip per clip, RGB -> L, /255, (x - mean)/std, erase box of erase_size at erase[b] = (x_s, y_s), frames
 * t >= len[b] zero.  out is the (B, 1, Tout, OH, OW) volume Visual_front.forward takes. */
int vca_build_video_f32(int B, int Tin, int Tout, int H0, int W0, int OH, int OW, int K, int crop_l, int crop_u,
                        float mean, float std, const unsigned char* frames, const int* xmin, const int* xcnt,
                        const int* xk, const int* ymin, const int* ycnt, const int* yk, const int* start,
                        const int* len, const unsigned char* flip, const int* erase, int erase_size, float* out,
                        void* stream);

/* Loss heads of sync_Discriminator.forward (generator.py:347-361) on the embeddings v, a [B][T][C] (T <= 160).
 * nce (gen=False, :352-361): loss[b] = -1/2 mean_i (log_softmax(S,2)_ii + log_softmax(S,1)_ii) with
 * S = normalize(v) normalize(a)^T / temp; S [B][T][T] and stats [B][4][T] (1/|v_i|, 1/|a_i|, row and column
 * log-sum-exp) are kept for the backward call, which writes dv and/or da (either may be NULL) for the incoming
 * gradient gloss[B].  cos (gen=True, :348-350): loss[b] = 5 - mean_t |cos(v_t, a_t)|, cosv [B][T] kept. */
int vca_sync_nce_fwd_f32(int B, int T, int C, const float* v, const float* a, float temp, float* S, float* stats,
                         float* loss, void* stream);
int vca_sync_nce_bwd_f32(int B, int T, int C, const float* v, const float* a, const float* S, const float* stats,
                         const float* gloss, float temp, float* dv, float* da, void* stream);
int vca_sync_cos_fwd_f32(int B, int T, int C, const float* v, const float* a, float* cosv, float* loss, void* stream);
int vca_sync_cos_bwd_f32(int B, int T, int C, const float* v, const float* a, const float* cosv, const float* gloss,
                         float* dv, float* da, void* stream);

/* torch.optim.Adam step (coupled L2 weight decay; amsgrad when vmax != NULL) over a flat buffer, n % 4 == 0
 * (train.py:82-83,211,237). `step` is the 1-based step count; when step_dev != NULL the count is read from that
 * device word instead (so the launch can be replayed from a captured HIP graph), and when lr_dev != NULL the learning
 * rate is read from that device float instead of `lr` (the MultiStepLR schedules of train.py:85-86 keep working under
 * replay: the host rewrites the word between replays). */
int vca_adam_f32(long long n, float* p, const float* g, float* m, float* v, float* vmax, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int step, const int* step_dev, const float* lr_dev,
                 void* stream);

/* Dynamic loss scaling for half-precision gradients (no counterpart in the reference, which trains in fp32; the
 * semantics are those of torch.cuda.amp.GradScaler: skip the optimizer step and halve the scale on overflow, double it
 * after `interval` clean steps).  All state is in device memory so that a captured step keeps adapting:
 * scale = float[2] {scale, 1/scale}; state = int[4] {clean steps in a row, overflow flag D side, overflow flag G side,
 * skipped steps}.
 *   vca_nonfinite_f32      sets *flag to 1 if any of the n (n % 4 == 0, 16-byte aligned) gradients is Inf / NaN
 *   vca_loss_scale_count   ++*step_dev unless *flag (the Adam step count of the side, before its launches)
 *   vca_adam_scaled_f32    vca_adam_f32 on g / scale; a no-op when *flag is up (train.py:211,237 with a scaler)
 *   vca_loss_scale_update  once per training step, after both optimizers: adapts the scale, clears the flags */
int vca_nonfinite_f32(long long n, const float* g, int* flag, void* stream);
int vca_loss_scale_count(int* step_dev, const int* flag, void* stream);
int vca_adam_scaled_f32(long long n, float* p, const float* g, float* m, float* v, float* vmax, float beta1,
                        float beta2, float eps, float weight_decay, const int* step_dev, const float* lr_dev,
                        const float* scale, const int* flag, void* stream);
int vca_loss_scale_update(float* scale, int* state, float growth, float backoff, int interval, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VCA_HIP_H */
