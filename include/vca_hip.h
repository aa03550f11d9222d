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
} vca_conv_desc;

/* y = conv(prologue(x), w) + bias (+ res), * out_scale.
 * wpack: [KD*KH*KW][Cin][Cout4] with Cout4 = Cout rounded up to 4 (see vca_pack_weight_f32).
 * bias/in_scale/in_shift/in_alpha/res/stats may be NULL.  stats (2*Cout floats, zeroed by the
 * caller) receives per-output-channel sum and sum of squares of (y - bias) for a following
 * train-mode BatchNorm. */
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

#ifdef __cplusplus
}
#endif
#endif /* VCA_HIP_H */
