// Video input pipeline of the dataset (vid_aud_grid.py:94-121, build_tensor) as one kernel over decoded RGB frames
// resident on the device: crop -> Pillow's antialiased bilinear resize on uint8 (two 8-bit passes, 22 fractional
// bits, horizontal then vertical, each rounded to uint8) -> optional horizontal flip -> RGB -> L (ITU-R 601-2 in
// 16.16 fixed point) -> /255 -> Normalize(mean, std) -> random erasing box -> zero frames behind the clip's length,
// written straight into the (B, 1, T, OH, OW) float volume the visual front-end reads.  Integer stages are
// bit-identical to Pillow's.
#include "common.h"

namespace {

constexpr int THR = 256;
constexpr int PREC = 22;   // Pillow's PRECISION_BITS for 8-bit channels

__device__ __forceinline__ int clip8(int v) {
  v >>= PREC;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

struct VidArgs {
  int B, Tin, Tout, H0, W0, OH, OW, K;
  int crop_l, crop_u;
  float mean, stdv;
  const unsigned char* frames;   // [B][Tin][H0][W0][3]
  const int* xmin; const int* xcnt; const int* xk;   // [OW], [OW], [OW][K]
  const int* ymin; const int* ycnt; const int* yk;   // [OH], [OH], [OH][K]
  const int* start;   // [B] first frame of the window (may be NULL = 0)
  const int* len;     // [B] frames of the clip from `start` on
  const unsigned char* flip;   // [B] (may be NULL)
  const int* erase;   // [B][2] = x_s, y_s of the 56x56 box (may be NULL); a box fully outside erases nothing
  int erase_size;
  float* out;         // [B][1][Tout][OH][OW]
};

__global__ __launch_bounds__(THR) void build_video_kernel(VidArgs a) {
  const size_t total = (size_t)a.B * a.Tout * a.OH * a.OW;
  for (size_t i = (size_t)blockIdx.x * THR + threadIdx.x; i < total; i += (size_t)gridDim.x * THR) {
    const int ox = (int)(i % a.OW);
    size_t r = i / a.OW;
    const int oy = (int)(r % a.OH); r /= a.OH;
    const int t = (int)(r % a.Tout);
    const int b = (int)(r / a.Tout);
    float v = 0.f;
    bool live = t < a.len[b];
    if (live && a.erase) {
      const int xs = a.erase[2 * b], ys = a.erase[2 * b + 1];
      if (ox >= xs && ox < xs + a.erase_size && oy >= ys && oy < ys + a.erase_size) live = false;
    }
    if (live) {
      const int sx = (a.flip && a.flip[b]) ? a.OW - 1 - ox : ox;
      const int ft = (a.start ? a.start[b] : 0) + t;
      const unsigned char* img = a.frames + (((size_t)b * a.Tin + ft) * a.H0) * a.W0 * 3;
      const int x0 = a.crop_l + a.xmin[sx], nx = a.xcnt[sx];
      const int y0 = a.crop_u + a.ymin[oy], ny = a.ycnt[oy];
      int acc[3] = {1 << (PREC - 1), 1 << (PREC - 1), 1 << (PREC - 1)};
      for (int j = 0; j < ny; ++j) {
        const unsigned char* row = img + ((size_t)(y0 + j) * a.W0 + x0) * 3;
        int h[3] = {1 << (PREC - 1), 1 << (PREC - 1), 1 << (PREC - 1)};
        for (int k = 0; k < nx; ++k) {
          const int w = a.xk[sx * a.K + k];
          h[0] += row[3 * k] * w;
          h[1] += row[3 * k + 1] * w;
          h[2] += row[3 * k + 2] * w;
        }
        const int wy = a.yk[oy * a.K + j];
        acc[0] += clip8(h[0]) * wy;
        acc[1] += clip8(h[1]) * wy;
        acc[2] += clip8(h[2]) * wy;
      }
      const int R = clip8(acc[0]), G = clip8(acc[1]), Bc = clip8(acc[2]);
      const int L = (R * 19595 + G * 38470 + Bc * 7471 + 0x8000) >> 16;
      v = ((float)L / 255.0f - a.mean) / a.stdv;
    }
    a.out[i] = v;
  }
}

}  // namespace

extern "C" int vca_build_video_f32(int B, int Tin, int Tout, int H0, int W0, int OH, int OW, int K, int crop_l,
                                   int crop_u, float mean, float std, const unsigned char* frames, const int* xmin,
                                   const int* xcnt, const int* xk, const int* ymin, const int* ycnt, const int* yk,
                                   const int* start, const int* len, const unsigned char* flip, const int* erase,
                                   int erase_size, float* out, void* stream) {
  VCA_CHECK_ARG(B > 0 && Tin > 0 && Tout > 0 && H0 > 0 && W0 > 0 && OH > 0 && OW > 0 && K > 0 && crop_l >= 0 &&
                crop_u >= 0 && std > 0.f && frames && xmin && xcnt && xk && ymin && ycnt && yk && len && out);
  VidArgs a{B, Tin, Tout, H0, W0, OH, OW, K, crop_l, crop_u, mean, std, frames, xmin, xcnt, xk,
            ymin, ycnt, yk, start, len, flip, erase, erase_size, out};
  size_t g = ((size_t)B * Tout * OH * OW + THR - 1) / THR;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(build_video_kernel, dim3((unsigned)g), dim3(THR), 0, (hipStream_t)stream, a);
  return vca_launch_status();
}
