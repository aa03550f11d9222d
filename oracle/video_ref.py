"""CPU restatement (numpy) of the reference's video input pipeline, vid_aud_grid.py:94-121 (build_tensor):
ToPILImage -> Crop([59, 95, 195, 231]) -> Resize([112, 112]) -> [horizontal flip] -> Grayscale -> ToTensor ->
Normalize(0.4136, 0.17), stacked into a zero-initialised (max_T, 1, 112, 112) volume, then random erasing of a
56 x 56 box (set to 0 in normalised space) and a transpose to (1, T, 112, 112).

TEST INFRASTRUCTURE ONLY.  torchvision is not installed, so the reference's own transform objects cannot be run:
the two steps with non-trivial integer arithmetic - Pillow's antialiased bilinear resize on uint8 RGB and its
RGB -> L conversion - are restated from Pillow's published algorithm (Resample.c: precompute_coeffs,
normalize_coeffs_8bpc, the two 8-bit passes with 22 fractional bits; Convert.c: L24) and pinned bit-exactly
against the installed Pillow in tests/test_video_oracle_cpu.py.
"""
import numpy as np

CROP = (59, 95, 195, 231)          # left, upper, right, lower (vid_aud_grid.py:99)
OUT = 112
MEAN, STD = 0.4136, 0.17
PRECISION_BITS = 32 - 8 - 2


def resize_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter over the whole axis.
    Returns (xmin[out], count[out], k[out][ksize] int32)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = np.zeros(n, np.float64)
        for x in range(n):
            t = abs((x + lo - center + 0.5) * ss)
            w[x] = 1.0 - t if t < 1.0 else 0.0
        tot = w.sum()
        if tot != 0.0:
            w = w / tot
        for x in range(n):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        xmin[xx], cnt[xx] = lo, n
    return xmin, cnt, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, out_h=OUT, out_w=OUT):
    """img (H, W, C) uint8 -> (out_h, out_w, C) uint8: horizontal pass to uint8, then vertical pass."""
    H, W, C = img.shape
    xmin, xcnt, xk = resize_coeffs(W, out_w)
    tmp = np.zeros((H, out_w, C), np.uint8)
    for xx in range(out_w):
        acc = np.full((H, C), 1 << (PRECISION_BITS - 1), np.int64)
        for j in range(xcnt[xx]):
            acc += img[:, xmin[xx] + j, :].astype(np.int64) * int(xk[xx, j])
        tmp[:, xx, :] = _clip8(acc)
    ymin, ycnt, yk = resize_coeffs(H, out_h)
    out = np.zeros((out_h, out_w, C), np.uint8)
    for yy in range(out_h):
        acc = np.full((out_w, C), 1 << (PRECISION_BITS - 1), np.int64)
        for j in range(ycnt[yy]):
            acc += tmp[ymin[yy] + j].astype(np.int64) * int(yk[yy, j])
        out[yy] = _clip8(acc)
    return out


def rgb_to_l(img):
    """Pillow convert('L'): (R*19595 + G*38470 + B*7471 + 0x8000) >> 16."""
    i = img.astype(np.int64)
    return ((i[..., 0] * 19595 + i[..., 1] * 38470 + i[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def build_tensor(frames, max_t, flip=False, erase=None):
    """frames (T, H0, W0, 3) uint8 -> (1, max_t, 112, 112) float32, vid_aud_grid.py:94-121.  erase = (x_s, y_s) or None."""
    vol = np.zeros((max_t, 1, OUT, OUT), np.float32)
    l, u, r, lo = CROP
    for i, f in enumerate(frames):
        img = resize_bilinear_u8(f[u:lo, l:r])
        if flip:
            img = img[:, ::-1]
        g = rgb_to_l(img).astype(np.float32) / np.float32(255.0)
        vol[i, 0] = (g - np.float32(MEAN)) / np.float32(STD)
    if erase is not None:
        x_s, y_s = erase
        vol[:, :, max(0, y_s):min(OUT, y_s + 56), max(0, x_s):min(OUT, x_s + 56)] = 0.0
    return vol.transpose(1, 0, 2, 3)
