"""GPU side of the dataset's video transform (vid_aud_grid.py:94-121 build_tensor, :172-188 extract_window): decoded
RGB frames go to the device as uint8 once and one kernel produces the normalised (B, 1, T, 112, 112) volume, with
the crop, Pillow-exact resize, flip, grayscale, normalisation, random erasing and zero padding fused.  The random
draws (flip, erase position, window start) stay on the host, like the reference's `random` calls."""
import random

import numpy as np
import torch

from . import _lib
from ._lib import check, stream

CROP = (59, 95, 195, 231)      # left, upper, right, lower (vid_aud_grid.py:99)
MEAN, STD = 0.4136, 0.1700     # vid_aud_grid.py:108
_PREC = 22


def _coeffs(in_size, out_size):
    """Pillow's bilinear resampling coefficients (Resample.c precompute_coeffs + normalize_coeffs_8bpc)."""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = fs
    K = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, K), np.int32)
    for xx in range(out_size):
        c = (xx + 0.5) * scale
        lo = max(int(c - support + 0.5), 0)
        hi = min(int(c + support + 0.5), in_size)
        w = np.array([max(0.0, 1.0 - abs((x + lo - c + 0.5) / fs)) for x in range(hi - lo)], np.float64)
        if w.sum() != 0.0:
            w = w / w.sum()
        kk[xx, :hi - lo] = [int(0.5 + v * (1 << _PREC)) for v in w]
        xmin[xx], cnt[xx] = lo, hi - lo
    return xmin, cnt, kk


class VideoTransform:
    """build_tensor for a batch: frames (B, Tin, H0, W0, 3) uint8 on the device -> (B, 1, max_t, 112, 112) float32."""

    def __init__(self, device, crop=CROP, out_size=112, mean=MEAN, std=STD, erase_size=56):
        self.crop, self.out, self.mean, self.std, self.erase_size = crop, out_size, mean, std, erase_size
        l, u, r, lo = crop
        xs = _coeffs(r - l, out_size)
        ys = _coeffs(lo - u, out_size)
        self.K = max(xs[2].shape[1], ys[2].shape[1])

        def dev(t):
            return torch.from_numpy(np.ascontiguousarray(t)).to(device)

        def padk(k):
            o = np.zeros((k.shape[0], self.K), np.int32)
            o[:, :k.shape[1]] = k
            return o
        self.tabs = [dev(xs[0]), dev(xs[1]), dev(padk(xs[2])), dev(ys[0]), dev(ys[1]), dev(padk(ys[2]))]
        self.device = device

    def __call__(self, frames, lens, max_t, flip=None, erase=None, start=None):
        """lens: frames per clip; flip: (B,) bools or None; erase: (B, 2) ints (x_s, y_s) or None; start: (B,) first
        frame of each clip's window (extract_window) or None."""
        assert frames.dtype == torch.uint8 and frames.is_cuda and frames.is_contiguous() and frames.shape[-1] == 3
        B, Tin, H0, W0, _ = frames.shape
        l, u, r, lo = self.crop
        assert r <= W0 and lo <= H0

        def ints(v, shape):
            if v is None:
                return None
            t = torch.as_tensor(np.asarray(v), dtype=torch.int32).reshape(shape).contiguous()
            return t.to(self.device)
        lens_t, start_t, erase_t = ints(lens, (B,)), ints(start, (B,)), ints(erase, (B, 2))
        flip_t = None if flip is None else torch.as_tensor(np.asarray(flip), dtype=torch.uint8).reshape(B).to(self.device)
        hi = np.asarray(lens) + (0 if start is None else np.asarray(start))
        assert int(hi.max()) <= Tin and int(np.asarray(lens).max()) <= max_t
        out = torch.empty((B, 1, max_t, self.out, self.out), device=self.device, dtype=torch.float32)
        p = lambda t: None if t is None else t.data_ptr()
        check(_lib.lib().vca_build_video_f32(B, Tin, max_t, H0, W0, self.out, self.out, self.K, l, u, self.mean,
                                             self.std, frames.data_ptr(), *[t.data_ptr() for t in self.tabs],
                                             p(start_t), p(lens_t), p(flip_t), p(erase_t), self.erase_size,
                                             out.data_ptr(), stream()), 'build_video')
        return out


def draw_augmentations(batch, augment):
    """The reference's host-side draws: StatefulRandomHorizontalFlip(0.5) per clip (vid_aud_grid.py:96) and the erase
    box start x_s, y_s ~ randint(-10, 66) (vid_aud_grid.py:116)."""
    if not augment:
        return None, None
    flip = [random.random() < 0.5 for _ in range(batch)]
    erase = [[random.randint(-10, 66), random.randint(-10, 66)] for _ in range(batch)]
    return flip, erase


def draw_window(num_frames, window_size):
    """extract_window's start frame (vid_aud_grid.py:176) and the matching mel start / length for 25 fps, 16 kHz, hop 160."""
    st = random.randint(0, num_frames - window_size)
    vid_2_aud = 16000 / 25 / 160
    return st, int(st * vid_2_aud), int(window_size * vid_2_aud)


class VideoPrefetcher:
    """Feeds VideoTransform one batch ahead of the training step: the role of DataLoader(pin_memory=True) +
    `vid.cuda()` in the reference loop (train.py:139-146,173), with extract_window's sampling (vid_aud_grid.py:172-188)
    done on the device through the transform kernel's per-clip start offsets.

    `source` yields (frames, n_frames) or (frames, n_frames, extras): decoded RGB clips as one uint8 array
    (B, Tin, H0, W0, 3) on the host, the number of valid frames of each clip, and optionally a dict of further host
    tensors name -> (tensor (B, ..., ratio * Tin), ratio) whose last axis is time at `ratio` steps per video frame (the
    mel and spectrogram targets, ratio 4: vid_aud_grid.py:160,172-188 cuts them with the clip's window).  They are
    cut to the drawn window on the host, staged through pinned buffers of their own and copied on the same side
    stream, so a training step finds all of its inputs on the device; they come back as meta['extra'][name].  Per batch the host draws the window starts and the augmentations, copies
    the frames into one of `depth` pinned staging buffers, and on a side stream issues the host-to-device copy and the
    transform kernel; iteration hands out the finished (B, 1, T, 112, 112) tensor after making the consumer's stream
    wait for that work, and immediately stages the next batch, so copy and transform overlap the step that is
    running.  `window`: frames per training window (train.py:43 window_size), or None for whole clips padded to
    `max_t` (the validation path)."""

    def __init__(self, source, transform, max_t, window=None, augment=False, depth=2):
        assert depth >= 2
        self.src = iter(source)
        self.vt, self.max_t, self.window, self.augment = transform, max_t, window, augment
        self.side = torch.cuda.Stream(device=transform.device)
        self.slots = [dict(pinned=None, dev=None, copied=None, extra={}) for _ in range(depth)]
        self.queue = []          # staged batches in order: (tensor, lens, meta, ready event)
        self.k = 0
        self.done = False
        for _ in range(depth - 1):
            self._stage()

    def _stage(self):
        if self.done:
            return
        try:
            item = next(self.src)
        except StopIteration:
            self.done = True
            return
        frames, n_frames = item[0], item[1]
        extras = item[2] if len(item) > 2 else {}
        frames = torch.as_tensor(frames)
        assert frames.dtype == torch.uint8 and frames.dim() == 5
        B = frames.shape[0]
        if self.window is not None:
            start = [random.randint(0, int(n) - self.window) for n in n_frames]     # extract_window
            lens = [self.window] * B
        else:
            start, lens = None, [int(n) for n in n_frames]
        flip, erase = draw_augmentations(B, self.augment)
        slot = self.slots[self.k % len(self.slots)]
        self.k += 1
        if slot['copied'] is not None:
            slot['copied'].synchronize()             # the copy that last read this pinned buffer has finished
        if slot['pinned'] is None or slot['pinned'].shape != frames.shape:
            slot['pinned'] = torch.empty(frames.shape, dtype=torch.uint8, pin_memory=True)
            slot['dev'] = torch.empty(frames.shape, dtype=torch.uint8, device=self.vt.device)
        slot['pinned'].copy_(frames)
        ex_host = {}
        for name, (t, ratio) in extras.items():
            t = torch.as_tensor(t)
            if start is not None:       # the clip's window on this tensor's time axis
                t = torch.stack([t[b, ..., ratio * start[b]:ratio * (start[b] + self.window)] for b in range(B)])
            buf = slot['extra'].get(name)
            if buf is None or buf[0].shape != t.shape or buf[0].dtype != t.dtype:
                buf = slot['extra'][name] = (torch.empty(t.shape, dtype=t.dtype, pin_memory=True),
                                             torch.empty(t.shape, dtype=t.dtype, device=self.vt.device))
            buf[0].copy_(t)
            ex_host[name] = buf
        # the device staging buffer is only ever touched on the side stream, so stream order protects its reuse and
        # nothing here waits for the consumer's stream (that wait would serialise the copy behind the running step)
        with torch.cuda.stream(self.side):
            slot['dev'].copy_(slot['pinned'], non_blocking=True)
            extra = {}
            if ex_host and getattr(self, 'release', None) is not None:
                self.side.wait_event(self.release)
            for name, (pin, dev) in ex_host.items():
                dev.copy_(pin, non_blocking=True)
                extra[name] = dev
            slot['copied'] = torch.cuda.Event()
            slot['copied'].record(self.side)
            out = self.vt(slot['dev'], lens, self.max_t, flip, erase, start)
            ready = torch.cuda.Event()
            ready.record(self.side)
        # (an extra's device buffer belongs to its slot: it is overwritten `depth` batches later, behind the
        # consumer's wait below, so a consumer that keeps it longer than that clones it)
        self.queue.append((out, lens, dict(flip=flip, erase=erase, start=start, extra=extra), ready))

    def __iter__(self):
        return self

    def __next__(self):
        if not self.queue:
            self._stage()
        if not self.queue:
            raise StopIteration
        out, lens, meta, ready = self.queue.pop(0)
        cur = torch.cuda.current_stream(self.vt.device)
        cur.wait_event(ready)
        out.record_stream(cur)
        # Everything the consumer has enqueued so far belongs to earlier batches.  The staging below reuses the slot of
        # the batch handed out before this one, whose extras live in the slot's own device buffers: the side stream
        # waits for this point of the consumer's stream before it overwrites them.
        self.release = torch.cuda.Event()
        self.release.record(cur)
        self._stage()                                # the next batch's copy + transform overlap the caller's step
        return out, lens, meta


class InputPrefetcher:
    """Host-to-device copies of a training step's inputs one batch ahead, on a side stream: the pin_memory + .cuda()
    hand-over of the reference loop (train.py:139-146,173,182-186) without the copy sitting in front of the step.
    `batches` yields dicts of pinned host tensors (equal shapes from batch to batch).  Iteration returns a dict of
    device tensors valid until the next call of next(); two device buffer sets alternate: while the consumer's step
    reads one, the side stream fills the other, and the side stream waits for the consumer's stream before it
    overwrites a set (an event recorded at the following next())."""

    def __init__(self, batches, device):
        self.src = iter(batches)
        self.device = device
        self.side = torch.cuda.Stream(device=device)
        self.sets = [None, None]
        self.free = [None, None]       # event after which set i may be overwritten
        self.k = 0
        self.staged = None
        self._stage()

    def _stage(self):
        try:
            host = next(self.src)
        except StopIteration:
            self.staged = None
            return
        i = self.k % 2
        self.k += 1
        if self.sets[i] is None:
            self.sets[i] = {n: torch.empty(t.shape, dtype=t.dtype, device=self.device) for n, t in host.items()
                            if torch.is_tensor(t)}
        with torch.cuda.stream(self.side):
            if self.free[i] is not None:
                self.side.wait_event(self.free[i])
            for n, t in host.items():
                if torch.is_tensor(t):
                    self.sets[i][n].copy_(t, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.side)
        self.staged = (i, dict(self.sets[i], **{n: t for n, t in host.items() if not torch.is_tensor(t)}), ready)

    def __iter__(self):
        return self

    def __next__(self):
        if self.staged is None:
            raise StopIteration
        i, out, ready = self.staged
        cur = torch.cuda.current_stream(self.device)
        # whatever the consumer enqueued so far (its reads of the OTHER set) is what the next copy into that set waits for
        ev = torch.cuda.Event()
        ev.record(cur)
        self.free[1 - i] = ev
        cur.wait_event(ready)
        self._stage()
        return out
