"""GPU video transform (vcagan_amd.video, mirror of vid_aud_grid.py:94-121) vs the numpy restatement
oracle/video_ref.py (itself bit-exact against Pillow): integer stages identical, floats within 1e-6."""
import numpy as np
import pytest
import torch

from oracle import video_ref as V

gpu = pytest.mark.gpu


def _frames(b, t, h=288, w=360, seed=0):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, (b, t, h // 8, w // 8, 3)).astype(np.uint8)
    f = np.repeat(np.repeat(base, 8, 2), 8, 3)
    return np.clip(f.astype(np.int64) + rng.randint(-20, 21, f.shape), 0, 255).astype(np.uint8)


@gpu
def test_build_tensor_matches_oracle():
    from vcagan_amd import video
    B, Tin, max_t = 3, 6, 5
    fr = _frames(B, Tin)
    lens, start = [5, 3, 4], [1, 0, 2]
    flip, erase = [False, True, True], [[-10, 60], [30, -5], [66, 66]]
    vt = video.VideoTransform(torch.device('cuda', 0))
    out = vt(torch.from_numpy(fr).cuda(), lens, max_t, flip, erase, start).cpu().numpy()
    assert out.shape == (B, 1, max_t, 112, 112)
    for b in range(B):
        ref = V.build_tensor(fr[b, start[b]:start[b] + lens[b]], max_t, flip[b], tuple(erase[b]))
        assert np.abs(out[b] - ref).max() < 1e-6, (b, np.abs(out[b] - ref).max())
    # no augmentation, no window
    out2 = vt(torch.from_numpy(fr).cuda(), [6, 6, 6], 8).cpu().numpy()
    ref2 = V.build_tensor(fr[1], 8)
    assert np.abs(out2[1] - ref2).max() < 1e-6 and np.all(out2[:, :, 6:] == 0)
    # the grey levels behind the floats are Pillow's integers
    lvl = np.rint((out2[1, 0, 0] * V.STD + V.MEAN) * 255)
    assert np.array_equal(lvl, np.rint((ref2[0, 0] * V.STD + V.MEAN) * 255))


@gpu
def test_prefetcher_matches_synchronous_path():
    """The double-buffered pinned-copy + side-stream pipeline hands out exactly the tensors the synchronous call
    produces for the same draws, in order, with device-side window sampling, while the consumer keeps its stream busy."""
    import random
    from vcagan_amd import video
    dev = torch.device('cuda', 0)
    vt = video.VideoTransform(dev)
    B, Tin, W = 3, 9, 5
    batches = [(_frames(B, Tin, seed=10 + i), [9, 7 + (i % 2), 6]) for i in range(5)]
    random.seed(123)
    pf = video.VideoPrefetcher(iter(batches), vt, max_t=W, window=W, augment=True, depth=2)
    busy = torch.randn(2048, 2048, device=dev)
    got = []
    for out, lens, meta in pf:
        busy = busy @ busy * 1e-3                     # work on the consumer's stream between batches
        got.append((out.clone(), lens, meta))
    assert len(got) == len(batches)
    for (fr, n), (out, lens, meta) in zip(batches, got):
        assert lens == [W] * B and all(0 <= s <= k - W for s, k in zip(meta['start'], n))
        ref = vt(torch.from_numpy(fr).to(dev), lens, W, meta['flip'], meta['erase'], meta['start'])
        assert torch.equal(out, ref)
    # whole clips (validation path): lengths pass through, frames behind them are zero
    pf2 = video.VideoPrefetcher(iter(batches[:2]), vt, max_t=Tin, window=None, augment=False)
    out, lens, meta = next(pf2)
    assert lens == [9, 7, 6] and meta['start'] is None
    assert torch.equal(out, vt(torch.from_numpy(batches[0][0]).to(dev), lens, Tin))
    assert float(out[1, :, 7:].abs().max()) == 0.0
