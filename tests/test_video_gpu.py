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


@gpu
def test_prefetcher_carries_the_window_of_the_audio_targets():
    """A source that also yields the mel / spectrogram targets (4 frames per video frame, vid_aud_grid.py:160): the
    prefetcher cuts them with each clip's drawn window (extract_window, vid_aud_grid.py:172-188), stages them through
    pinned buffers and copies them on its side stream; every batch arrives with the slices the host would have cut,
    while the consumer keeps its stream busy and reads the tensors late."""
    import random
    from vcagan_amd import video
    dev = torch.device('cuda', 0)
    vt = video.VideoTransform(dev)
    B, Tin, W = 2, 9, 5
    g = torch.Generator().manual_seed(4)
    batches = []
    for i in range(6):
        mel = torch.randn(B, 1, 80, 4 * Tin, generator=g)
        spec = torch.rand(B, 1, 321, 4 * Tin, generator=g)
        batches.append((_frames(B, Tin, seed=30 + i), [9, 8], dict(mel=(mel, 4), spec=(spec, 4))))
    random.seed(5)
    pf = video.VideoPrefetcher(iter(batches), vt, max_t=W, window=W, augment=False, depth=2)
    busy = torch.randn(2048, 2048, device=dev)
    got = []
    for out, lens, meta in pf:
        busy = busy @ busy * 1e-3
        mel_d, spec_d = meta['extra']['mel'], meta['extra']['spec']
        assert mel_d.is_cuda and mel_d.shape == (B, 1, 80, 4 * W) and spec_d.shape == (B, 1, 321, 4 * W)
        got.append((meta['start'], (mel_d * 1.0), (spec_d * 1.0)))     # read on the consumer's stream, behind `busy`
    assert len(got) == len(batches)
    for (fr, n, ex), (start, mel_d, spec_d) in zip(batches, got):
        for b in range(B):
            assert torch.equal(mel_d[b].cpu(), ex['mel'][0][b, ..., 4 * start[b]:4 * (start[b] + W)])
            assert torch.equal(spec_d[b].cpu(), ex['spec'][0][b, ..., 4 * start[b]:4 * (start[b] + W)])


@gpu
def test_input_prefetcher_hands_out_each_batch_once_and_intact():
    """InputPrefetcher (host-to-device copies one batch ahead on a side stream, two alternating device buffer sets):
    every batch arrives intact and in order although the consumer's stream is kept busy and reads its tensors late."""
    from vcagan_amd import video
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(9)
    host = [dict(vid=torch.randn(2, 1, 6, 32, 32, generator=g).pin_memory(),
                 mel=torch.randn(2, 1, 80, 24, generator=g).pin_memory(), vid_len=[6, 5 - (i % 2)]) for i in range(7)]
    busy = torch.randn(2048, 2048, device=dev)
    seen = []
    for batch in video.InputPrefetcher(iter(host), dev):
        busy = busy @ busy * 1e-3
        seen.append((batch['vid'] + 0.0, batch['mel'] + 0.0, batch['vid_len']))
    assert len(seen) == len(host)
    for h, (v, m, n) in zip(host, seen):
        assert torch.equal(v.cpu(), h['vid']) and torch.equal(m.cpu(), h['mel']) and n == h['vid_len']
