"""GPU waveform <-> spectrogram path (vcagan_amd.audio: mirrors of the reference's STFT, TacotronSTFT, griffin_lim,
inverse_mel / inverse_spec) against the numpy restatement oracle/audio_ref.py.  fp32; tolerances in the asserts."""
import numpy as np
import pytest
import torch

from oracle import audio_ref as A

gpu = pytest.mark.gpu


def _sig(b, n, seed=0):
    rng = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    x = 0.4 * np.sin(2 * np.pi * (200 + 37 * np.arange(b))[:, None] * t[None]) + 0.1 * rng.randn(b, n)
    return np.clip(x, -1, 1).astype(np.float32)


@gpu
@pytest.mark.parametrize('b,n', [(2, 4800), (3, 48000), (1, 5000)])
def test_stft_transform_and_inverse(b, n):
    from vcagan_amd import audio
    x = _sig(b, n)
    st = audio.STFT(640, 160, 640).cuda()
    assert tuple(st.forward_basis.shape) == (642, 1, 640) and tuple(st.inverse_basis.shape) == (642, 1, 640)
    fb, ib = A.bases()
    assert np.abs(st.forward_basis.cpu().numpy()[:, 0] - fb).max() < 1e-6
    assert np.abs(st.inverse_basis.cpu().numpy()[:, 0] - ib).max() < 1e-6
    mag, ph = st.transform(torch.from_numpy(x).cuda())
    rm, rp = A.transform(x)
    assert mag.shape == rm.shape
    assert np.abs(mag.cpu().numpy() - rm).max() < 2e-4 * max(1.0, rm.max())
    big = rm > 1e-2 * rm.max()
    dphi = np.angle(np.exp(1j * (ph.cpu().numpy() - rp)))
    assert np.abs(dphi[big]).max() < 2e-3
    y = st.inverse(mag, ph)
    ry = A.inverse(rm, rp)
    assert tuple(y.shape) == (b, 1, ry.shape[1])
    assert np.abs(y.cpu().numpy()[:, 0] - ry).max() < 1e-4
    if n % 160 == 0:
        assert np.abs(y.cpu().numpy()[:, 0] - x)[:, 320:-320].max() < 1e-4       # reconstruction identity


@gpu
def test_griffin_lim_matches_oracle_and_converges():
    from vcagan_amd import audio
    x = _sig(2, 4800, 1)
    st = audio.STFT(640, 160, 640).cuda()
    rm, _ = A.transform(x)
    ang0 = np.random.RandomState(2).uniform(-np.pi, np.pi, rm.shape).astype(np.float32)
    mag = torch.from_numpy(rm).cuda()
    y3 = audio.griffin_lim(mag, st, 3, torch.from_numpy(ang0)).cpu().numpy()[:, 0]
    r3 = A.griffin_lim(rm, ang0, 3)
    assert np.abs(y3 - r3).max() < 2e-3 * max(1.0, np.abs(r3).max())
    # spectral convergence: the magnitude error of the reconstruction falls with the iterations
    e = []
    for it in (0, 30):
        y = audio.griffin_lim(mag, st, it, torch.from_numpy(ang0)).squeeze(1)
        e.append((st.transform(y)[0] - mag).abs().mean().item())
    assert e[1] < 0.6 * e[0], e


@gpu
def test_mel_spectrogram_inverse_mel_and_deemphasis():
    from vcagan_amd import audio
    x = _sig(2, 4800, 3) * 0.9
    ts = audio.TacotronSTFT(640, 160, 640, 80, 16000, 55.0, 7500.0).cuda()
    mb = A.mel_filterbank()
    assert np.abs(ts.mel_basis.cpu().numpy() - mb).max() < 1e-6
    mel, mag = ts.mel_spectrogram(torch.from_numpy(x).cuda())
    rmel, rmag = A.mel_spectrogram(x, mb)
    assert np.abs(mel.cpu().numpy() - rmel).max() < 2e-3
    # inverse_mel: de-normalise, exp, mel -> linear, x1000, Griffin-Lim, de-emphasis + clip (vid_aud_grid.py:190-210)
    log1e5 = np.log(1e-5)
    norm = (((rmel - log1e5) / (-log1e5 / 2)) - 1).astype(np.float32)
    ang0 = np.random.RandomState(4).uniform(-np.pi, np.pi, rmag.shape).astype(np.float32)
    wav = audio.inverse_mel(torch.from_numpy(norm[:, None]).cuda(), ts, 2, torch.from_numpy(ang0)).cpu().numpy()
    spec = np.einsum('btm,mf->bft', np.exp(rmel).transpose(0, 2, 1), mb) * 1000
    ref = A.deemphasize(A.griffin_lim(spec.astype(np.float32), ang0, 2))
    assert wav.shape == ref.shape and np.abs(wav - ref).max() < 5e-3
    assert np.abs(wav).max() <= 1.0
    # de-emphasis alone, long signal (segment carries across all 64 lanes)
    z = (_sig(3, 48000, 5) * 0.02).astype(np.float32)
    out = audio.deemphasize_clip(torch.from_numpy(z).cuda()).cpu().numpy()
    assert np.abs(out - A.deemphasize(z)).max() < 2e-5
