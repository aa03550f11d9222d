"""The numpy restatement of the reference's STFT / Griffin-Lim path (oracle/audio_ref.py) against independent
mathematics.  The reference modules themselves need librosa (absent here): parity against them is unpinned."""
import numpy as np

from oracle import audio_ref as A


def _sig(b=2, n=4800, seed=0):
    rng = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    x = 0.4 * np.sin(2 * np.pi * 220 * t)[None] + 0.1 * rng.randn(b, n)
    return np.clip(x, -1, 1).astype(np.float32)


def test_transform_is_windowed_rfft():
    """stft.py:46-94: the strided convolution with the windowed DFT basis equals rfft(hann * frame)."""
    x = _sig()
    mag, ph = A.transform(x)
    xp = np.pad(x, ((0, 0), (320, 320)), mode='reflect')
    T = (xp.shape[1] - 640) // 160 + 1
    fr = np.stack([xp[:, t * 160:t * 160 + 640] * A.hann(640) for t in range(T)], 1)
    ref = np.fft.rfft(fr, axis=-1).transpose(0, 2, 1)
    assert mag.shape == (2, 321, T)
    assert np.abs(mag - np.abs(ref)).max() < 1e-5
    big = np.abs(ref) > 1e-2
    assert np.abs(np.angle(np.exp(1j * (ph - np.angle(ref)))))[big].max() < 1e-3


def test_inverse_reconstructs_signal():
    """stft.py:96-127: pinv basis + window-envelope division makes inverse(transform(x)) = x."""
    x = _sig()
    rec = A.inverse(*A.transform(x))
    assert rec.shape == x.shape
    assert np.abs(rec - x)[:, 320:-320].max() < 1e-5


def test_window_sumsquare_is_constant_inside():
    ws = A.window_sumsquare(31)
    assert ws.shape == (640 + 160 * 30,)
    assert np.allclose(ws[640:-640], 1.5, atol=1e-5)       # hann^2 at 75 % overlap sums to 3/2


def test_griffin_lim_keeps_magnitude_consistent():
    x = _sig(1, 3200)
    mag, _ = A.transform(x)
    ang0 = np.random.RandomState(1).uniform(-np.pi, np.pi, mag.shape).astype(np.float32)
    e = []
    for it in (0, 20):
        y = A.griffin_lim(mag, ang0, it)
        e.append(np.abs(A.transform(y)[0] - mag).mean())
    assert e[1] < 0.6 * e[0]


def test_mel_filterbank_shape_and_area_norm():
    mb = A.mel_filterbank()
    assert mb.shape == (80, 321) and (mb >= 0).all()
    # Slaney normalisation: every triangle has unit area in Hz (bin spacing 25 Hz)
    area = mb.sum(1) * 25.0
    assert area.min() > 0.85 and area.max() < 1.15 and np.allclose(area[40:], 1.0, atol=0.03)


def test_deemphasis_is_first_order_recursion():
    x = _sig(2, 500)
    y = A.deemphasize(x * 0.05)
    z = np.zeros_like(x[0]); acc = 0.0
    for n in range(500):
        acc = 0.05 * x[0, n] + 0.97 * acc
        z[n] = acc
    assert np.abs(y[0] - np.clip(z, -1, 1)).max() < 1e-5
