"""CPU restatement (numpy) of the reference's waveform <-> spectrogram path used by validation.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing in the product path).

Follows src/data/stft.py:35-129 (STFT.transform / STFT.inverse: windowed DFT bases applied as a strided convolution
and its transpose), src/data/audio_processing.py:7-48 (window_sumsquare), :51-66 (griffin_lim),
src/data/vid_aud_grid.py:190-210 (inverse_mel), :212-224 (inverse_spec), :230-232 (deemphasize), :271-307
(TacotronSTFT.mel_spectrogram).

Parity status: the reference modules import librosa (absent here), so they cannot be run in this container: PARITY
UNPINNED against the reference itself.  The restatement is pinned instead against independent mathematics:
numpy.fft.rfft of the windowed frames for the analysis transform, the reconstruction identity for the synthesis,
scipy.signal.lfilter for the de-emphasis.  The mel filterbank restates librosa.filters.mel's published algorithm
(Slaney mel scale, Slaney area normalisation - librosa's defaults since 0.6; the reference pins no librosa version,
README.md:22-23).
"""
import numpy as np


def hann(n):
    """scipy.signal.get_window('hann', n, fftbins=True) (stft.py:57)."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(np.float64)


def bases(filter_length=640, hop_length=160, win_length=640):
    """forward_basis (2*cutoff, filter_length) and inverse_basis (2*cutoff, filter_length), stft.py:46-66."""
    assert win_length == filter_length, 'pad_center of a shorter window is not needed by the hot path'
    scale = filter_length / hop_length
    fb = np.fft.fft(np.eye(filter_length))
    cutoff = filter_length // 2 + 1
    fb = np.vstack([np.real(fb[:cutoff]), np.imag(fb[:cutoff])])
    inv = np.linalg.pinv(scale * fb).T
    w = hann(win_length)
    return (fb * w).astype(np.float32), (inv * w).astype(np.float32)


def window_sumsquare(n_frames, hop_length=160, win_length=640, n_fft=640):
    """audio_processing.py:7-48 with window='hann', norm=None."""
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, dtype=np.float32)
    win_sq = hann(win_length) ** 2
    for i in range(n_frames):
        s = i * hop_length
        x[s:min(n, s + n_fft)] += win_sq[:max(0, min(n_fft, n - s))]
    return x


def transform(x, filter_length=640, hop_length=160):
    """x (B, L) -> magnitude, phase (B, F, T): stft.py:68-94."""
    fwd, _ = bases(filter_length, hop_length, filter_length)
    pad = filter_length // 2
    xp = np.pad(x.astype(np.float32), ((0, 0), (pad, pad)), mode='reflect')
    T = (xp.shape[1] - filter_length) // hop_length + 1
    idx = np.arange(T)[:, None] * hop_length + np.arange(filter_length)[None, :]
    frames = xp[:, idx]                                    # (B, T, N)
    ft = np.einsum('btn,cn->bct', frames.astype(np.float64), fwd.astype(np.float64))
    cutoff = filter_length // 2 + 1
    re, im = ft[:, :cutoff], ft[:, cutoff:]
    return np.sqrt(re ** 2 + im ** 2).astype(np.float32), np.arctan2(im, re).astype(np.float32)


def inverse(mag, phase, filter_length=640, hop_length=160):
    """magnitude, phase (B, F, T) -> signal (B, hop*(T-1)): stft.py:96-127."""
    _, inv = bases(filter_length, hop_length, filter_length)
    B, F, T = mag.shape
    rect = np.concatenate([mag * np.cos(phase), mag * np.sin(phase)], 1).astype(np.float64)   # (B, 2F, T)
    n = filter_length + hop_length * (T - 1)
    out = np.zeros((B, n), dtype=np.float64)
    fr = np.einsum('bct,cn->btn', rect, inv.astype(np.float64))                               # (B, T, N)
    for t in range(T):
        out[:, t * hop_length:t * hop_length + filter_length] += fr[:, t]
    ws = window_sumsquare(T, hop_length, filter_length, filter_length)
    nz = ws > np.finfo(np.float32).tiny
    out[:, nz] /= ws[nz]
    out *= float(filter_length) / hop_length
    half = filter_length // 2
    return out[:, half:-half].astype(np.float32)


def griffin_lim(mag, angles0, n_iters=60, filter_length=640, hop_length=160):
    """audio_processing.py:51-66 with the initial angles passed in (the reference draws them from numpy's RNG)."""
    sig = inverse(mag, angles0, filter_length, hop_length)
    for _ in range(n_iters):
        _, ang = transform(sig, filter_length, hop_length)
        sig = inverse(mag, ang, filter_length, hop_length)
    return sig


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=16000, n_fft=640, n_mels=80, fmin=55.0, fmax=7500.0):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults (htk=False, norm='slaney'), as called at
    vid_aud_grid.py:278-279.  Returns (n_mels, n_fft//2 + 1) float32."""
    fftfreqs = np.linspace(0, sr / 2.0, n_fft // 2 + 1)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_fft // 2 + 1))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


def mel_spectrogram(y, mel_basis, filter_length=640, hop_length=160):
    """TacotronSTFT.mel_spectrogram, vid_aud_grid.py:291-307: (log-mel (B, 80, T), magnitudes (B, F, T))."""
    mag, _ = transform(y, filter_length, hop_length)
    mel = np.einsum('mf,bft->bmt', mel_basis, mag)
    return np.log(np.clip(mel, 1e-5, None)).astype(np.float32), mag


def deemphasize(w, coef=0.97):
    """scipy.signal.lfilter([1], [1, -0.97], w) then np.clip(-1, 1): vid_aud_grid.py:203-210,230-232."""
    from scipy import signal
    return np.clip(signal.lfilter([1], [1, -coef], w, axis=-1), -1, 1).astype(np.float32)
