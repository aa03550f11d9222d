"""Waveform <-> spectrogram path of validation on the GPU: mirrors of the reference's STFT (src/data/stft.py:35-129),
TacotronSTFT (src/data/vid_aud_grid.py:271-307), griffin_lim (src/data/audio_processing.py:51-66) and the dataset's
inverse_mel / inverse_spec (vid_aud_grid.py:190-224), same names and argument meaning.

The two dense contractions run on the exact-fp32 matrix-core convolution kernels: the hop divides the window, so
with the padded signal viewed as rows of `hop` samples a frame is KT = win/hop consecutive rows; the analysis
transform is a KT-tap convolution over rows (hop input channels -> 2*(win/2+1) output channels) and the synthesis
overlap-add (the reference's conv_transpose1d with stride hop) is exactly that convolution's input gradient.
Everything else is the small kernels of csrc/audio.hip.  Tensors stay on the device for the whole Griffin-Lim loop.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check, ptr, stream


def _hann(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)      # get_window('hann', n, fftbins=True)


def _mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel with its defaults (Slaney mel scale, Slaney area normalisation), vid_aud_grid.py:278-279."""
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, sr / 2.0, n_fft // 2 + 1)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.maximum(0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    return (w * (2.0 / (mel_f[2:] - mel_f[:-2]))[:, None]).astype(np.float32)


class STFT(nn.Module):
    """stft.py:35-129.  transform(x (B, L)) -> magnitude, phase (B, F, T); inverse(magnitude, phase) -> (B, 1, L')."""

    def __init__(self, filter_length=800, hop_length=200, win_length=800, window='hann'):
        super().__init__()
        assert window == 'hann' and win_length == filter_length and filter_length % hop_length == 0
        self.filter_length, self.hop_length, self.win_length, self.window = filter_length, hop_length, win_length, window
        n, hop = filter_length, hop_length
        self.kt = n // hop
        self.cutoff = n // 2 + 1
        fb = np.fft.fft(np.eye(n))
        fb = np.vstack([np.real(fb[:self.cutoff]), np.imag(fb[:self.cutoff])])
        inv = np.linalg.pinv((n / hop) * fb).T
        w = _hann(win_length)
        fwd, inv = (fb * w).astype(np.float32), (inv * w).astype(np.float32)
        # the reference's buffers (same names and shapes, stft.py:65-66) ...
        self.register_buffer('forward_basis', torch.from_numpy(fwd[:, None, :].copy()))
        self.register_buffer('inverse_basis', torch.from_numpy(inv[:, None, :].copy()))
        # ... and the same numbers as convolution weights [2F][hop][kt] over rows of `hop` samples
        self._fwd_w = nn.Parameter(torch.from_numpy(fwd.reshape(-1, self.kt, hop).transpose(0, 2, 1).copy()
                                                    ).reshape(-1, hop, 1, self.kt).contiguous(), requires_grad=False)
        self._inv_w = nn.Parameter(torch.from_numpy(inv.reshape(-1, self.kt, hop).transpose(0, 2, 1).copy()
                                                    ).reshape(-1, hop, 1, self.kt).contiguous(), requires_grad=False)
        self._win_sq = (w ** 2).astype(np.float32)
        self._wsum = {}

    # -- pieces shared with griffin_lim ------------------------------------------------------------------------
    def _rows(self, x):
        """x (B, L) -> reflect-padded rows (B, 1, R, hop), number of frames T."""
        B, L = x.shape
        n, hop = self.filter_length, self.hop_length
        lp = L + n
        T = (lp - n) // hop + 1
        R = T + self.kt - 1
        y = torch.empty((B, 1, R, hop), device=x.device, dtype=torch.float32)
        check(_lib.lib().vca_reflect_pad_f32(B, L, n // 2, R * hop, ptr(ops._c(x)), ptr(y), stream()), 'reflect_pad')
        return y, T

    def _analysis(self, rows):
        """rows (B, 1, R, hop) -> (B, T, 2F) [real | imaginary]."""
        cfg = ops.ConvCfg(self._fwd_w.shape, rows.shape, 1, (0, 0), 0)
        with ops.use_precision('fp32'):
            y = ops._conv_fwd_raw(rows, self._fwd_w, None, cfg)
        return y.view(rows.shape[0], cfg.ow, 2 * self.cutoff)

    def _synthesis(self, rect):
        """rect (B, T, 2F) -> signal (B, hop*(T-1)): overlap-add, window-envelope division, scaling, trimming."""
        B, T, _ = rect.shape
        n, hop = self.filter_length, self.hop_length
        R = T + self.kt - 1
        cfg = ops.ConvCfg(self._inv_w.shape, (B, 1, R, hop), 1, (0, 0), 0)
        with ops.use_precision('fp32'):
            ola = ops._conv_dgrad_raw(rect.reshape(B, 1, T, 2 * self.cutoff), self._inv_w, cfg)   # (B, 1, R, hop)
        ws = self._window_sum(T, rect.device)
        lout = hop * (T - 1)
        y = torch.empty((B, lout), device=rect.device, dtype=torch.float32)
        check(_lib.lib().vca_istft_finish_f32(B, R * hop, lout, n // 2, float(n) / hop,
                                              float(np.finfo(np.float32).tiny), ptr(ola), ptr(ws), ptr(y), stream()),
              'istft_finish')
        return y

    def _window_sum(self, n_frames, device):
        """audio_processing.py:7-48 (window_sumsquare), cached per frame count."""
        key = (n_frames, str(device))
        if key not in self._wsum:
            n, hop = self.filter_length, self.hop_length
            x = np.zeros(n + hop * (n_frames - 1), dtype=np.float32)
            for i in range(n_frames):
                x[i * hop:i * hop + n] += self._win_sq
            self._wsum[key] = torch.from_numpy(x).to(device)
        return self._wsum[key]

    # -- the reference interface ----------------------------------------------------------------------------------
    def transform(self, input_data):
        rows, T = self._rows(input_data)
        spec = self._analysis(rows)[:, :T].contiguous()
        B, F = spec.shape[0], self.cutoff
        mag = torch.empty((B, F, T), device=spec.device, dtype=torch.float32)
        phase = torch.empty_like(mag)
        check(_lib.lib().vca_mag_phase_f32(B, T, F, ptr(spec), ptr(mag), ptr(phase), stream()), 'mag_phase')
        return mag, phase

    def inverse(self, magnitude, phase):
        B, F, T = magnitude.shape
        rect = torch.empty((B, T, 2 * F), device=magnitude.device, dtype=torch.float32)
        check(_lib.lib().vca_polar_to_rect_f32(B, T, F, ptr(ops._c(magnitude)), ptr(ops._c(phase)), ptr(rect),
                                               stream()), 'polar_to_rect')
        return self._synthesis(rect).unsqueeze(1)

    def forward(self, input_data):
        self.magnitude, self.phase = self.transform(input_data)
        return self.inverse(self.magnitude, self.phase)


def griffin_lim(magnitudes, stft_fn, n_iters=30, angles=None):
    """audio_processing.py:51-66.  magnitudes (B, F, T) on the device.  `angles` (B, F, T) replaces the reference's
    np.random.rand draw when given.  The loop never forms an angle: each iteration keeps re/|z|, im/|z| of the
    current transform and imposes the target magnitude."""
    B, F, T = magnitudes.shape
    if angles is None:
        angles = np.angle(np.exp(2j * np.pi * np.random.rand(B, F, T))).astype(np.float32)
        angles = torch.from_numpy(angles)
    mag = ops._c(magnitudes)
    signal = stft_fn.inverse(mag, angles.to(mag.device)).squeeze(1)
    rect = torch.empty((B, T, 2 * F), device=mag.device, dtype=torch.float32)
    for _ in range(n_iters):
        rows, T2 = stft_fn._rows(signal)
        spec = stft_fn._analysis(rows)
        assert T2 == T and spec.shape[1] == T
        check(_lib.lib().vca_gl_project_f32(B, T, F, ptr(spec), ptr(mag), ptr(rect), stream()), 'gl_project')
        signal = stft_fn._synthesis(rect)
    return signal.unsqueeze(1)


class TacotronSTFT(nn.Module):
    """vid_aud_grid.py:271-307."""

    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, sampling_rate=22050,
                 mel_fmin=0.0, mel_fmax=8000.0):
        super().__init__()
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self.stft_fn = STFT(filter_length, hop_length, win_length)
        mel_basis = torch.from_numpy(_mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax))
        self.register_buffer('mel_basis', mel_basis)
        self._mel_w = nn.Parameter(mel_basis.clone(), requires_grad=False)                   # (80, F): spec -> mel
        self._mel_wt = nn.Parameter(mel_basis.t().contiguous(), requires_grad=False)         # (F, 80): mel -> spec

    def spectral_normalize(self, magnitudes):
        return torch.log(torch.clamp(magnitudes, min=1e-5))          # audio_processing.py:69-75

    def spectral_de_normalize(self, magnitudes):
        return torch.exp(magnitudes)                                  # audio_processing.py:78-85

    def mel_spectrogram(self, y):
        """y (B, L) in [-1, 1] -> (log-mel (B, n_mel, T), magnitudes (B, F, T))."""
        assert torch.min(y) >= -1 and torch.max(y) <= 1
        magnitudes, _ = self.stft_fn.transform(y)
        with ops.use_precision('fp32'), torch.no_grad():
            mel = ops.linear(magnitudes.transpose(1, 2).contiguous(), self._mel_w)            # (B, T, n_mel)
        return self.spectral_normalize(mel.transpose(1, 2).contiguous()), magnitudes


_LOG1E5 = float(np.log(1e-5))


def denormalize(melspec):
    return ((melspec + 1) * (-_LOG1E5 / 2)) + _LOG1E5                 # vid_aud_grid.py:238-240


def deemphasize_clip(wav, coef=0.97):
    """scipy.signal.lfilter([1], [1, -0.97], w) and np.clip(-1, 1) of vid_aud_grid.py:203-210,230-232, on the device."""
    wav = ops._c(wav)
    B, L = wav.shape
    out = torch.empty_like(wav)
    check(_lib.lib().vca_deemphasis_clip_f32(B, L, coef, ptr(wav), ptr(out), stream()), 'deemphasis')
    return out


def inverse_mel(mel, stft, n_iters=60, angles=None):
    """vid_aud_grid.py:190-210: normalised log-mel (B, 1, 80, T) -> waveforms (B, L) as a device tensor."""
    if mel.dim() < 4:
        mel = mel.unsqueeze(0)
    mel = stft.spectral_de_normalize(denormalize(mel.float()))
    mel = mel.squeeze(1).transpose(1, 2).contiguous()                                         # (B, T, 80)
    with ops.use_precision('fp32'), torch.no_grad():
        spec = ops.linear(mel, stft._mel_wt)                                                  # (B, T, F)
    spec = (spec * 1000).transpose(1, 2).contiguous()                                         # (B, F, T)
    wav = griffin_lim(spec, stft.stft_fn, n_iters, angles).squeeze(1)
    return deemphasize_clip(wav)


def inverse_spec(spec, stft, n_iters=60, angles=None):
    """vid_aud_grid.py:212-224: linear magnitudes (B, 1, F, T) -> waveforms (B, L) as a device tensor."""
    if spec.dim() < 4:
        spec = spec.unsqueeze(0)
    wav = griffin_lim(spec.squeeze(1), stft.stft_fn, n_iters, angles).squeeze(1)
    return deemphasize_clip(wav)
