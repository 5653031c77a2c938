"""Host-built constants of the mel front-end, with the reference's own arithmetic.

``torchaudio.transforms.MelSpectrogram`` (model/diffwave.py:635; torchaudio 0.11) builds its Hann window with
``torch.hann_window`` and its filterbank with ``torchaudio.functional.melscale_fbanks`` - both in fp32.  The fp32
rounding of the filterbank is NOT negligible: against the exact (double precision) triangles its weights differ
by up to 0.7 % for the narrow low-frequency filters, which moves the normalised log-mel by 2e-5 - ten times the
rest of the front-end's error.  Parity therefore needs the SAME expressions in the same precision, evaluated by
torch on the host (as diffroll_amd.schedule does for the noise schedule); the engine receives the tables through
``dr_set_frontend_tables`` and does the per-clip arithmetic (FFT, |.|^2, filterbank GEMM, log, normalise).
"""
from __future__ import annotations

import math
from typing import Tuple

import torch


def hann_window(n_fft: int) -> torch.Tensor:
    """MelSpectrogram's default window: ``torch.hann_window(win_length)`` (periodic), fp32."""
    return torch.hann_window(n_fft)


def window_norm(window: torch.Tensor) -> float:
    """``normalized=True``: the spectrum is divided by ``window.pow(2.).sum().sqrt()`` (fp32, torchaudio
    functional.spectrogram)."""
    return float(window.pow(2.0).sum().sqrt())


def _hz_to_mel_htk(f: float) -> float:
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks_htk(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None,
    mel_scale='htk') -> (n_freqs, n_mels), the published formula evaluated with the same fp32 tensor operations:
    linspace of the bin frequencies and of the mel points, mel -> Hz, the two slopes, max(0, min(down, up))."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel_htk(f_min), _hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down_slopes, up_slopes))


def frontend_tables(n_fft: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> Tuple[torch.Tensor, float, torch.Tensor]:
    """(window (n_fft,), window norm, filterbank (n_fft // 2 + 1, n_mels)) as contiguous fp32 host tensors."""
    w = hann_window(n_fft).contiguous().float()
    fb = melscale_fbanks_htk(n_fft // 2 + 1, float(f_min), float(f_max), int(n_mels), int(sample_rate)).contiguous().float()
    return w, window_norm(w), fb
