"""Audio ingestion of the reference's Custom dataset (utils/custom_dataset.py:55-91): load -> mono -> resample ->
crop / zero-pad.  Host-side, as in the reference (it runs in the DataLoader, once per file).

``resample`` restates ``torchaudio.functional.resample`` as of torchaudio 0.11 (requirements.txt:13; the call at
utils/custom_dataset.py:62 uses its defaults: windowed-sinc interpolation with a Hann window,
``lowpass_filter_width=6``, ``rolloff=0.99``): the polyphase kernel bank is evaluated in float64 and rounded to
float32, the waveform is padded by (width, width + orig_freq) and convolved with stride orig_freq.  torchaudio is
not installed here, so this is pinned by construction (the published algorithm, same torch operations in the same
order) and by properties (tests/test_host_cpu.py): PARITY UNPINNED against the package itself.
mp3 decoding needs a codec this image lacks and is not attempted.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
                         rolloff: float = 0.99) -> Tuple[torch.Tensor, int, int, int]:
    """torchaudio 0.11 ``_get_sinc_resample_kernel`` (resampling_method='sinc_interpolation', dtype=None):
    -> (kernels (new_freq', 1, 2 * width + orig_freq') float32, width, orig_freq', new_freq') with the rates reduced
    by their gcd."""
    gcd = math.gcd(int(orig_freq), int(new_freq))
    orig = int(orig_freq) // gcd
    new = int(new_freq) // gcd
    base_freq = min(orig, new) * rolloff                     # anti-aliasing cut-off, in units of the reduced rates
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)
    kernels = []
    for i in range(new):
        t = (-i / new + idx / orig) * base_freq
        t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
        window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2        # Hann window over the clamped support
        t *= math.pi
        kernel = torch.where(t == 0, torch.tensor(1.0).to(t), torch.sin(t) / t)
        kernel.mul_(window)
        kernels.append(kernel)
    scale = base_freq / orig
    bank = torch.stack(kernels).view(new, 1, -1).mul_(scale).to(dtype=torch.float32)
    return bank, width, orig, new


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
             rolloff: float = 0.99) -> torch.Tensor:
    """torchaudio.functional.resample(waveform, orig_freq, new_freq) with the 0.11 defaults; (..., L) ->
    (..., ceil(new * L / orig))."""
    if int(orig_freq) == int(new_freq):
        return waveform
    bank, width, orig, new = sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff)
    shape = waveform.size()
    wav = waveform.reshape(-1, shape[-1]).to(torch.float32)
    num_wavs, length = wav.shape
    wav = torch.nn.functional.pad(wav, (width, width + orig))
    out = torch.nn.functional.conv1d(wav[:, None], bank, stride=orig)
    out = out.transpose(1, 2).reshape(num_wavs, -1)
    target_length = int(math.ceil(new * length / orig))
    out = out[..., :target_length]
    return out.view(shape[:-1] + out.shape[-1:])


def to_mono(waveform: torch.Tensor) -> torch.Tensor:
    """utils/custom_dataset.py:56-59 on a (channels, L) tensor: the mean of EXACTLY two channels, else channel 0."""
    return waveform.mean(0) if waveform.shape[0] == 2 else waveform[0]


def crop_or_pad(waveform: torch.Tensor, segment_samples: int) -> torch.Tensor:
    """utils/custom_dataset.py:82-86: the first segment_samples samples, zero-padded at the end when shorter."""
    if waveform.shape[0] >= segment_samples:
        return waveform[:segment_samples]
    return torch.nn.functional.pad(waveform, [0, segment_samples - waveform.shape[0]], value=0)


def load_wav(path: str) -> Tuple[torch.Tensor, int]:
    """A .wav file as torchaudio.load returns it: ((channels, L) float32 in [-1, 1), sample rate); integer PCM is
    scaled by 2^(bits-1) (8-bit PCM is unsigned with a 128 offset)."""
    import numpy as np
    from scipy.io import wavfile
    rate, data = wavfile.read(path)
    if data.ndim == 1:
        data = data[:, None]
    if np.issubdtype(data.dtype, np.integer):
        if data.dtype == np.uint8:
            x = (data.astype(np.float32) - 128.0) / 128.0
        else:
            x = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    else:
        x = data.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x.T)), int(rate)


def ingest(path: str, sample_rate: int, segment_samples: int) -> torch.Tensor:
    """One clip as AudioDataset.__getitem__ builds it (utils/custom_dataset.py:55-91): (segment_samples,) float32."""
    wav, rate = load_wav(path)
    mono = to_mono(wav)
    if rate != sample_rate:
        mono = resample(mono, rate, sample_rate)
    return crop_or_pad(mono, int(segment_samples))
