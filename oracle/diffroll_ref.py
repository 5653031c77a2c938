"""CPU oracle: a plain restatement of the DiffRoll sampling hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it - as the checker / the timed CPU baseline, never as something the
shipped engine (``diffroll_amd``) calls.

It restates, in torch-CPU fp32 ops and with no import of the reference, the
arithmetic of the reference path (all file:line are under /root/reference):

  * noise schedule            task/diffusion.py:239-256  (+ :28-29)
  * step embedding            model/diffwave.py:58-88
  * network forward           model/diffwave.py:637-686, residual block :134-151
  * spectrogram normalisation model/utils.py:21-32
  * samplers                  task/diffusion.py:831-853 (ddpm_x0), :943-969
                              (cfdg_ddpm_x0), :971-997 (generation_ddpm_x0),
                              :999-1025 (inpainting_ddpm_x0), :855-875 (ddim_x0),
                              :1027-1055 (cfdg_ddim_x0), :804-829 (ddpm), :877-892
                              (ddim), :894-911 (ddim2ddpm)
  * the sampling loop         task/diffusion.py:528-534 / :779-788
  * mel front-end             torchaudio==0.11.0 MelSpectrogram (third party,
                              NOT in /root/reference; pinned in
                              requirements.txt:13).  Call site
                              model/diffwave.py:635,643; arguments
                              config/spec/mel.yaml:1-10.

Pinning status
--------------
The reference has no tests, golden vectors or known-answer fixtures
(SURVEY.md section 4).  Everything except the mel front-end is pinned against
the reference itself: ``tests/golden/make_golden.py`` imports
/root/reference (with stub modules for the packages missing in this image),
runs it on seeded inputs and commits the inputs/outputs as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against those vectors.

**Mel front-end: parity unpinned for the filterbank half.**  torchaudio is
absent from this image, so ``MelSpectrogram`` is restated here from the
documented torchaudio 0.11 algorithm (``torch.stft`` - present - pins the STFT
half; the HTK triangular filterbank follows torchaudio.functional
.melscale_fbanks as published).  The golden vectors for the front-end are
therefore outputs of THIS restatement driven through the reference's own
``forward`` (which calls it via the stubbed ``torchaudio`` module).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------
# hyper-parameters (config/model/ClassifierFreeDiffRoll.yaml:1-15,
# config/task/transcription.yaml, config/spec/mel.yaml:1-10, config/sampling.yaml:1-4)
# --------------------------------------------------------------------------
DEFAULT_HP = dict(
    residual_channels=512,
    residual_layers=15,
    kernel_size=9,
    dilation_base=2,
    dilation_bound=4,
    n_mels=229,
    timesteps=200,
    beta_start=1e-4,
    beta_end=0.02,
    sample_rate=16000,
    n_fft=2048,
    hop_length=512,
    f_min=0.0,
    f_max=8000.0,
)


def layer_dilation(hp: dict, i: int) -> int:
    """model/diffwave.py:623: dilation_base ** (i % dilation_bound)."""
    return int(hp["dilation_base"]) ** (i % int(hp["dilation_bound"]))


def conv_padding(kernel_size: int, dilation: int) -> int:
    """model/diffwave.py:124."""
    return ((kernel_size - 1) * (dilation - 1) + kernel_size - 1) // 2


# --------------------------------------------------------------------------
# noise schedule  (task/diffusion.py:239-256)
# --------------------------------------------------------------------------
def schedule(beta_start: float, beta_end: float, timesteps: int, betas: Optional[Tensor] = None) -> Dict[str, Tensor]:
    if betas is None:
        betas = torch.linspace(beta_start, beta_end, timesteps)  # :28-29
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, axis=0)
    alphas_cumprod_prev = F.pad(alphas_cumprod[:-1], (1, 0), value=1.0)
    return dict(
        betas=betas,
        alphas=alphas,
        sqrt_recip_alphas=torch.sqrt(1.0 / alphas),
        sqrt_alphas_cumprod=torch.sqrt(alphas_cumprod),
        sqrt_one_minus_alphas_cumprod=torch.sqrt(1.0 - alphas_cumprod),
        posterior_variance=betas * (1.0 - alphas_cumprod_prev) / (1 - alphas_cumprod),
    )


# --------------------------------------------------------------------------
# step embedding  (model/diffwave.py:58-88)
# --------------------------------------------------------------------------
def build_embedding(max_steps: int) -> Tensor:
    """model/diffwave.py:83-88 (_build_embedding)."""
    steps = torch.arange(max_steps).unsqueeze(1)
    dims = torch.arange(64).unsqueeze(0)
    table = steps * 10.0 ** (dims * 4.0 / 63.0)
    return torch.cat([torch.sin(table), torch.cos(table)], dim=1)


def silu(x: Tensor) -> Tensor:
    """model/diffwave.py:53-55."""
    return x * torch.sigmoid(x)


def diffusion_embedding(params: Dict[str, Tensor], table: Tensor, t: Tensor) -> Tensor:
    """model/diffwave.py:65-74 (integer steps only; the lerp branch is off-path)."""
    x = table[t]
    x = F.linear(x, params["diffusion_embedding.projection1.weight"],
                 params["diffusion_embedding.projection1.bias"])
    x = silu(x)
    x = F.linear(x, params["diffusion_embedding.projection2.weight"],
                 params["diffusion_embedding.projection2.bias"])
    return silu(x)


# --------------------------------------------------------------------------
# mel front-end  (torchaudio 0.11 MelSpectrogram, restated; see module docstring)
# --------------------------------------------------------------------------
def _hz_to_mel_htk(f: float) -> float:
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks_htk(n_freqs: int, f_min: float, f_max: float, n_mels: int,
                        sample_rate: int) -> Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> (n_freqs, n_mels)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = _hz_to_mel_htk(f_min)
    m_max = _hz_to_mel_htk(f_max)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down_slopes, up_slopes))


def mel_spectrogram(waveform: Tensor, hp: dict) -> Tensor:
    """MelSpectrogram(sample_rate, n_fft, hop_length, n_mels, f_min, f_max, center=True,
    normalized=True, pad_mode='reflect') with torchaudio 0.11 defaults: win_length=n_fft,
    periodic Hann, power=2.0, onesided, norm=None, mel_scale='htk'.
    (B, L) -> (B, n_mels, L//hop + 1).  Call site model/diffwave.py:643."""
    n_fft = int(hp["n_fft"])
    hop = int(hp["hop_length"])
    window = torch.hann_window(n_fft)
    spec_f = torch.stft(waveform, n_fft=n_fft, hop_length=hop, win_length=n_fft, window=window,
                        center=True, pad_mode="reflect", normalized=False, onesided=True,
                        return_complex=True)
    spec_f = spec_f / window.pow(2.0).sum().sqrt()   # normalized=True ('window')
    spec = spec_f.abs().pow(2.0)                      # power=2.0
    fb = melscale_fbanks_htk(n_fft // 2 + 1, float(hp["f_min"]), float(hp["f_max"]),
                             int(hp["n_mels"]), int(hp["sample_rate"]))
    return torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)


def normalize_imagewise(x: Tensor, lo: float = 0.0, hi: float = 1.0) -> Tensor:
    """model/utils.py:21-32: per-sample min-max over all (F, T) values; NaN -> lo."""
    x_max = x.flatten(1).max(1, keepdim=True)[0].unsqueeze(1)
    x_min = x.flatten(1).min(1, keepdim=True)[0].unsqueeze(1)
    x_std = (x - x_min) / (x_max - x_min)
    x_scaled = x_std * (hi - lo) + lo
    x_scaled[torch.isnan(x_scaled)] = lo
    return x_scaled


def normalize_framewise(x: Tensor, lo: float = 0.0, hi: float = 1.0) -> Tensor:
    """model/utils.py:11-19: min-max over the frequency axis of every frame of (B, F, T); NaN (constant frame)
    -> 0 BEFORE the affine map."""
    x_max = x.max(1, keepdim=True)[0]
    x_min = x.min(1, keepdim=True)[0]
    x_std = (x - x_min) / (x_max - x_min)
    x_std[torch.isnan(x_std)] = 0
    return x_std * (hi - lo) + lo


def frontend(waveform: Tensor, hp: dict, T_roll: int, sampling: bool = False,
             inpainting_t: Optional[Sequence[int]] = None,
             inpainting_f: Optional[Sequence[int]] = None) -> Tensor:
    """model/diffwave.py:643-662: mel -> log(+1e-6) -> imagewise [0,1] -> inpainting mask
    -> (sampling=True: all -1) -> trim to min(T_roll, T_spec).  Returns (B, n_mels, T)."""
    spec = mel_spectrogram(waveform, hp)
    spec = torch.log(spec + 1e-6)
    if hp.get("norm_mode", "imagewise") == "framewise":        # norm_args[2], model/diffwave.py:632
        spec = normalize_framewise(spec, 0.0, 1.0)
    else:
        spec = normalize_imagewise(spec, 0.0, 1.0)
    if inpainting_t and inpainting_f is None:
        spec[:, :, int(inpainting_t[0]):int(inpainting_t[1])] = -1
    elif inpainting_t is None and inpainting_f:
        spec[:, int(inpainting_f[0]):int(inpainting_f[1]), :] = -1
    elif inpainting_t and inpainting_f:
        spec[:, int(inpainting_f[0]):int(inpainting_f[1]),
             int(inpainting_t[0]):int(inpainting_t[1])] = -1
    if sampling:
        spec = torch.full_like(spec, -1)
    T_min = min(T_roll, spec.shape[-1])
    return spec[..., :T_min]


def uncond_spec(params: Dict[str, Tensor], hp: dict, like: Tensor) -> Tensor:
    """The spectrogram of the unconditional branch (forward(sampling=True), model/diffwave.py:656-660):
    all -1 for condition 'fixed'; for condition 'trainable_spec' the learned (n_mels, 641) parameter, trimmed to
    the roll length (trim_spec_roll, :30-39) and broadcast over the batch (the reference feeds it 2-D)."""
    if hp.get("condition", "fixed") == "trainable_spec":
        T = like.shape[-1]
        return params["trainable_parameters"][:, :T].unsqueeze(0).expand(like.shape[0], -1, -1)
    return torch.full_like(like, -1)


# --------------------------------------------------------------------------
# network  (model/diffwave.py:134-151, :664-686)
# --------------------------------------------------------------------------
def residual_block(params: Dict[str, Tensor], i: int, hp: dict, x: Tensor, emb: Tensor,
                   spectrogram: Tensor) -> Tuple[Tensor, Tensor]:
    """model/diffwave.py:134-151."""
    p = f"residual_layers.{i}."
    k = int(hp["kernel_size"])
    dil = layer_dilation(hp, i)
    d = F.linear(emb, params[p + "diffusion_projection.weight"],
                 params[p + "diffusion_projection.bias"]).unsqueeze(-1)
    y = x + d
    cond = F.conv1d(spectrogram, params[p + "conditioner_projection.weight"],
                    params[p + "conditioner_projection.bias"])
    y = F.conv1d(y, params[p + "dilated_conv.weight"], params[p + "dilated_conv.bias"],
                 padding=conv_padding(k, dil), dilation=dil) + cond
    gate, filt = torch.chunk(y, 2, dim=1)
    y = torch.sigmoid(gate) * torch.tanh(filt)
    y = F.conv1d(y, params[p + "output_projection.weight"], params[p + "output_projection.bias"])
    residual, skip = torch.chunk(y, 2, dim=1)
    return (x + residual) / math.sqrt(2.0), skip


def denoise(params: Dict[str, Tensor], hp: dict, x_t: Tensor, spectrogram: Tensor,
            t: Tensor, table: Optional[Tensor] = None) -> Tensor:
    """The part of forward() after the front-end (model/diffwave.py:664-686).
    x_t (B,1,T,88), spectrogram (B,n_mels,T), t (B,) int64 -> x0_pred (B,1,T,88)."""
    if table is None:
        table = build_embedding(int(hp["timesteps"]))
    L = int(hp["residual_layers"])
    x = x_t.squeeze(1).transpose(1, 2)
    x = x[..., :spectrogram.shape[-1]]
    x = F.conv1d(x, params["input_projection.weight"], params["input_projection.bias"])
    x = F.relu(x)
    emb = diffusion_embedding(params, table, t)
    skip = None
    for i in range(L):
        x, s = residual_block(params, i, hp, x, emb, spectrogram)
        skip = s if skip is None else s + skip
    x = skip / math.sqrt(L)
    x = F.conv1d(x, params["skip_projection.weight"], params["skip_projection.bias"])
    x = F.relu(x)
    x = F.conv1d(x, params["output_projection.weight"], params["output_projection.bias"])
    return x.transpose(1, 2).unsqueeze(1)


def forward(params: Dict[str, Tensor], hp: dict, x_t: Tensor, waveform: Tensor, t: Tensor,
            sampling: bool = False, inpainting_t=None, inpainting_f=None,
            table: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """ClassifierFreeDiffRoll.forward in eval mode (model/diffwave.py:637-686)."""
    spec = frontend(waveform, hp, x_t.shape[2], sampling, inpainting_t, inpainting_f)
    if sampling:
        spec = uncond_spec(params, hp, spec)
    return denoise(params, hp, x_t, spec, t, table), spec


# --------------------------------------------------------------------------
# samplers  (task/diffusion.py:831-853, :943-1025)
# --------------------------------------------------------------------------
def q_sample(x_start: Tensor, t: Tensor, sqrt_alphas_cumprod: Tensor, sqrt_one_minus_alphas_cumprod: Tensor,
             noise: Tensor) -> Tensor:
    """task/diffusion.py:31-46: forward process, per-sample step index t (B,) broadcast over (B,1,T,F)."""
    a = sqrt_alphas_cumprod[t][:, None, None, None]
    c = sqrt_one_minus_alphas_cumprod[t][:, None, None, None]
    return a * x_start + c * noise


def extract_x0(x_t: Tensor, epsilon: Tensor, t: Tensor, sqrt_alphas_cumprod: Tensor,
               sqrt_one_minus_alphas_cumprod: Tensor) -> Tensor:
    """task/diffusion.py:49-64: x_0 from x_t and the predicted noise (inverse of eq. 4 of DDPM)."""
    a = sqrt_alphas_cumprod[t][:, None, None, None]
    c = sqrt_one_minus_alphas_cumprod[t][:, None, None, None]
    return (x_t - c * epsilon) / a


def posterior_update(sch: Dict[str, Tensor], x: Tensor, x0_pred: Tensor, t_index: int,
                     z: Optional[Tensor]) -> Tensor:
    """task/diffusion.py:957-967 (identical in all *_ddpm_x0 samplers)."""
    sac = sch["sqrt_alphas_cumprod"]
    s1m = sch["sqrt_one_minus_alphas_cumprod"]
    alphas = sch["alphas"]
    if t_index == 0:
        return x0_pred / sac[t_index]
    sigma = (s1m[t_index - 1] / s1m[t_index]) * torch.sqrt(1 - alphas[t_index])
    return (sac[t_index - 1]) * x0_pred + (
        torch.sqrt(1 - sac[t_index - 1] ** 2 - sigma ** 2) * (
            x - sac[t_index] * x0_pred) / s1m[t_index]) + (sigma * z)


def eps_update(sch: Dict[str, Tensor], sampler: str, x: Tensor, eps: Tensor, t_index: int,
               z: Optional[Tensor]) -> Tensor:
    """The epsilon-prediction updates: ddpm (task/diffusion.py:804-829), ddim (:877-892),
    ddim2ddpm (:894-911).  Same expressions, same order."""
    sac = sch["sqrt_alphas_cumprod"]
    s1m = sch["sqrt_one_minus_alphas_cumprod"]
    if sampler == "ddpm":
        model_mean = sch["sqrt_recip_alphas"][t_index] * (x - sch["betas"][t_index] * eps / s1m[t_index])
        if t_index == 0:
            return model_mean
        return model_mean + torch.sqrt(sch["posterior_variance"][t_index]) * z
    if t_index == 0:
        return (x - s1m[t_index] * eps) / sac[t_index]
    if sampler == "ddim":
        return (sac[t_index - 1]) * ((x - s1m[t_index] * eps) / sac[t_index]) + (s1m[t_index - 1] * eps)
    if sampler == "ddim2ddpm":
        sigma = (s1m[t_index - 1] / s1m[t_index]) * torch.sqrt(1 - sch["alphas"][t_index])
        return (sac[t_index - 1]) * ((x - s1m[t_index] * eps) / sac[t_index]) + (
            torch.sqrt(1 - sac[t_index - 1] ** 2 - sigma ** 2) * eps) + sigma * z
    raise ValueError(sampler)


def ddim_x0_update(sch: Dict[str, Tensor], x: Tensor, x0_pred: Tensor, t_index: int) -> Tensor:
    """ddim_x0 / cfdg_ddim_x0 (task/diffusion.py:864-873, :1044-1053): the x0 update with sigma = 0
    (the reference still adds 0 * randn_like(x))."""
    sac = sch["sqrt_alphas_cumprod"]
    s1m = sch["sqrt_one_minus_alphas_cumprod"]
    if t_index == 0:
        return x0_pred / sac[t_index]
    sigma = 0
    return (sac[t_index - 1]) * x0_pred + (
        torch.sqrt(1 - sac[t_index - 1] ** 2 - sigma ** 2) * (x - sac[t_index] * x0_pred) / s1m[t_index])


def reverse_step(params, hp, sch, sampler: str, x: Tensor, spec_c: Optional[Tensor],
                 t_index: int, z: Optional[Tensor], w: float = 0.0,
                 table: Optional[Tensor] = None) -> Tensor:
    """One reverse-diffusion step with the front-end hoisted (it is clip-invariant).
    spec_c: conditional spectrogram (B,n_mels,T) (already masked for inpainting);
    the unconditional branch uses spec = -1 (model/diffwave.py:656-660)."""
    B = x.shape[0]
    t = torch.tensor(t_index).repeat(B)
    if sampler in ("cfdg_ddpm_x0", "inpainting_ddpm_x0"):
        x0_c = denoise(params, hp, x, spec_c, t, table)
        x0_u = denoise(params, hp, x, uncond_spec(params, hp, spec_c), t, table)
        x0 = (1 + w) * x0_c - w * x0_u                      # :953 / :1009
    elif sampler == "generation_ddpm_x0":
        T = x.shape[2]
        spec_u = uncond_spec(params, hp, torch.empty(B, int(hp["n_mels"]), T))
        x0 = denoise(params, hp, x, spec_u, t, table)         # :979-980
    elif sampler == "ddpm_x0":
        x0 = denoise(params, hp, x, spec_c, t, table)         # :839
    elif sampler == "ddim_x0":
        return ddim_x0_update(sch, x, denoise(params, hp, x, spec_c, t, table), t_index)      # :863
    elif sampler == "cfdg_ddim_x0":
        # :1039-1041.  NB the "unconditional" branch is forward(zeros_like(waveform)) WITHOUT
        # sampling=True: the spectrogram of silence normalises to NaN -> 0 (model/utils.py:29-31),
        # i.e. spec == 0 everywhere, not -1.
        x0_c = denoise(params, hp, x, spec_c, t, table)
        x0_z = denoise(params, hp, x, torch.zeros_like(spec_c), t, table)
        return ddim_x0_update(sch, x, (1 + w) * x0_c - w * x0_z, t_index)
    elif sampler in ("ddpm", "ddim", "ddim2ddpm"):
        return eps_update(sch, sampler, x, denoise(params, hp, x, spec_c, t, table), t_index, z)
    else:
        raise ValueError(sampler)
    return posterior_update(sch, x, x0, t_index, z)


def sample_chain(params, hp, sampler: str, x_T: Tensor, waveform: Optional[Tensor],
                 noise: Tensor, w: float = 0.0, inpainting_t=None, inpainting_f=None,
                 steps: Optional[Sequence[int]] = None) -> Tensor:
    """The loop of task/diffusion.py:528-534: t = S-1 .. 0.  ``noise[t]`` is the z drawn at
    step t (t >= 1; the reference draws none at t == 0).  Returns the final roll."""
    S = int(hp["timesteps"])
    sch = schedule(float(hp["beta_start"]), float(hp["beta_end"]), S)
    table = build_embedding(S)
    T = x_T.shape[2]
    spec_c = None
    if sampler != "generation_ddpm_x0":
        it = inpainting_t if sampler == "inpainting_ddpm_x0" else None
        i_f = inpainting_f if sampler == "inpainting_ddpm_x0" else None
        spec_c = frontend(waveform, hp, T, False, it, i_f)
    x = x_T
    for t_index in (reversed(range(S)) if steps is None else steps):
        z = noise[t_index] if t_index > 0 else None
        x = reverse_step(params, hp, sch, sampler, x, spec_c, t_index, z, w, table)
    return x


# --------------------------------------------------------------------------
# synthetic weights: same shapes / init family as the reference
# constructors; output_projection re-initialised N(0, 0.02^2) because the reference
# zero-inits it (model/diffwave.py:630).
# --------------------------------------------------------------------------
def synthetic_params(hp: dict, seed: int = 0) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    C = int(hp["residual_channels"])
    L = int(hp["residual_layers"])
    k = int(hp["kernel_size"])
    M = int(hp["n_mels"])

    def kaiming(co, ci, kk):      # nn.init.kaiming_normal_ (fan_in, gain sqrt(2))
        return torch.randn(co, ci, kk, generator=g) * math.sqrt(2.0 / (ci * kk))

    def unif(shape, fan_in):      # default nn.Linear / nn.Conv1d bias & Linear weight init
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(*shape, generator=g) * 2 - 1) * b

    p: Dict[str, Tensor] = {}
    p["input_projection.weight"] = kaiming(C, 88, 1)
    p["input_projection.bias"] = unif((C,), 88)
    p["diffusion_embedding.projection1.weight"] = unif((512, 128), 128)
    p["diffusion_embedding.projection1.bias"] = unif((512,), 128)
    p["diffusion_embedding.projection2.weight"] = unif((512, 512), 512)
    p["diffusion_embedding.projection2.bias"] = unif((512,), 512)
    for i in range(L):
        q = f"residual_layers.{i}."
        p[q + "dilated_conv.weight"] = kaiming(2 * C, C, k)
        p[q + "dilated_conv.bias"] = unif((2 * C,), C * k)
        p[q + "diffusion_projection.weight"] = unif((C, 512), 512)
        p[q + "diffusion_projection.bias"] = unif((C,), 512)
        p[q + "conditioner_projection.weight"] = kaiming(2 * C, M, 1)
        p[q + "conditioner_projection.bias"] = unif((2 * C,), M)
        p[q + "output_projection.weight"] = kaiming(2 * C, C, 1)
        p[q + "output_projection.bias"] = unif((2 * C,), C)
    p["skip_projection.weight"] = kaiming(C, C, 1)
    p["skip_projection.bias"] = unif((C,), C)
    p["output_projection.weight"] = torch.randn(88, C, 1, generator=g) * 0.02
    p["output_projection.bias"] = unif((88,), C)
    if hp.get("condition", "fixed") == "trainable_spec":      # model/diffwave.py:601 (initialised to -1; here: "trained")
        p["trainable_parameters"] = torch.rand(M, 641, generator=g) * 2 - 1
    return p


# --------------------------------------------------------------------------
# post-processing: roll -> notes  (task/diffusion.py:1185-1233)
# --------------------------------------------------------------------------
def extract_notes_wo_velocity(onsets, frames, onset_threshold=0.5, frame_threshold=0.5, rule="rule1"):
    """numpy restatement of task/diffusion.py:1185-1233: a note starts at a rising edge of the thresholded
    onset roll (rule1: where the frame roll is on too) and lasts while either roll stays on.
    onsets, frames: (T, 88) arrays -> (pitches (N,), intervals (N, 2) [onset, offset))."""
    import numpy as np
    onsets = (np.asarray(onsets) > onset_threshold).astype(int)
    frames = (np.asarray(frames) > frame_threshold).astype(int)
    onset_diff = np.concatenate([onsets[:1, :], onsets[1:, :] - onsets[:-1, :]], axis=0) == 1
    if rule == "rule1":
        onset_diff = onset_diff & (frames == 1)
    elif rule != "rule2":
        raise NameError("Please enter the correct rule name")
    pitches, intervals = [], []
    frame_locs, pitch_locs = np.nonzero(onset_diff)
    for frame, pitch in zip(frame_locs, pitch_locs):
        onset = offset = frame
        while onsets[offset, pitch] or frames[offset, pitch]:
            offset += 1
            if offset == onsets.shape[0]:
                break
        if offset > onset:
            pitches.append(pitch)
            intervals.append([onset, offset])
    return np.array(pitches), np.array(intervals)
