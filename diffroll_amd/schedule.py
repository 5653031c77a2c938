"""Host-side constant tables of the sampler (tiny, built once per model).

The reference keeps these as plain CPU tensors computed in its constructor
(task/diffusion.py:239-256) and evaluates the per-step scalars inline at every step
(:957-967).  They are evaluated here with the same torch fp32 expressions so the
coefficients the update kernel reads are bit-equal to the reference's, then handed to the
engine through dr_set_tables().
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def linear_beta_schedule(beta_start: float, beta_end: float, timesteps: int) -> torch.Tensor:
    """task/diffusion.py:28-29."""
    return torch.linspace(beta_start, beta_end, timesteps)


_BETA_LO, _BETA_HI = 0.0001, 0.02     # the fixed end points of the quadratic / sigmoid schedules


def cosine_beta_schedule(timesteps, s=0.008):
    """Nichol & Dhariwal's cosine schedule (arXiv:2102.09672) as model/unet.py:558-567 evaluates it: the
    normalised squared-cosine curve f on timesteps + 1 grid points, beta_t = 1 - f[t+1] / f[t], clipped."""
    grid = torch.linspace(0, timesteps, timesteps + 1)
    f = torch.cos(((grid / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    f = f / f[0]
    return torch.clip(1 - (f[1:] / f[:-1]), 0.0001, 0.9999)


def quadratic_beta_schedule(timesteps):
    """model/unet.py:570-573: linear in sqrt(beta)."""
    return torch.linspace(_BETA_LO**0.5, _BETA_HI**0.5, timesteps) ** 2


def sigmoid_beta_schedule(timesteps):
    """model/unet.py:575-579: a sigmoid ramp over [-6, 6] between the two end points."""
    ramp = torch.sigmoid(torch.linspace(-6, 6, timesteps))
    return ramp * (_BETA_HI - _BETA_LO) + _BETA_LO


def make_schedule(beta_start: float, beta_end: float, timesteps: int, betas: torch.Tensor = None) -> Dict[str, torch.Tensor]:
    """The six schedule vectors of SpecRollDiffusion.__init__ (task/diffusion.py:239-256).  `betas` replaces the
    linear schedule (e.g. one of the model/unet.py schedules above): everything downstream - the coefficient
    tables the engine reads - only sees these vectors."""
    betas = linear_beta_schedule(beta_start, beta_end, timesteps) if betas is None else betas.to(torch.float32)
    alphas = 1. - betas
    alphas_cumprod = torch.cumprod(alphas, axis=0)
    alphas_cumprod_prev = F.pad(alphas_cumprod[:-1], (1, 0), value=1.0)
    return {
        "betas": betas,
        "alphas": alphas,
        "sqrt_recip_alphas": torch.sqrt(1.0 / alphas),
        "sqrt_alphas_cumprod": torch.sqrt(alphas_cumprod),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1. - alphas_cumprod),
        "posterior_variance": betas * (1. - alphas_cumprod_prev) / (1 - alphas_cumprod),
    }


def posterior_coef_table(sch: Dict[str, torch.Tensor]) -> torch.Tensor:
    """(S, 5) fp32: [sqrt_acp[t-1], sqrt(1 - sqrt_acp[t-1]**2 - sigma**2), sqrt_acp[t],
    sqrt_1m_acp[t], sigma] - the scalars of the x0-prediction DDPM update,
    task/diffusion.py:957-967 (identical in ddpm_x0 / cfdg / generation / inpainting).
    Row 0 only uses column 2 (x = x0 / sqrt_acp[0]; no noise)."""
    sac = sch["sqrt_alphas_cumprod"]
    s1m = sch["sqrt_one_minus_alphas_cumprod"]
    alphas = sch["alphas"]
    S = sac.shape[0]
    out = torch.zeros(S, 5, dtype=torch.float32)
    for t in range(S):
        if t == 0:
            sigma = (1 / s1m[t]) * torch.sqrt(1 - alphas[t])
            out[t, 2] = sac[t]
            out[t, 3] = s1m[t]
            out[t, 4] = sigma
        else:
            sigma = (s1m[t - 1] / s1m[t]) * torch.sqrt(1 - alphas[t])
            out[t, 0] = sac[t - 1]
            out[t, 1] = torch.sqrt(1 - sac[t - 1] ** 2 - sigma ** 2)
            out[t, 2] = sac[t]
            out[t, 3] = s1m[t]
            out[t, 4] = sigma
    return out


def build_embedding(max_steps: int) -> torch.Tensor:
    """Sinusoidal step table (S, 128) of DiffusionEmbedding (model/diffwave.py:83-88)."""
    steps = torch.arange(max_steps).unsqueeze(1)
    dims = torch.arange(64).unsqueeze(0)
    table = steps * 10.0 ** (dims * 4.0 / 63.0)
    return torch.cat([torch.sin(table), torch.cos(table)], dim=1)


def sampler_coef_tables(sch: Dict[str, torch.Tensor]) -> torch.Tensor:
    """(5, S, 5) fp32: one table of per-step scalars per coefficient family (DR_COEF_* in
    include/diffroll_amd.h), each scalar evaluated with the reference's own torch expression:
      0 ddpm_x0 family   task/diffusion.py:957-967   (posterior_coef_table)
      1 ddim_x0 family   :864-873, :1044-1053        (sigma = 0)
      2 ddpm  (epsilon)  :807-829    [sqrt_recip_alphas[t], betas[t], sqrt_1m_acp[t], sqrt(posterior_variance[t]), 0]
      3 ddim  (epsilon)  :885-890    [sqrt_acp[t-1], sqrt_1m_acp[t-1], sqrt_acp[t], sqrt_1m_acp[t], 0]
      4 ddim2ddpm (eps)  :902-909    [sqrt_acp[t-1], sqrt(1 - sqrt_acp[t-1]**2 - sigma**2), sqrt_acp[t], sqrt_1m_acp[t], sigma]
    """
    sac = sch["sqrt_alphas_cumprod"]
    s1m = sch["sqrt_one_minus_alphas_cumprod"]
    alphas = sch["alphas"]
    S = sac.shape[0]
    out = torch.zeros(5, S, 5, dtype=torch.float32)
    out[0] = posterior_coef_table(sch)
    for t in range(S):
        out[1, t, 2] = sac[t]
        out[1, t, 3] = s1m[t]
        out[2, t, 0] = sch["sqrt_recip_alphas"][t]
        out[2, t, 1] = sch["betas"][t]
        out[2, t, 2] = s1m[t]
        out[2, t, 3] = torch.sqrt(sch["posterior_variance"][t])
        for fam in (3, 4):
            out[fam, t, 2] = sac[t]
            out[fam, t, 3] = s1m[t]
        if t > 0:
            sigma = 0
            out[1, t, 0] = sac[t - 1]
            out[1, t, 1] = torch.sqrt(1 - sac[t - 1] ** 2 - sigma ** 2)
            out[3, t, 0] = sac[t - 1]
            out[3, t, 1] = s1m[t - 1]
            sigma = (s1m[t - 1] / s1m[t]) * torch.sqrt(1 - alphas[t])
            out[4, t, 0] = sac[t - 1]
            out[4, t, 1] = torch.sqrt(1 - sac[t - 1] ** 2 - sigma ** 2)
            out[4, t, 4] = sigma
    return out
