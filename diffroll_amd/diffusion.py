"""The forward-process free functions of the reference's task/diffusion.py, on the GPU.

    linear_beta_schedule   task/diffusion.py:28-29
    q_sample               task/diffusion.py:31-46   x_t = sqrt(acp_t) x_0 + sqrt(1 - acp_t) noise
    extract_x0             task/diffusion.py:49-64   x_0 = (x_t - sqrt(1 - acp_t) eps) / sqrt(acp_t)

Same names, argument order and broadcasting ((B,) step indices against (B, 1, T, F) tensors); the arithmetic
runs in one HBM-bound HIP kernel through the C-ABI (dr_q_sample / dr_extract_x0), bit-identical to the
reference's torch expression.  No CPU fallback: tensors must live on (or are moved to) a ROCm device.
"""
import ctypes as C

import torch

from . import _cabi


def linear_beta_schedule(beta_start, beta_end, timesteps):
    return torch.linspace(beta_start, beta_end, timesteps)


def _mix(fn_name, a, b, t, sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod):
    if not torch.cuda.is_available():
        raise RuntimeError("diffroll_amd needs a ROCm GPU (no CPU fallback)")
    lib = _cabi.load_library()
    dev = a.device if a.is_cuda else torch.device("cuda", torch.cuda.current_device())
    a32 = a.to(dev, torch.float32).contiguous()
    b32 = b.to(dev, torch.float32).contiguous()
    if a32.shape != b32.shape:
        b32 = b32.expand_as(a32).contiguous()
    B = a32.shape[0]
    tt = torch.as_tensor(t).reshape(-1).to(dev, torch.int64).contiguous()
    if tt.numel() != B:
        raise ValueError(f"t has {tt.numel()} entries for a batch of {B}")
    sac = sqrt_alphas_cumprod.to(dev, torch.float32).contiguous()
    s1m = sqrt_one_minus_alphas_cumprod.to(dev, torch.float32).contiguous()
    out = torch.empty_like(a32)
    with torch.cuda.device(dev):
        rc = getattr(lib, fn_name)(None, a32.data_ptr(), b32.data_ptr(), tt.data_ptr(), sac.data_ptr(), s1m.data_ptr(),
                                   int(sac.numel()), int(B), C.c_size_t(a32.numel() // B), out.data_ptr(),
                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc:
        raise RuntimeError(lib.dr_last_error(None).decode())
    return out


def q_sample(x_start, t, sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod, noise=None):
    """task/diffusion.py:31-46.  x_start (B,1,T,F), t (B,) -> x_t.  `noise` is required, as in the reference
    (its default None fails in the multiplication)."""
    if noise is None:
        raise TypeError("q_sample needs `noise` (the reference multiplies it unconditionally)")
    return _mix("dr_q_sample", x_start, noise, t, sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod)


def extract_x0(x_t, epsilon, t, sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod):
    """task/diffusion.py:49-64: invert q_sample given the predicted noise."""
    return _mix("dr_extract_x0", x_t, epsilon, t, sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod)
