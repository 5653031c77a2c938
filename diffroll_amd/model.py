"""Drop-in façade for the reference's sampling surface.

``ClassifierFreeDiffRoll`` keeps the constructor keywords, ``hparams`` attribute access, method
names, argument meaning, return conventions and error behaviour of the reference class
(model/diffwave.py:579-686 on top of task/diffusion.py:219-256, :513-538, :765-790, :831-853,
:943-1025), but every tensor operation runs in the HIP engine (diffroll_amd/csrc) through the
C-ABI.  The torch modules created in the constructor are only PARAMETER CONTAINERS, so that a
reference ``state_dict`` / Lightning checkpoint loads by name; they are never called.

What is deliberately different from the reference:
  * the mel front-end and the conditioner projections are computed once per clip, not twice per
    step; the step-embedding MLP is a table built at load time;
  * nothing is copied to the host inside the loop (task/diffusion.py:530) - ``predict_step`` /
    ``sampling`` return device tensors; figures / gif / MIDI side effects are not produced;
  * classifier-free guidance evaluates the conditional and unconditional branch as one 2B batch.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn

from .engine import Engine

_SAMPLERS = ("ddpm_x0", "cfdg_ddpm_x0", "generation_ddpm_x0", "inpainting_ddpm_x0",
             "ddim_x0", "cfdg_ddim_x0", "ddpm", "ddim", "ddim2ddpm")
_GUIDED = ("cfdg_ddpm_x0", "inpainting_ddpm_x0", "cfdg_ddim_x0")


class AttrDict(dict):
    """hparams-style attribute access (``self.hparams.sampling.w``)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _attr(obj):
    if isinstance(obj, dict):
        return AttrDict({k: _attr(v) for k, v in obj.items()})
    if hasattr(obj, "items") and not isinstance(obj, (str, bytes)):   # OmegaConf DictConfig & friends
        try:
            return AttrDict({k: _attr(v) for k, v in obj.items()})
        except Exception:
            return obj
    return obj


def _conv1d(cin, cout, k):
    layer = nn.Conv1d(cin, cout, k)
    nn.init.kaiming_normal_(layer.weight)        # model/diffwave.py:41-44
    return layer


class _DiffusionEmbedding(nn.Module):            # parameter container for model/diffwave.py:58-63
    def __init__(self):
        super().__init__()
        self.projection1 = nn.Linear(128, 512)
        self.projection2 = nn.Linear(512, 512)


class _ResidualBlock(nn.Module):                 # parameter container for model/diffwave.py:108-132
    def __init__(self, n_mels, residual_channels, kernel_size):
        super().__init__()
        self.dilated_conv = _conv1d(residual_channels, 2 * residual_channels, kernel_size)
        self.diffusion_projection = nn.Linear(512, residual_channels)
        self.conditioner_projection = _conv1d(n_mels, 2 * residual_channels, 1)
        self.output_projection = _conv1d(residual_channels, 2 * residual_channels, 1)


class ClassifierFreeDiffRoll(nn.Module):
    def __init__(self,
                 residual_channels,
                 unconditional,
                 condition,
                 n_mels,
                 norm_args,
                 residual_layers=30,
                 kernel_size=3,
                 dilation_base=1,
                 dilation_bound=4,
                 spec_args={},
                 spec_dropout=0.5,
                 inpainting_t=None,
                 inpainting_f=None,
                 # SpecRollDiffusion (task/diffusion.py:220-232)
                 lr=1e-4,
                 timesteps=200,
                 loss_type="l2",
                 loss_keys=("diffusion_loss",),
                 beta_start=1e-4,
                 beta_end=0.02,
                 frame_threshold=0.5,
                 training=None,
                 sampling=None,
                 debug=False,
                 generation_filter=0.0,
                 device=None,
                 precision="f32",
                 beta_schedule="linear",
                 accumulation="auto"):
        super().__init__()
        if condition not in ("fixed", "trainable_spec"):
            if condition == "trainable_z":
                raise NotImplementedError(
                    "condition='trainable_z' cannot be constructed in the reference either (model/diffwave.py:619 "
                    "passes kernel_size into ResidualBlockz's `uncond`; SURVEY.md Appendix B)")
            raise ValueError(f"unrecognized condition '{condition}'")        # model/diffwave.py:610
        if unconditional:
            raise NotImplementedError("unconditional=True cannot run in the reference either: forward() still passes the "
                                      "spectrogram to blocks built without a conditioner and trips the assertion at "
                                      "model/diffwave.py:135-136")
        if len(norm_args) < 3 or norm_args[2] not in ("imagewise", "framewise"):
            raise ValueError(f"norm_args[2] must be 'imagewise' or 'framewise' (model/utils.py:10-35), got {norm_args!r}")
        sampling = _attr(sampling if sampling is not None else {"type": "cfdg_ddpm_x0", "w": 0.0})
        training = _attr(training if training is not None else {"mode": "x_0"})
        spec_args = _attr(dict(spec_args))
        if sampling.type not in _SAMPLERS:
            raise AttributeError(sampling.type)                               # getattr at task/diffusion.py:255
        self.hparams = AttrDict(
            residual_channels=residual_channels, unconditional=unconditional, condition=condition,
            n_mels=n_mels, norm_args=list(norm_args), residual_layers=residual_layers,
            kernel_size=kernel_size, dilation_base=dilation_base, dilation_bound=dilation_bound,
            spec_args=spec_args, spec_dropout=spec_dropout, inpainting_t=inpainting_t,
            inpainting_f=inpainting_f, lr=lr, timesteps=timesteps, loss_type=loss_type,
            loss_keys=list(loss_keys), beta_start=beta_start, beta_end=beta_end,
            frame_threshold=frame_threshold, training=training, sampling=sampling, debug=debug,
            generation_filter=generation_filter, beta_schedule=beta_schedule)
        if beta_schedule not in ("linear", "cosine", "quadratic", "sigmoid"):
            raise ValueError(f"unknown beta_schedule '{beta_schedule}'")
        self.spec_dropout = spec_dropout

        if condition == "trainable_spec":                                    # model/diffwave.py:600-604
            self.trainable_parameters = nn.Parameter(torch.full((int(spec_args.get("n_mels", n_mels)), 641), -1.0))
        # parameter containers, same names/shapes/initialisation as the reference
        self.input_projection = _conv1d(88, residual_channels, 1)
        self.diffusion_embedding = _DiffusionEmbedding()
        self.residual_layers = nn.ModuleList(
            [_ResidualBlock(n_mels, residual_channels, kernel_size) for _ in range(residual_layers)])
        self.skip_projection = _conv1d(residual_channels, residual_channels, 1)
        self.output_projection = _conv1d(residual_channels, 88, 1)
        nn.init.zeros_(self.output_projection.weight)                         # model/diffwave.py:630
        for p in self.parameters():
            p.requires_grad_(False)

        sa = spec_args
        self._engine_kwargs = dict(
            residual_channels=residual_channels, residual_layers=residual_layers,
            kernel_size=kernel_size, dilation_base=dilation_base, dilation_bound=dilation_bound,
            n_mels=n_mels, timesteps=timesteps, beta_start=beta_start, beta_end=beta_end,
            sample_rate=int(sa.get("sample_rate", 16000)), n_fft=int(sa.get("n_fft", 2048)),
            hop_length=int(sa.get("hop_length", 512)), f_min=float(sa.get("f_min", 0.0)),
            f_max=float(sa.get("f_max", 8000.0)))
        # The front-end kernels implement config/spec/mel.yaml: center=True and normalized=True.  torchaudio's own
        # default for `normalized` is False, so a spec_args without the key would build a DIFFERENT front-end in the
        # reference (log(x + 1e-6) then min-max does not cancel the window scale): the key must be present and true.
        if not sa.get("center", True):
            raise NotImplementedError("spec_args.center=False is not supported (config/spec/mel.yaml)")
        if "normalized" not in sa:
            raise NotImplementedError("spec_args.normalized is required and must be True (config/spec/mel.yaml:10): "
                                      "torchaudio's default is False, which this front-end does not implement")
        if not sa["normalized"]:
            raise NotImplementedError("spec_args.normalized=False is not supported (config/spec/mel.yaml)")
        # the remaining keys default to config/spec/mel.yaml's values (n_fft 2048, hop 512, f_max 8000, sr 16000),
        # NOT to torchaudio's (400 / 200 / sr/2): the engine is built for the released configuration
        if sa.get("pad_mode", "reflect") != "reflect":
            raise NotImplementedError("only pad_mode='reflect' is supported (config/spec/mel.yaml)")
        # every other torchaudio MelSpectrogram argument must be at the value the front-end kernels implement
        # (torchaudio 0.11 defaults) - nothing is silently ignored
        fixed = {"win_length": (None, int(sa.get("n_fft", 2048))), "power": (2.0, 2), "mel_scale": ("htk",),
                 "norm": (None,), "onesided": (True,), "pad": (0,), "window_fn": (torch.hann_window,),
                 "wkwargs": (None,)}
        known = {"sample_rate", "n_fft", "hop_length", "n_mels", "f_min", "f_max", "center", "normalized", "pad_mode"}
        for key, val in sa.items():
            if key in known:
                continue
            if key not in fixed:
                raise TypeError(f"spec_args: unknown MelSpectrogram argument '{key}'")
            if val not in fixed[key]:
                raise NotImplementedError(f"spec_args.{key}={val!r} is not supported (front-end implements {fixed[key][0]!r})")
        if int(sa.get("n_mels", n_mels)) != int(n_mels):
            raise ValueError(f"spec_args.n_mels={sa.get('n_mels')} differs from n_mels={n_mels} (the conditioner's input width)")
        self._device = device
        self.precision = precision          # 'f32' (exact, default) | 'bf16x3' (opt-in split precision)
        # accumulation order of the dilated conv (an extension): 'auto' = 'blocked' = one fp32 chain per
        # 32-channel chunk, chunk sums added up separately - like a CPU library's K-blocked GEMM - in every flavour;
        # 'single_chain' = 128-frame blocks (16 guided clips per GPU) and the 96 / 160-frame flavours (640-frame rolls) contract
        # all of K as one chain, the rounds 1-3 numerics: 0.2-0.8 % faster, 2-3x the rounding error against float64
        if accumulation not in ("auto", "blocked", "single_chain"):
            raise ValueError("accumulation is 'auto', 'blocked' or 'single_chain'")
        self.accumulation = accumulation
        self._engine: Optional[Engine] = None
        self._dirty = True
        self._fe_key = None
        self._fe_spec = None
        self.reverse_diffusion = getattr(self, sampling.type)                  # task/diffusion.py:255

    # ------------------------------------------------------------------ plumbing
    def _betas(self):
        """None for the reference's linear schedule (task/diffusion.py:239), else one of the model/unet.py:558-579
        schedules (an extension: `beta_schedule=` is not a reference kwarg)."""
        from . import schedule as S
        kind = self.__dict__["hparams"].get("beta_schedule", "linear")
        if kind == "linear":
            return None
        return {"cosine": S.cosine_beta_schedule, "quadratic": S.quadratic_beta_schedule,
                "sigmoid": S.sigmoid_beta_schedule}[kind](self.__dict__["hparams"].timesteps)

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            self._engine = Engine(device=self._device, betas=self._betas(), norm_mode=str(self.hparams.norm_args[2]),
                                  fe_window=self.__dict__.get("_ckpt_window"), fe_fb=self.__dict__.get("_ckpt_fb"),
                                  **self._engine_kwargs)
            self._dirty = True
        if self._dirty:
            self._engine.load_params({k: v for k, v in self.state_dict().items()})
            self._dirty = False
            self._fe_key = None
        if self._engine.precision != self.precision:
            self._engine.set_precision(self.precision)
        want = 1 if self.accumulation == "single_chain" else 2
        if getattr(self._engine, "_blocked", None) != want:
            self._engine.set_option("blocked_accumulation", want)
            self._engine._blocked = want
        return self._engine

    # schedule vectors, exposed like the reference's attributes (task/diffusion.py:239-256)
    def __getattr__(self, name):
        if name in ("betas", "alphas", "sqrt_recip_alphas", "sqrt_alphas_cumprod",
                    "sqrt_one_minus_alphas_cumprod", "posterior_variance"):
            from .schedule import make_schedule
            hp = self.__dict__["hparams"]
            return make_schedule(hp.beta_start, hp.beta_end, hp.timesteps, self._betas())[name]
        return super().__getattr__(name)

    def load_state_dict(self, state_dict, strict: bool = True):
        own = {k: v for k, v in state_dict.items()
               if not k.startswith("mel_layer.") and k != "diffusion_embedding.embedding"}
        out = super().load_state_dict(own, strict=strict)
        # the MelSpectrogram buffers of a reference checkpoint (torchaudio 0.11 names) are the front-end tables
        # themselves: use them (they equal diffroll_amd.frontend_tables' output, which is what runs without them)
        win, fb = state_dict.get("mel_layer.spectrogram.window"), state_dict.get("mel_layer.mel_scale.fb")
        n_fft, n_mels = self._engine_kwargs["n_fft"], self._engine_kwargs["n_mels"]
        changed = False
        if win is not None and tuple(win.shape) == (n_fft,):
            self.__dict__["_ckpt_window"] = win.detach().float().cpu()
            changed = True
        if fb is not None and tuple(fb.shape) == (n_fft // 2 + 1, n_mels):
            self.__dict__["_ckpt_fb"] = fb.detach().float().cpu()
            changed = True
        if changed and self._engine is not None:          # tables are fixed at engine creation: rebuild lazily
            self._engine.close()
            self._engine = None
        self._dirty = True
        return out

    def to(self, *args, **kwargs):           # parameters stay on the host; the engine owns device copies
        for a in args:
            if isinstance(a, (str, torch.device)) and torch.device(a).type == "cuda":
                self._device = torch.device(a)
        if "device" in kwargs and torch.device(kwargs["device"]).type == "cuda":
            self._device = torch.device(kwargs["device"])
        return self

    def cuda(self, device=None):
        self._device = torch.device("cuda", device if device is not None else torch.cuda.current_device())
        return self

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, trust=False, **overrides):
        """Lightning-style: ``{'state_dict', 'hyper_parameters'}``; keyword overrides win
        (sampling.py:54-65).  OmegaConf containers inside a real reference checkpoint are read without
        omegaconf / pytorch_lightning installed, and without executing anything the file names: an allow-listing
        unpickler (diffroll_amd/checkpoint.py); ``trust=True`` = the full unpickle Lightning itself does."""
        from .checkpoint import constructor_kwargs, load_checkpoint
        ckpt = load_checkpoint(checkpoint_path, trust=trust)
        m = cls(**constructor_kwargs(cls, ckpt["hyper_parameters"], overrides))
        m.load_state_dict(ckpt["state_dict"], strict=False)
        return m

    # ------------------------------------------------------------------ forward
    def _frontend(self, waveform: torch.Tensor, T_roll: int, inpainting_t, inpainting_f) -> torch.Tensor:
        eng = self.engine
        try:
            version = waveform._version
        except Exception:                      # inference-mode tensors do not track a version: never cached
            version = None
        key = (waveform.data_ptr(), tuple(waveform.shape), version, T_roll,
               tuple(inpainting_t) if inpainting_t else None, tuple(inpainting_f) if inpainting_f else None)
        if version is None or key != self._fe_key:
            self._fe_spec = eng.frontend(waveform, T_roll, inpainting_t, inpainting_f)
            self._fe_key = key
            self._fe_wave = waveform    # keep alive so data_ptr cannot be recycled
        return self._fe_spec

    def forward(self, x_t, waveform, diffusion_step, sampling=False, inpainting_t=None, inpainting_f=None):
        """(x_t (B,1,T,88), waveform (B,L), diffusion_step (B,) int) -> (x0_pred (B,1,T,88), spec (B,n_mels,T'))
        with T' = min(T, L//hop + 1) (model/diffwave.py:637-686, eval mode)."""
        eng = self.engine
        if diffusion_step.dtype not in (torch.int32, torch.int64):
            raise NotImplementedError("fractional diffusion steps (lerp branch, model/diffwave.py:76-81) are off-path")
        steps = [int(v) for v in diffusion_step.flatten().tolist()]
        B, _, T, K = x_t.shape
        if len(steps) != B:
            raise ValueError(f"diffusion_step has {len(steps)} entries for a batch of {B}")
        t = steps[0]
        uniform = all(v == t for v in steps)      # the samplers' case (task/diffusion.py:947)
        if sampling is True and self.hparams.condition == "trainable_spec":
            # the learned unconditional spectrogram replaces the clip's (model/diffwave.py:656-658); it is 2-D and
            # 641 frames long, and trim_spec_roll (:662) trims the roll to it
            Tm = min(T, 641)
            spec = self.trainable_parameters.detach()[:, :Tm].to(eng.device, torch.float32)
        elif sampling is True:
            TF = waveform.shape[-1] // eng.hop_length + 1
            Tm = min(T, TF)
            spec = torch.full((B, eng.n_mels, Tm), -1.0, device=eng.device)
        else:
            spec = self._frontend(waveform, T, inpainting_t, inpainting_f)
            Tm = spec.shape[-1]
        x = x_t.to(eng.device, torch.float32).squeeze(1)[:, :Tm, :].contiguous()
        if uniform:
            x0 = self._verified(lambda: eng.forward(x, t, uncond=(sampling is True)))
        else:                                      # one step per sample, as the reference's step() calls forward
            x0 = self._verified(lambda: eng.forward_steps(x, steps, uncond=(sampling is True)))
        return x0.unsqueeze(1), spec

    def _verified(self, fn):
        """Run fn (engine launches returning a result tensor) and hand the result out only after engine.finish() has
        confirmed that no fused launch timed out; after a time-out (healed by finish(): per-phase launches from then
        on) it is recomputed once.  The reference's methods return finished tensors - never silently invalid ones."""
        from .engine import EngineTimeout
        eng = self.engine
        try:
            out = fn()
            eng.finish()
            return out
        except EngineTimeout:
            try:                   # (a time-out left pending by earlier unchecked calls is cleared here)
                eng.finish()
            except EngineTimeout:
                pass
            out = fn()
            eng.finish()
            return out

    # ------------------------------------------------------------------ samplers (one step)
    def _one_step(self, sampler: str, x, waveform, t_index: int, noise=None):
        eng = self.engine
        B, _, T, _ = x.shape
        spec = None
        if sampler != "generation_ddpm_x0":
            it = self.hparams.inpainting_t if sampler == "inpainting_ddpm_x0" else None
            i_f = self.hparams.inpainting_f if sampler == "inpainting_ddpm_x0" else None
            spec = self._frontend(waveform, T, it, i_f)
            Tm = spec.shape[-1]
        else:
            Tm = min(T, waveform.shape[-1] // eng.hop_length + 1) if waveform is not None else T
            if self.hparams.condition == "trainable_spec":
                Tm = min(T, 641)
        x_in = x.to(eng.device, torch.float32).squeeze(1)[:, :Tm, :].contiguous()
        w = float(self.hparams.sampling.get("w", 0.0)) if sampler in _GUIDED else 0.0
        z = None
        if noise is not None:
            z = noise.to(eng.device, torch.float32).reshape(B, Tm, 88).contiguous()
        elif t_index > 0:
            z = torch.randn(B, Tm, 88, device=eng.device)   # reference: torch.randn_like(x), global generator
        xx = self._verified(lambda: eng.step(sampler, x_in.clone(), z, t_index, w))
        if spec is None:
            spec = self._uncond_spec(B, Tm)
        return xx.unsqueeze(1), spec

    def _uncond_spec(self, B, Tm):
        """The spectrogram generation_ddpm_x0 returns (task/diffusion.py:979-997: that of its sampling=True forward):
        -1 everywhere, or the learned one under condition='trainable_spec' (model/diffwave.py:656-660)."""
        eng = self.engine
        if self.hparams.condition == "trainable_spec":
            return self.trainable_parameters.detach()[:, :Tm].to(eng.device, torch.float32)
        return torch.full((B, eng.n_mels, Tm), -1.0, device=eng.device)

    def ddpm_x0(self, x, waveform, t_index, noise=None):
        """task/diffusion.py:831-853."""
        return self._one_step("ddpm_x0", x, waveform, t_index, noise)

    def cfdg_ddpm_x0(self, x, waveform, t_index, noise=None):
        """task/diffusion.py:943-969."""
        return self._one_step("cfdg_ddpm_x0", x, waveform, t_index, noise)

    def generation_ddpm_x0(self, x, waveform, t_index, noise=None):
        """task/diffusion.py:971-997."""
        return self._one_step("generation_ddpm_x0", x, waveform, t_index, noise)

    def inpainting_ddpm_x0(self, x, waveform, t_index, noise=None):
        """task/diffusion.py:999-1025."""
        return self._one_step("inpainting_ddpm_x0", x, waveform, t_index, noise)

    # same kernels, other per-step coefficients
    def ddim_x0(self, x, waveform, t_index, noise=None):
        """task/diffusion.py:855-875 (sigma = 0; `noise` is accepted and ignored, as 0 * randn_like)."""
        return self._one_step("ddim_x0", x, waveform, t_index, noise)

    def cfdg_ddim_x0(self, x, waveform, t_index, noise=None):
        """task/diffusion.py:1027-1055 (second branch: spectrogram of a zero waveform = all 0, not -1)."""
        return self._one_step("cfdg_ddim_x0", x, waveform, t_index, noise)

    def ddpm(self, x, waveform, t_index, noise=None):
        """task/diffusion.py:804-829 (network output interpreted as epsilon)."""
        return self._one_step("ddpm", x, waveform, t_index, noise)

    def ddim(self, x, waveform, t_index, noise=None):
        """task/diffusion.py:877-892 (epsilon prediction, deterministic)."""
        return self._one_step("ddim", x, waveform, t_index, noise)

    def ddim2ddpm(self, x, waveform, t_index, noise=None):
        """task/diffusion.py:894-911 (epsilon prediction)."""
        return self._one_step("ddim2ddpm", x, waveform, t_index, noise)

    # ------------------------------------------------------------------ whole chain
    def output_frames(self, T: int, waveform_samples: Optional[int]) -> int:
        """Frames of the roll sample() returns for a T-frame x_T (trim_spec_roll, model/diffwave.py:30-39, :662): the
        spectrogram's length when that is shorter - the clip's L // hop + 1, or the 641 frames of the learned
        unconditional spectrogram under condition='trainable_spec' for generation."""
        sampler = self.hparams.sampling.type
        hop = self._engine_kwargs["hop_length"]
        if sampler != "generation_ddpm_x0":
            return min(T, waveform_samples // hop + 1)
        Tm = T if waveform_samples is None else min(T, waveform_samples // hop + 1)
        if self.hparams.condition == "trainable_spec":
            Tm = min(T, 641)
        return Tm

    @torch.no_grad()
    def sample(self, x_T, waveform=None, noise=None, seed: int = 0, first_sample: int = 0,
               use_graph: bool = True, check: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        """The reverse chain t = timesteps-1 .. 0 (task/diffusion.py:528-534) on the device with no
        host round trip.  x_T (B,1,T,88); noise: None (on-device Philox keyed by seed and global
        sample index) or (timesteps, B, 1, T, 88) injected z's (row t is used at step t >= 1).
        Returns (roll (B,1,T',88), spec (B,n_mels,T')).
        check=True (default): the call returns with the FINISHED, verified roll, as task/diffusion.py:528-538 does
        (synchronous; a fused-kernel time-out caused by another tenant of the device is healed by re-running the chain
        on the per-phase kernels - Engine.sample).  check=False: asynchronous; call engine.finish() before use."""
        eng = self.engine
        sampler = self.hparams.sampling.type
        B, _, T, _ = x_T.shape
        if sampler != "generation_ddpm_x0":
            if waveform is None:
                raise ValueError("waveform is required for conditional samplers")
            it = self.hparams.inpainting_t if sampler == "inpainting_ddpm_x0" else None
            i_f = self.hparams.inpainting_f if sampler == "inpainting_ddpm_x0" else None
            spec = self._frontend(waveform, T, it, i_f)
            Tm = spec.shape[-1]
        else:
            Tm = T if waveform is None else min(T, waveform.shape[-1] // eng.hop_length + 1)
            if self.hparams.condition == "trainable_spec":
                Tm = min(T, 641)
            spec = self._uncond_spec(B, Tm)
        # a fresh roll buffer per call: the engine's captured chain runs on its own work buffer, so caller
        # addresses never force a re-capture
        xb = x_T.to(eng.device, torch.float32).squeeze(1)[:, :Tm, :].clone(memory_format=torch.contiguous_format)
        z = None
        if noise is not None:
            S = self.hparams.timesteps
            z = noise.to(eng.device, torch.float32).reshape(S, B, T, 88)
            if Tm != T or not z.is_contiguous():
                z = z[:, :, :Tm, :].contiguous()
        w = float(self.hparams.sampling.get("w", 0.0)) if sampler in _GUIDED else 0.0
        eng.sample(sampler, xb, z, w, seed, first_sample, use_graph, check)
        return xb.unsqueeze(1), spec

    def sample_trajectory(self, x_T, waveform=None, noise=None, seed: int = 0, first_sample: int = 0):
        """The same chain, keeping every intermediate roll on the device: returns (trajectory (timesteps, B, 1,
        T', 88) with row i = x after step t = timesteps-1-i, spec).  This is what the reference's sampling()
        collects as `noise_list` - on the host, with one D2H copy per step (task/diffusion.py:779-788) - for its
        animation; here it is an opt-in eager loop over dr_step (one launch sequence per step, no graph), and the
        last row equals sample()'s result bit for bit."""
        eng = self.engine
        sampler = self.hparams.sampling.type
        S = int(self.hparams.timesteps)
        B = x_T.shape[0]
        x = x_T
        rows = []
        spec = None
        for t in range(S - 1, -1, -1):
            z = None if noise is None else noise[t]
            if z is None and t > 0:       # Philox keyed by (seed, global sample, step): same draws as sample()
                xx, spec = self._step_philox(sampler, x, waveform, t, seed, first_sample)
            else:
                xx, spec = self._one_step(sampler, x, waveform, t, z if t > 0 else torch.zeros_like(x))
            rows.append(xx)
            x = xx
        return torch.stack(rows, 0), spec

    def _step_philox(self, sampler, x, waveform, t_index, seed, first_sample):
        eng = self.engine
        B, _, T, _ = x.shape
        spec = None
        if sampler != "generation_ddpm_x0":
            it = self.hparams.inpainting_t if sampler == "inpainting_ddpm_x0" else None
            i_f = self.hparams.inpainting_f if sampler == "inpainting_ddpm_x0" else None
            spec = self._frontend(waveform, T, it, i_f)
            Tm = spec.shape[-1]
        else:
            Tm = min(T, waveform.shape[-1] // eng.hop_length + 1) if waveform is not None else T
            if self.hparams.condition == "trainable_spec":
                Tm = min(T, 641)
        x_in = x.to(eng.device, torch.float32).squeeze(1)[:, :Tm, :].contiguous()
        w = float(self.hparams.sampling.get("w", 0.0)) if sampler in _GUIDED else 0.0
        xx = self._verified(lambda: eng.step(sampler, x_in.clone(), None, t_index, w, seed, first_sample))
        if spec is None:
            spec = self._uncond_spec(B, Tm)
        return xx.unsqueeze(1), spec

    def predict_step(self, batch, batch_idx=0):
        """batch = (x_T, waveform[, ...]) as built by sampling.py:27-46.  Returns the final roll
        (B,1,T,88) (the reference returns nothing and writes figures/MIDI instead)."""
        noise, waveform = batch[0], batch[1]
        roll, _ = self.sample(noise, waveform, seed=batch_idx)
        return roll

    @staticmethod
    def frame_metrics(tp: int, fp: int, fn: int) -> Tuple[float, float, float]:
        """precision / recall / F1 from the confusion counts with sklearn's average='binary'
        conventions (0 where a denominator is 0)."""
        p = tp / (tp + fp) if tp + fp > 0 else 0.0
        r = tp / (tp + fn) if tp + fn > 0 else 0.0
        f = 2 * p * r / (p + r) if p + r > 0 else 0.0
        return p, r, f

    def test_step(self, batch, batch_idx=0):
        """Frame-level part of task/diffusion.py:312-428: sample the batch, threshold the final roll at
        hparams.frame_threshold and score it against batch['frame'] exactly as
        sklearn.metrics.precision_recall_fscore_support(label.flatten(), pred.flatten() > thr,
        average='binary') does (:381-383); then the note-level score of :385-410 - notes extracted from the
        prediction and from the label roll (GPU run-length scan), onset-only matching as mir_eval's
        precision_recall_f1_overlap(offset_ratio=None) (diffroll_amd/metrics.py; parity unpinned: mir_eval is
        absent here).  Returns the metrics the reference logs; Test/Note_F1 is the mean over the batch (the
        reference logs it for the samples of batch 0 only, :426)."""
        from . import midi, metrics
        roll, _ = self.sampling(batch, batch_idx)
        label = batch["frame"]
        Tm = roll.shape[2]
        label_dev = label[:, :Tm].to(roll.device, torch.float32).contiguous()
        thr = float(self.hparams.frame_threshold)
        tp, fp, fn = self.engine.frame_counts(roll[:, 0], label_dev, thr)
        p, r, f = self.frame_metrics(tp, fp, fn)
        sa = self.hparams.spec_args
        est = midi.extract_notes_wo_velocity(self.engine, roll, thr)
        ref = midi.extract_notes_wo_velocity(self.engine, label_dev, thr)
        notes = metrics.note_scores(ref, est, int(sa.get("hop_length", 512)), int(sa.get("sample_rate", 16000)))
        note_f1 = float(sum(n[2] for n in notes) / max(len(notes), 1))
        return {"Test/Frame_F1": f, "Test/Frame_precision": p, "Test/Frame_recall": r, "tp": tp, "fp": fp, "fn": fn,
                "Test/Note_F1": note_f1, "note_scores": notes}

    def export_midi(self, roll, path_prefix="raw_midi_", threshold=0.5, reference_timing=False, clean_prefix=None):
        """Post-processing of predict_step (task/diffusion.py:598-618): threshold the final roll (the
        reference uses the function default 0.5 there, not hparams.frame_threshold), extract notes on the
        GPU and write per sample `<path_prefix><i>.mid` (all notes: the reference's raw_midi_{batch}_{i}.mid) and,
        with clean_prefix, `<clean_prefix><i>.mid` without the notes not longer than hparams.generation_filter
        seconds (its clean_midi_e{batch}_{i}.mid).  Note times use the model's hop (512 / 16000 s per frame);
        reference_timing=True reproduces the reference's predict_step instead, which scales by its stale module
        constant HOP_LENGTH = 160 (task/diffusion.py:19,604: every time 3.2x too short, and the duration filter
        applied on that scale)."""
        from . import midi
        sa = self.hparams.spec_args
        hop = 160 if reference_timing else int(sa.get("hop_length", 512))
        return midi.export_midi(self.engine, roll, path_prefix, threshold, hop, int(sa.get("sample_rate", 16000)),
                                float(self.hparams.generation_filter), clean_prefix)

    def sampling(self, batch, batch_idx=0):
        """task/diffusion.py:765-790 with x_T drawn on the device; returns (roll, spec).  Two optional batch entries
        (an extension; the reference draws both from torch's global generator, :775 and :967) make a run repeatable
        against the reference on identical inputs: 'x_T' (B, 1, T, 88) and 'noise' (timesteps, B, 1, T, 88)."""
        if self.hparams.debug:
            raise NotImplementedError("debug=True feeds the label roll where the waveform belongs "
                                      "(task/diffusion.py:780-781): a development switch of the reference, not a mode")
        frame = batch["frame"]
        x_T = batch.get("x_T")
        if x_T is None:
            x_T = torch.randn(frame.shape[0], 1, frame.shape[1], frame.shape[2], device=self.engine.device)
        return self.sample(x_T, batch["audio"], noise=batch.get("noise"), seed=batch_idx)
