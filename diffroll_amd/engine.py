"""Thin torch-tensor wrapper over the C-ABI: tensors in, raw HIP pointers out.

PyTorch is used here only for device memory and streams; every computation is a hand-written
gfx950 kernel behind libdiffroll_amd.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _cabi
from .schedule import build_embedding, make_schedule, sampler_coef_tables


class EngineError(RuntimeError):
    pass


class EngineTimeout(EngineError):
    """A group barrier of the fused residual-stack kernel ran into its spin bound (DR_ETIMEOUT): the results since
    the last finish() are invalid.  finish() has already cleared the condition and switched the engine to per-phase
    launches, so recomputing is safe; sample(check=True) does that by itself and never raises this."""


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _clamp_range(r: Optional[Sequence[int]], n: int) -> Tuple[int, int]:
    """Python slice semantics of ``spec[..., int(r[0]):int(r[1])]`` -> clamped [lo, hi) or (-1,-1)."""
    if not r:
        return -1, -1
    lo, hi, _ = slice(int(r[0]), int(r[1])).indices(n)
    if hi < lo:
        hi = lo
    return lo, hi


class Engine:
    """One engine handle per (device, stream).  Not re-entrant."""

    def __init__(self, *, residual_channels: int, residual_layers: int, kernel_size: int,
                 dilation_base: int, dilation_bound: int, n_mels: int, timesteps: int,
                 beta_start: float, beta_end: float, sample_rate: int = 16000, n_fft: int = 2048,
                 hop_length: int = 512, f_min: float = 0.0, f_max: float = 8000.0,
                 device: Optional[torch.device] = None, betas: Optional[torch.Tensor] = None,
                 norm_mode: str = "imagewise", fe_window: Optional[torch.Tensor] = None,
                 fe_fb: Optional[torch.Tensor] = None):
        """betas: optional (timesteps,) schedule replacing linspace(beta_start, beta_end) (diffroll_amd.schedule).
        fe_window (n_fft,) / fe_fb (n_fft//2+1, n_mels): the MelSpectrogram buffers of a checkpoint, used instead of
        the tables diffroll_amd.frontend_tables evaluates (which are the same expressions)."""
        if not torch.cuda.is_available():
            raise EngineError("no ROCm device visible: diffroll_amd runs only on an MI355X (no CPU fallback)")
        self.lib = _cabi.load_library()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.cfg = _cabi.DrConfig(
            abi_version=_cabi.DR_ABI_VERSION, device=self.device.index,
            residual_channels=residual_channels, residual_layers=residual_layers,
            kernel_size=kernel_size, dilation_base=dilation_base, dilation_bound=dilation_bound,
            n_mels=n_mels, timesteps=timesteps, sample_rate=sample_rate, n_fft=n_fft,
            hop_length=hop_length, f_min=f_min, f_max=f_max, beta_start=beta_start, beta_end=beta_end)
        self.timesteps = timesteps
        self.n_mels = n_mels
        self.hop_length = hop_length
        self.n_fft = n_fft
        h = C.c_void_p()
        rc = self.lib.dr_create(C.byref(h), C.byref(self.cfg))
        if rc != 0:
            raise EngineError(f"dr_create failed ({rc}): {self.lib.dr_last_error(None).decode()}")
        self.h = h
        if norm_mode not in _cabi.NORM_MODES:
            raise ValueError(f"unknown spectrogram normalisation '{norm_mode}'")
        self._check(self.lib.dr_set_spec_norm(self.h, _cabi.NORM_MODES[norm_mode]))
        self.schedule = make_schedule(beta_start, beta_end, timesteps, betas)
        self._tables = (build_embedding(timesteps).contiguous().float(),
                        sampler_coef_tables(self.schedule).contiguous())
        self._check(self.lib.dr_set_tables(
            self.h, C.cast(self._tables[0].data_ptr(), C.POINTER(C.c_float)),
            C.cast(self._tables[1].data_ptr(), C.POINTER(C.c_float))))
        # front-end constants with the reference's fp32 arithmetic (torch.hann_window, torchaudio's melscale_fbanks)
        from .frontend_tables import frontend_tables
        w, wn, fb = frontend_tables(n_fft, f_min, f_max, n_mels, sample_rate)
        if fe_window is not None:
            w = fe_window.detach().to("cpu", torch.float32).contiguous()
            wn = float(w.pow(2.0).sum().sqrt())
        if fe_fb is not None:
            fb = fe_fb.detach().to("cpu", torch.float32).contiguous()
        if tuple(w.shape) != (n_fft,) or tuple(fb.shape) != (n_fft // 2 + 1, n_mels):
            raise ValueError(f"front-end tables of shape {tuple(w.shape)} / {tuple(fb.shape)} do not fit n_fft={n_fft}, n_mels={n_mels}")
        self._fe_tables = (w, wn, fb)
        self._check(self.lib.dr_set_frontend_tables(
            self.h, C.cast(self._fe_tables[0].data_ptr(), C.POINTER(C.c_float)), C.c_float(self._fe_tables[1]),
            C.cast(self._fe_tables[2].data_ptr(), C.POINTER(C.c_float))))
        self.committed = False
        self.precision = "f32"
        self._keep = []   # tensors referenced by a captured graph must stay alive

    # ------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            msg = self.lib.dr_last_error(self.h).decode()
            if rc in (_cabi.DR_EINVAL, _cabi.DR_ENAME):
                raise ValueError(msg)
            if rc == _cabi.DR_ETIMEOUT:
                raise EngineTimeout(f"[{rc}] {msg}")
            raise EngineError(f"[{rc}] {msg}")

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def close(self):
        if getattr(self, "h", None):
            self.lib.dr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def load_params(self, params: Dict[str, torch.Tensor]):
        """Reference state_dict names/layouts (mel_layer.* buffers are ignored: the window and
        filterbank are deterministic and rebuilt inside the engine)."""
        for name, t in params.items():
            if name.startswith("mel_layer.") or name.endswith("embedding"):
                continue
            a = t.detach().to("cpu", torch.float32).contiguous()
            self._check(self.lib.dr_set_param(self.h, name.encode(),
                                              C.cast(a.data_ptr(), C.POINTER(C.c_float)), a.numel()))
        with torch.cuda.device(self.device):
            self._check(self.lib.dr_commit(self.h, self._stream()))
        self.committed = True

    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(self.device, torch.float32).contiguous()
        return t

    def frontend(self, waveform: torch.Tensor, T_roll: int, inpainting_t=None, inpainting_f=None,
                 return_spec: bool = True) -> Optional[torch.Tensor]:
        wav = self._dev(waveform)
        B, L = wav.shape
        TF = L // self.hop_length + 1
        T = min(T_roll, TF)
        t0, t1 = _clamp_range(inpainting_t, TF)
        f0, f1 = _clamp_range(inpainting_f, self.n_mels)
        spec = torch.empty(B, self.n_mels, T, device=self.device, dtype=torch.float32) if return_spec else None
        with torch.cuda.device(self.device):
            self._check(self.lib.dr_frontend(self.h, wav.data_ptr(), B, L, T_roll, t0, t1, f0, f1,
                                             _ptr(spec), self._stream()))
        return spec

    def stft_power(self, waveform: torch.Tensor) -> torch.Tensor:
        """Diagnostic (dr_debug_stft_power): (B, L) -> power spectrogram (B, L // hop + 1, n_fft // 2 + 1) of the FFT
        stage alone: reflect pad, windowed FFT, / sqrt(sum w^2), |.|^2."""
        wav = self._dev(waveform)
        B, L = wav.shape
        out = torch.empty(B, L // self.hop_length + 1, self.n_fft // 2 + 1, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._check(self.lib.dr_debug_stft_power(self.h, wav.data_ptr(), B, L, out.data_ptr(), self._stream()))
        return out

    def forward(self, x: torch.Tensor, t: int, uncond: bool) -> torch.Tensor:
        """x (B, T, 88) -> x0 (B, T, 88)."""
        x = self._dev(x)
        B, T, K = x.shape
        assert K == 88
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            self._check(self.lib.dr_forward(self.h, x.data_ptr(), B, T, int(t),
                                            _cabi.COND_UNCOND if uncond else _cabi.COND_SPEC,
                                            out.data_ptr(), self._stream()))
        return out

    def forward_steps(self, x: torch.Tensor, steps, uncond: bool) -> torch.Tensor:
        """x (B, T, 88), steps: B ints (one diffusion step per sample) -> x0 (B, T, 88)."""
        x = self._dev(x)
        B, T, K = x.shape
        assert K == 88 and len(steps) == B
        out = torch.empty_like(x)
        arr = (C.c_int32 * B)(*[int(v) for v in steps])
        with torch.cuda.device(self.device):
            self._check(self.lib.dr_forward_steps(self.h, x.data_ptr(), B, T, arr,
                                                  _cabi.COND_UNCOND if uncond else _cabi.COND_SPEC,
                                                  out.data_ptr(), self._stream()))
        return out

    def step(self, sampler: str, x: torch.Tensor, noise: Optional[torch.Tensor], t: int, w: float = 0.0,
             seed: int = 0, first_sample: int = 0) -> torch.Tensor:
        """In place on x (B, T, 88) (must already be a contiguous fp32 device tensor)."""
        assert x.device == self.device and x.dtype == torch.float32 and x.is_contiguous()
        B, T, _ = x.shape
        z = None if noise is None else self._dev(noise)
        with torch.cuda.device(self.device):
            self._check(self.lib.dr_step(self.h, _cabi.SAMPLERS[sampler], x.data_ptr(), _ptr(z), B, T, int(t),
                                         float(w), int(seed), int(first_sample), self._stream()))
        return x

    def finish(self):
        """Synchronise the current stream and verify that no fused residual-stack launch since the last check timed
        out (include/diffroll_amd.h: dr_finish).  Raises EngineTimeout when one did - the engine has then already
        been healed (per-phase launches from now on) and whatever was computed since the last check must be redone.
        Call it before a roll obtained from forward() / step() / sample(check=False) is consumed."""
        with torch.cuda.device(self.device):
            self._check(self.lib.dr_finish(self.h, self._stream()))

    def launch_state(self) -> dict:
        """dr_launch_state: how this engine launches the residual layers and what has happened to that decision -
        {'mode': 'per_phase' | 'fused_stack' | 'fused_stack+tail' | 'none', 'fused_enabled', 'fallbacks', 'yields', 'rearms',
        'stack_launches', 'tail_launches'}.  A measurement checks that fallbacks / yields did not move under it."""
        info = _cabi.DrLaunchInfo()
        self._check(self.lib.dr_launch_state(self.h, C.byref(info)))
        out = {name: int(getattr(info, name)) for name, _ in _cabi.DrLaunchInfo._fields_}
        out["mode"] = _cabi.MODES.get(out["mode"], str(out["mode"]))
        return out

    @property
    def fallbacks(self) -> int:
        """How many fused-kernel time-outs this engine has detected and healed (0 in a healthy run)."""
        return self.launch_state()["fallbacks"]

    @property
    def yields(self) -> int:
        """How many times this engine gave up fusing because another process was computing on its GPU (0 on a GPU of
        its own)."""
        return self.launch_state()["yields"]

    def cold_times(self):
        """(pack_s, upload_s, tables_s of the last dr_commit; capture + instantiate seconds and kernel-node count of the
        last captured chain) - include/diffroll_amd.h: dr_cold_times."""
        out = (C.c_double * 5)()
        self._check(self.lib.dr_cold_times(self.h, out))
        return tuple(float(v) for v in out)

    def pending_timeout(self) -> bool:
        """True when a fused launch timed out and finish() has not been called since (dr_pending_timeout; synchronises the
        current stream only if fused launches are unverified)."""
        with torch.cuda.device(self.device):
            rc = self.lib.dr_pending_timeout(self.h, self._stream())
        if rc == _cabi.DR_ETIMEOUT:
            return True
        self._check(rc)
        return False

    @property
    def tail_launches(self) -> int:
        """Tail-kernel launches issued so far (option 'fused_tail')."""
        return self.launch_state()["tail_launches"]

    def sample(self, sampler: str, x: torch.Tensor, noise: Optional[torch.Tensor], w: float = 0.0,
               seed: int = 0, first_sample: int = 0, use_graph: bool = True, check: bool = True) -> torch.Tensor:
        """Whole reverse chain in place on x (B, T, 88); noise (S, B, T, 88) or None (Philox).
        check=True (default): synchronous and self-healing - returns only with the correct roll in x (a fused launch
        that timed out because something else held the device's CUs is detected and the chain re-run on the per-phase
        kernels, dr_sample_checked).  check=False: asynchronous on the current stream; call finish() before the
        roll is consumed."""
        assert x.device == self.device and x.dtype == torch.float32 and x.is_contiguous()
        B, T, _ = x.shape
        if noise is not None:
            assert noise.device == self.device and noise.dtype == torch.float32 and noise.is_contiguous()
            assert noise.shape[0] == self.timesteps and noise.numel() == self.timesteps * x.numel()
        if use_graph:
            self._keep = [x, noise]
        with torch.cuda.device(self.device):
            if check:
                rec = C.c_int32(0)
                self._check(self.lib.dr_sample_checked(self.h, _cabi.SAMPLERS[sampler], x.data_ptr(), _ptr(noise), B, T,
                                                       float(w), int(seed), int(first_sample), 1 if use_graph else 0,
                                                       C.byref(rec), self._stream()))
            else:
                self._check(self.lib.dr_sample(self.h, _cabi.SAMPLERS[sampler], x.data_ptr(), _ptr(noise), B, T,
                                               float(w), int(seed), int(first_sample), 1 if use_graph else 0,
                                               self._stream()))
        return x

    def frame_counts(self, pred: torch.Tensor, label: torch.Tensor, threshold: float) -> Tuple[int, int, int]:
        """(TP, FP, FN) of pred > threshold vs the binary label roll (any equal shapes)."""
        p = self._dev(pred)
        l = self._dev(label)
        assert p.numel() == l.numel()
        out = (C.c_int64 * 3)()
        with torch.cuda.device(self.device):
            self._check(self.lib.dr_frame_counts(self.h, p.data_ptr(), l.data_ptr(), p.numel(), float(threshold), out,
                                                 self._stream()))
        return int(out[0]), int(out[1]), int(out[2])

    def note_runs(self, roll: torch.Tensor, threshold: float) -> torch.Tensor:
        """roll (B, T, 88) -> int32 (B, T, 88): at each note start the (exclusive) end frame, else 0."""
        r = self._dev(roll)
        B, T, K = r.shape
        assert K == 88
        out = torch.empty(B, T, 88, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.dr_note_runs(self.h, r.data_ptr(), B, T, float(threshold), out.data_ptr(),
                                              self._stream()))
        return out

    def set_precision(self, mode: str):
        """'f32' (default, exact fp32 MFMA) or 'bf16x3' (opt-in split-bf16: three bf16 pieces per operand,
        six piece products, fp32 accumulation - fp32-level error at 2.67x the matrix rate)."""
        self._check(self.lib.dr_set_precision(self.h, _cabi.PRECISIONS[mode]))
        self.precision = mode

    # ------------------------------------------------------------------ measurement helpers
    def profile_enable(self, on: bool):
        self._check(self.lib.dr_profile_enable(self.h, 1 if on else 0))

    def profile_read(self, reset: bool = True) -> Tuple[int, float]:
        n = C.c_int64(0)
        ms = C.c_double(0.0)
        self._check(self.lib.dr_profile_read(self.h, C.byref(n), C.byref(ms), 1 if reset else 0))
        return n.value, ms.value

    def profile_read_ex(self, reset: bool = True):
        """(launches, total_ms, algorithmic_flops, kernel_name) of the timed (dominant) kernel."""
        n = C.c_int64(0)
        ms = C.c_double(0.0)
        fl = C.c_double(0.0)
        buf = C.create_string_buffer(256)
        self._check(self.lib.dr_profile_read_ex(self.h, C.byref(n), C.byref(ms), C.byref(fl), buf, 256, 1 if reset else 0))
        return n.value, ms.value, fl.value, buf.value.decode()

    def set_option(self, name: str, value: int):
        """Integer options of the engine: 'fused_stack', 'fused_tail', 'fused_rearm', 'blocked_accumulation' (dr_set_option,
        include/diffroll_amd.h); any other name - 'tune.*', 'fused_stack_xcd', 'stack_ticks', ... - is a lab knob
        (dr_debug_set_option, include/diffroll_amd_debug.h).  Unknown names and values out of range raise ValueError."""
        fn = self.lib.dr_set_option if name in _cabi.PUBLIC_OPTIONS else self.lib.dr_debug_set_option
        self._check(fn(self.h, name.encode(), int(value)))

    def stack_status(self, n_ticks: int = 0):
        """(timed_out, ticks): synchronises; timed_out != 0 means a fused-kernel barrier hit its spin bound.
        self.stack_launches = fused-kernel launches issued so far."""
        flag = C.c_int32(0)
        n = C.c_int64(0)
        arr = (C.c_int64 * max(n_ticks, 1))()
        self._check(self.lib.dr_stack_status(self.h, C.byref(flag), C.byref(n), arr, int(n_ticks)))
        self.stack_launches = int(n.value)
        return int(flag.value), [int(v) for v in arr[:n_ticks]]

    def bench_pointwise(self, layer: int, NB: int, T: int):
        with torch.cuda.device(self.device):
            self._check(self.lib.dr_bench_pointwise(self.h, layer, NB, T, self._stream()))

    def debug_ticks(self) -> Tuple[int, int]:
        a = C.c_int64(0)
        b = C.c_int64(0)
        self._check(self.lib.dr_debug_ticks(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def bench_layer(self, layer: int, NB: int, T: int, t: int, n_cond: int):
        with torch.cuda.device(self.device):
            self._check(self.lib.dr_bench_layer(self.h, layer, NB, T, t, n_cond, self._stream()))
