"""Import the real reference (/root/reference) in THIS container to pin the oracle.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (never on the GPU
box).  Nothing from the reference is copied: its modules are imported from where
they lie, after installing tiny stand-in modules for the third-party packages
that are absent from this image:

  * pytorch_lightning : ``LightningModule`` = nn.Module + save_hyperparameters()
    (collects the ctor arguments of every __init__ frame of the object into an
    attribute dict ``hparams``) + no-op ``log``.  Control flow only.
  * mir_eval, mido    : empty dummies (post-processing, off the path).
  * torchaudio        : ``transforms.MelSpectrogram`` forwarding to the
    restatement in oracle/diffroll_ref.py - the one piece of arithmetic that is
    third-party and un-vendored (torchaudio==0.11.0, requirements.txt:13).
    => the mel front-end is "parity unpinned" (see diffroll_ref docstring).
"""
from __future__ import annotations

import contextlib
import inspect
import io
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "model"))


class AttrDict(dict):
    """dict with attribute access, recursively (hparams.sampling.w etc.)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(obj):
    if isinstance(obj, dict):
        return AttrDict({k: to_attr(v) for k, v in obj.items()})
    return obj


def _install_stubs():
    if "pytorch_lightning" in sys.modules and getattr(sys.modules["pytorch_lightning"], "_dr_stub", False):
        return
    from . import diffroll_ref as R

    pl = types.ModuleType("pytorch_lightning")
    pl._dr_stub = True

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            hp = AttrDict()
            frame = inspect.currentframe().f_back
            while frame is not None:
                loc = frame.f_locals
                if frame.f_code.co_name == "__init__" and loc.get("self") is self:
                    info = inspect.getargvalues(frame)
                    for name in info.args:
                        if name != "self" and name not in hp:
                            hp[name] = to_attr(loc[name])
                    if info.keywords:
                        for kk, vv in loc[info.keywords].items():
                            if kk not in hp:
                                hp[kk] = to_attr(vv)
                frame = frame.f_back
            object.__setattr__(self, "hparams", hp)

        def log(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.Trainer = object
    cb = types.ModuleType("pytorch_lightning.callbacks")
    cb.LearningRateMonitor = object
    cb.ModelCheckpoint = object
    lg = types.ModuleType("pytorch_lightning.loggers")
    lg.TensorBoardLogger = object
    pl.callbacks = cb
    pl.loggers = lg
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.callbacks"] = cb
    sys.modules["pytorch_lightning.loggers"] = lg

    me = types.ModuleType("mir_eval")
    met = types.ModuleType("mir_eval.transcription")
    met.precision_recall_f1_overlap = lambda *a, **k: (0.0, 0.0, 0.0, 0.0)
    meu = types.ModuleType("mir_eval.util")
    meu.midi_to_hz = lambda x: x
    meu.hz_to_midi = lambda x: x
    me.transcription = met
    me.util = meu
    sys.modules["mir_eval"] = me
    sys.modules["mir_eval.transcription"] = met
    sys.modules["mir_eval.util"] = meu

    mido = types.ModuleType("mido")
    mido.Message = mido.MidiFile = mido.MidiTrack = object
    sys.modules["mido"] = mido

    ta = types.ModuleType("torchaudio")
    tat = types.ModuleType("torchaudio.transforms")

    class MelSpectrogram(nn.Module):
        def __init__(self, sample_rate=16000, n_fft=400, hop_length=None, n_mels=128, f_min=0.0,
                     f_max=None, center=True, normalized=False, pad_mode="reflect", **kw):
            super().__init__()
            assert center and normalized and pad_mode == "reflect"
            self.hp = dict(sample_rate=sample_rate, n_fft=n_fft, hop_length=hop_length,
                           n_mels=n_mels, f_min=f_min, f_max=f_max)
            # torchaudio 0.11 keeps its two tables as persistent buffers of two child modules, so a Lightning
            # checkpoint of the reference carries `mel_layer.spectrogram.window` / `mel_layer.mel_scale.fb`:
            # same names, same values here (forward below evaluates the same expressions itself)
            self.spectrogram = nn.Module()
            self.spectrogram.register_buffer("window", torch.hann_window(n_fft))
            self.mel_scale = nn.Module()
            self.mel_scale.register_buffer("fb", R.melscale_fbanks_htk(
                n_fft // 2 + 1, float(f_min), float(f_max if f_max is not None else sample_rate // 2),
                int(n_mels), int(sample_rate)))

        def forward(self, waveform):
            return R.mel_spectrogram(waveform, self.hp)

    tat.MelSpectrogram = MelSpectrogram
    ta.transforms = tat
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.transforms"] = tat


def import_reference_model():
    """Returns the reference's ``model`` package (model/__init__.py)."""
    if not reference_available():
        raise RuntimeError("/root/reference is not present on this machine")
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    with contextlib.redirect_stderr(io.StringIO()):
        import model as ref_model  # noqa: E402  (the reference's package)
    return ref_model


def build_reference(hp: dict, sampler: str, w: float = 0.0, inpainting_t=None, inpainting_f=None, lr: float = 1e-4):
    """Construct the reference ClassifierFreeDiffRoll (eval mode) from an oracle-style hp dict."""
    ref_model = import_reference_model()
    spec_args = to_attr(dict(sample_rate=hp["sample_rate"], n_fft=hp["n_fft"],
                             hop_length=hp["hop_length"], n_mels=hp["n_mels"], f_min=hp["f_min"],
                             f_max=hp["f_max"], center=True, normalized=True, pad_mode="reflect"))
    with contextlib.redirect_stderr(io.StringIO()):
        m = ref_model.ClassifierFreeDiffRoll(
            residual_channels=hp["residual_channels"], unconditional=False, condition=hp.get("condition", "fixed"),
            n_mels=hp["n_mels"], norm_args=[0, 1, hp.get("norm_mode", "imagewise")],
            residual_layers=hp["residual_layers"], kernel_size=hp["kernel_size"],
            dilation_base=hp["dilation_base"], dilation_bound=hp["dilation_bound"],
            spec_args=spec_args, spec_dropout=0.1, inpainting_t=inpainting_t,
            inpainting_f=inpainting_f,
            lr=lr, timesteps=hp["timesteps"], loss_type="l2", loss_keys=["diffusion_loss"],
            beta_start=hp["beta_start"], beta_end=hp["beta_end"], frame_threshold=0.5,
            training=to_attr({"mode": "x_0"}), sampling=to_attr({"type": sampler, "w": w}),
            debug=False, generation_filter=0.02)
    m.eval()
    return m


def load_params(m, params: dict):
    """Copy an oracle-style param dict into the reference module (strict on the names we own)."""
    sd = m.state_dict()
    for k, v in params.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    missing = [k for k in sd if k not in params and not k.startswith("mel_layer")]
    assert not missing, missing
    m.load_state_dict({**sd, **params})


@contextlib.contextmanager
def injected_noise(noises):
    """Replace torch.randn_like by an iterator over pre-drawn tensors (the samplers call it
    exactly once per step for t>0: task/diffusion.py:967)."""
    it = iter(noises)
    orig = torch.randn_like

    def fake(x, *a, **k):
        z = next(it)
        assert z.shape == x.shape
        return z

    torch.randn_like = fake
    try:
        yield
    finally:
        torch.randn_like = orig
