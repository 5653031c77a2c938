"""The trained regime, pinned by the reference itself.

``tests/golden/trained_small.ckpt`` is a small ClassifierFreeDiffRoll (C=64, 4 layers, k=9, 200 steps) TRAINED BY THE
REFERENCE'S OWN ``step()`` + Adam on a seeded synthetic transcription task (harmonic partials <-> their 88-key roll;
``tests/golden/make_golden.py::train_small``) and written as a Lightning-shaped checkpoint from the reference
module's own ``state_dict()`` / ``hparams`` (its ``mel_layer.*`` buffers included).  ``trained_small.npz`` holds what
the imported reference then computes on it: the roll of its ``test_step`` (200-step cfdg_ddpm_x0, w = 0.5) with the
Frame_F1 it logs, generation / inpainting chains, single evaluations.  The (S, B, 1, T, 88) noise is stored as a seed
of torch's CPU generator + checksums (18 MB otherwise); a checksum mismatch fails the test loudly.

CPU part: the oracle and the checkpoint reader against the fixture.  GPU part (-m gpu): load_from_checkpoint ->
sample / test_step on the HIP path: |d| <= 1e-5, identical thresholded roll, integer-equal TP / FP / FN.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import diffroll_ref as R

ATOL = 1e-5          # final 200-step roll, HIP vs the reference (values up to 1.85; observed: see profiles/r05_parity_margins.txt)
ATOL_ORACLE = 2e-5   # CPU oracle vs the reference (same bound as tests/test_oracle_golden.py)


def _load(golden_dir):
    z = np.load(os.path.join(golden_dir, "trained_small.npz"))
    return {k: z[k] for k in z.files}


def _noise(g):
    """x_T and the injected z's, regenerated from the stored seed exactly as make_golden.seeded_noise drew them."""
    hp = json.loads(str(g["hp"]))
    S, (B, Tn, _) = int(hp["timesteps"]), g["label"].shape
    gen = torch.Generator().manual_seed(int(g["noise_seed"]))
    x_T = torch.randn(B, 1, Tn, 88, generator=gen)
    noise = torch.randn(S, B, 1, Tn, 88, generator=gen)
    def bits(t):
        b = t.contiguous().view(torch.int32).to(torch.int64)
        return [int(b.sum()), int((b >> 9).sum())]
    if bits(x_T) != [int(v) for v in g["x_T_bits"]] or bits(noise) != [int(v) for v in g["noise_bits"]] \
            or not np.array_equal(noise[S - 1].numpy(), g["noise_last"]):
        pytest.fail("torch's CPU generator does not reproduce the fixture's noise on this machine "
                    "(trained_small.npz stores the seed, not the 18 MB of draws): the comparison would be meaningless")
    return x_T, noise


def _ckpt_path(golden_dir):
    return os.path.join(golden_dir, "trained_small.ckpt")


def _params(golden_dir, g):
    ck = torch.load(_ckpt_path(golden_dir), map_location="cpu", weights_only=False)
    p = {k: v for k, v in ck["state_dict"].items() if not k.startswith("mel_layer")}
    wsum = float(sum(v.double().abs().sum().item() for v in p.values()))
    assert wsum == float(g["wsum"]), "checkpoint and fixture were not generated together"
    return ck, p


# ------------------------------------------------------------------------------------------------ CPU
def test_checkpoint_is_in_the_trained_regime(golden_dir):
    ck, _ = _params(golden_dir, _load(golden_dir))
    curve = ck["loss_curve"]
    assert curve[0] > 5 * curve[-1], curve           # 0.0084 -> 0.0010 (mean of 100 steps)
    assert ck["hyper_parameters"]["spec_dropout"] == 0.1 and ck["hyper_parameters"]["training"]["mode"] == "x_0"
    assert float(ck["state_dict"]["output_projection.weight"].abs().max()) > 0.05     # zero-initialised (model/diffwave.py:630)


def test_oracle_on_the_trained_checkpoint_vs_reference(golden_dir):
    g = _load(golden_dir)
    hp = json.loads(str(g["hp"]))
    _, p = _params(golden_dir, g)
    x_T, noise = _noise(g)
    wav = torch.from_numpy(g["wav"])
    B = wav.shape[0]
    with torch.no_grad():
        for t in (199, 100, 0):
            xq = x_T if t == 199 else noise[t]
            tt = torch.tensor(t).repeat(B)
            c = R.forward(p, hp, xq, wav, tt)[0]
            u = R.forward(p, hp, xq, wav, tt, sampling=True)[0]
            assert float((c - torch.from_numpy(g[f"x0_c_t{t}"])).abs().max()) <= ATOL_ORACLE
            assert float((u - torch.from_numpy(g[f"x0_u_t{t}"])).abs().max()) <= ATOL_ORACLE
        roll = R.sample_chain(p, hp, "cfdg_ddpm_x0", x_T, wav, noise, w=float(g["w"]))
    ref = torch.from_numpy(g["cfdg_roll"])
    assert float((roll - ref).abs().max()) <= ATOL_ORACLE
    thr = float(g["frame_threshold"])
    assert float(g["cfdg_margin"]) > 1e-4            # nothing near the threshold: the counts below cannot flip
    pred = roll[:, 0] > thr
    lab = torch.from_numpy(g["label"]) > 0.5
    assert (int((pred & lab).sum()), int((pred & ~lab).sum()), int((~pred & lab).sum())) == (int(g["tp"]), int(g["fp"]), int(g["fn"]))
    tp, fp, fn = int(g["tp"]), int(g["fp"]), int(g["fn"])
    assert abs(2 * tp / (2 * tp + fp + fn) - float(g["frame_f1"])) < 1e-12      # what the reference's test_step logged


def test_checkpoint_reader_on_the_reference_written_checkpoint(golden_dir):
    """diffroll_amd.checkpoint + the facade's load_from_checkpoint on a checkpoint the reference module wrote: every
    tensor of the state_dict is consumed by name, the mel_layer buffers become the front-end tables and equal the
    tables the engine builds without them, keyword overrides win (sampling.py:54-65).  No GPU needed."""
    from diffroll_amd import ClassifierFreeDiffRoll
    from diffroll_amd.checkpoint import load_checkpoint
    from diffroll_amd.frontend_tables import frontend_tables
    ck = load_checkpoint(_ckpt_path(golden_dir))
    m = ClassifierFreeDiffRoll.load_from_checkpoint(_ckpt_path(golden_dir))
    own = m.state_dict()
    for k, v in ck["state_dict"].items():
        if k.startswith("mel_layer."):
            continue
        assert k in own and torch.equal(own[k], v), k
    assert set(own) == {k for k in ck["state_dict"] if not k.startswith("mel_layer.")}
    hp = m.hparams
    assert (hp.residual_channels, hp.residual_layers, hp.kernel_size, hp.timesteps) == (64, 4, 9, 200)
    assert hp.sampling.type == "cfdg_ddpm_x0" and hp.sampling.w == 0.5 and hp.spec_args.hop_length == 512
    win, _, fb = frontend_tables(2048, 0.0, 8000.0, 229, 16000)
    assert torch.equal(m.__dict__["_ckpt_window"], win) and torch.equal(m.__dict__["_ckpt_fb"], fb)
    m2 = ClassifierFreeDiffRoll.load_from_checkpoint(_ckpt_path(golden_dir), sampling={"type": "inpainting_ddpm_x0", "w": 0.5},
                                                     inpainting_t=[16, 32])
    assert m2.hparams.sampling.type == "inpainting_ddpm_x0" and m2.hparams.inpainting_t == [16, 32]


# ------------------------------------------------------------------------------------------------ GPU
def _maxdiff(a, b):
    from test_gpu_parity import maxdiff
    return maxdiff(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_trained_checkpoint_transcription_vs_reference(golden_dir, precision):
    """load_from_checkpoint -> test_step (sampling -> sample -> frame counts) on the HIP path against what the
    reference's own test_step produced on the same checkpoint, clips, x_T and noise."""
    from diffroll_amd import ClassifierFreeDiffRoll
    g = _load(golden_dir)
    x_T, noise = _noise(g)
    m = ClassifierFreeDiffRoll.load_from_checkpoint(_ckpt_path(golden_dir), precision=precision)
    wav, label = torch.from_numpy(g["wav"]), torch.from_numpy(g["label"])
    roll, spec = m.sample(x_T, wav, noise=noise)
    d = _maxdiff(roll.cpu().numpy(), g["cfdg_roll"])
    assert d <= ATOL, d
    assert _maxdiff(spec.cpu().numpy(), g["cfdg_spec"]) <= 4e-5
    thr = float(g["frame_threshold"])
    assert np.array_equal(roll.cpu().numpy() > thr, g["cfdg_roll"] > thr)          # identical thresholded roll
    out = m.test_step({"frame": label, "audio": wav, "x_T": x_T, "noise": noise}, 1)
    assert (out["tp"], out["fp"], out["fn"]) == (int(g["tp"]), int(g["fp"]), int(g["fn"]))
    assert abs(out["Test/Frame_F1"] - float(g["frame_f1"])) < 1e-12
    assert abs(out["Test/Frame_precision"] - float(g["frame_p"])) < 1e-12 and abs(out["Test/Frame_recall"] - float(g["frame_r"])) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("sampler", ["generation_ddpm_x0", "inpainting_ddpm_x0"])
def test_trained_checkpoint_generation_and_inpainting_vs_reference(golden_dir, sampler, precision):
    from diffroll_amd import ClassifierFreeDiffRoll
    g = _load(golden_dir)
    x_T, noise = _noise(g)
    it = [int(v) for v in g["inpainting_t"]] if sampler == "inpainting_ddpm_x0" else None
    m = ClassifierFreeDiffRoll.load_from_checkpoint(_ckpt_path(golden_dir), sampling={"type": sampler, "w": float(g["w"])},
                                                     inpainting_t=it, precision=precision)
    roll, _ = m.sample(x_T, torch.from_numpy(g["wav"]), noise=noise)
    ref = g[f"{sampler}_roll"]
    d = _maxdiff(roll.cpu().numpy(), ref)
    assert d <= ATOL, d
    thr = float(g["frame_threshold"])
    assert float(g[f"{sampler}_margin"]) > 1e-4
    assert np.array_equal(roll.cpu().numpy() > thr, ref > thr)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_trained_checkpoint_single_evaluations_vs_reference(golden_dir, precision):
    from diffroll_amd import ClassifierFreeDiffRoll
    g = _load(golden_dir)
    x_T, noise = _noise(g)
    m = ClassifierFreeDiffRoll.load_from_checkpoint(_ckpt_path(golden_dir), precision=precision)
    wav = torch.from_numpy(g["wav"])
    B = wav.shape[0]
    for t in (199, 100, 0):
        xq = x_T if t == 199 else noise[t]
        tt = torch.tensor(t).repeat(B)
        c, _ = m(xq, wav, tt)
        u, _ = m(xq, torch.zeros_like(wav), tt, sampling=True)
        assert _maxdiff(c.cpu().numpy(), g[f"x0_c_t{t}"]) <= ATOL
        assert _maxdiff(u.cpu().numpy(), g[f"x0_u_t{t}"]) <= ATOL
