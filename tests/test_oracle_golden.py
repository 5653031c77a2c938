"""The CPU oracle (oracle/diffroll_ref.py) against vectors produced by the reference itself
(tests/golden/*.npz, made by tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import diffroll_ref as R

# fp32 CPU vs fp32 CPU, same op order up to hoisting: tight tolerance.
ATOL = 2e-5


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: z[k] for k in z.files}


def T(a):
    return torch.from_numpy(np.asarray(a))


def params_for(g):
    hp = json.loads(str(g["hp"]))
    p = R.synthetic_params(hp, seed=int(g["seed"]))
    wsum = float(sum(v.double().abs().sum().item() for v in p.values()))
    assert abs(wsum - float(g["wsum"])) <= 1e-6 * abs(float(g["wsum"])), "synthetic weights differ from fixture"
    return hp, p


@pytest.mark.parametrize("S", [50, 200])
def test_schedule_bit_equal(golden_dir, S):
    g = load(golden_dir, f"schedule_{S}")
    sch = R.schedule(1e-4, 0.02, S)
    for k in ("betas", "alphas", "sqrt_recip_alphas", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "posterior_variance"):
        assert np.array_equal(sch[k].numpy(), g[k]), k
    assert np.array_equal(R.build_embedding(S).numpy(), g["embedding"])


def test_frontend(golden_dir):
    g = load(golden_dir, "frontend")
    hp = json.loads(str(g["hp"]))
    wav = T(g["wav"])
    Tn = int(g["T"])
    assert np.allclose(R.mel_spectrogram(wav, hp).numpy(), g["mel_raw"], rtol=1e-5, atol=1e-9)
    assert np.allclose(R.frontend(wav, hp, Tn).numpy(), g["spec"], atol=ATOL)
    assert np.allclose(R.frontend(wav, hp, Tn, inpainting_t=[4, 9]).numpy(), g["spec_t"], atol=ATOL)
    assert np.allclose(R.frontend(wav, hp, Tn, inpainting_f=[20, 100]).numpy(), g["spec_f"], atol=ATOL)
    assert np.allclose(R.frontend(wav, hp, Tn, inpainting_t=[4, 9], inpainting_f=[20, 100]).numpy(),
                       g["spec_tf"], atol=ATOL)
    assert np.allclose(R.frontend(wav, hp, Tn, sampling=True).numpy(), g["spec_u"], atol=0)
    # silence -> NaN -> 0 (model/utils.py:29-31): last sample is all zeros
    assert np.all(g["spec"][3] == 0.0)
    assert g["spec"].shape == (4, 229, Tn)


@pytest.mark.parametrize("name", ["forward_k3", "forward_k9", "forward_k15", "forward_wide_k9"])
def test_forward(golden_dir, name):
    g = load(golden_dir, name)
    hp, p = params_for(g)
    x, wav = T(g["x"]), T(g["wav"])
    t = torch.tensor(int(g["t"])).repeat(x.shape[0])
    with torch.no_grad():
        x0_c, spec = R.forward(p, hp, x, wav, t)
        x0_u, _ = R.forward(p, hp, x, torch.zeros_like(wav), t, sampling=True)
    assert np.allclose(x0_c.numpy(), g["x0_c"], atol=ATOL)
    assert np.allclose(x0_u.numpy(), g["x0_u"], atol=ATOL)
    if "x0_i" in g:
        with torch.no_grad():
            x0_i, spec_i = R.forward(p, hp, x, wav, t, inpainting_t=[10, 20])
        assert np.allclose(x0_i.numpy(), g["x0_i"], atol=ATOL)
        assert np.allclose(spec_i.numpy(), g["spec_i"], atol=ATOL)
        assert np.allclose(spec.numpy(), g["spec"], atol=ATOL)


@pytest.mark.parametrize("sampler", ["cfdg_ddpm_x0", "inpainting_ddpm_x0", "generation_ddpm_x0", "ddpm_x0"])
def test_steps_and_chain(golden_dir, sampler):
    g = load(golden_dir, "steps_chain_k9")
    hp, p = params_for(g)
    S = hp["timesteps"]
    x, wav, noise = T(g["x"]), T(g["wav"]), T(g["noise"])
    w = float(g["w"])
    it = [int(v) for v in g["inpainting_t"]] if sampler == "inpainting_ddpm_x0" else None
    sch = R.schedule(hp["beta_start"], hp["beta_end"], S)
    spec_c = R.frontend(wav, hp, x.shape[2], False, it, None)
    with torch.no_grad():
        for t_index in (S - 1, 1, 0):
            out = R.reverse_step(p, hp, sch, sampler, x, spec_c, t_index, noise[t_index], w)
            assert np.allclose(out.numpy(), g[f"{sampler}_t{t_index}"], atol=ATOL), t_index
        final = R.sample_chain(p, hp, sampler, x, wav, noise, w, inpainting_t=it)
    assert np.allclose(final.numpy(), g[f"{sampler}_chain"], atol=ATOL)


@pytest.mark.parametrize("sampler", ["ddim_x0", "cfdg_ddim_x0", "ddpm", "ddim", "ddim2ddpm"])
def test_extra_samplers_steps_and_chain(golden_dir, sampler):
    """SURVEY 8f-3 samplers against the reference's own outputs."""
    g = load(golden_dir, "steps_chain_extra_k9")
    hp, p = params_for(g)
    S = hp["timesteps"]
    x, wav, noise = T(g["x"]), T(g["wav"]), T(g["noise"])
    w = float(g["w"])
    sch = R.schedule(hp["beta_start"], hp["beta_end"], S)
    spec_c = R.frontend(wav, hp, x.shape[2])
    with torch.no_grad():
        for t_index in (S - 1, 1, 0):
            out = R.reverse_step(p, hp, sch, sampler, x, spec_c, t_index, noise[t_index], w)
            assert np.allclose(out.numpy(), g[f"{sampler}_t{t_index}"], atol=ATOL), t_index
        final = R.sample_chain(p, hp, sampler, x, wav, noise, w)
    assert np.allclose(final.numpy(), g[f"{sampler}_chain"], atol=ATOL)


def test_note_extraction_matches_reference(golden_dir):
    """integer/index work - bit exact against the reference's own function."""
    g = load(golden_dir, "notes")
    for i in range(int(g["n"])):
        for thr in (0.5, 0.8):
            p_, i_ = R.extract_notes_wo_velocity(g[f"roll{i}"], g[f"roll{i}"], thr, thr)
            assert np.array_equal(np.asarray(p_, dtype=np.int64), g[f"pitches{i}_{thr}"])
            assert np.array_equal(np.asarray(i_, dtype=np.int64).reshape(-1, 2), g[f"intervals{i}_{thr}"])


def test_q_sample_extract_x0_bit_exact(golden_dir):
    """oracle q_sample / extract_x0 == the reference's free functions (task/diffusion.py:31-64), bit for bit."""
    g = np.load(os.path.join(golden_dir, "qsample.npz"))
    T_ = lambda k: torch.from_numpy(np.asarray(g[k]))
    xt = R.q_sample(T_("x0"), T_("t"), T_("sac"), T_("s1m"), T_("noise"))
    assert torch.equal(xt, T_("xt"))
    x0b = R.extract_x0(T_("xt"), T_("eps"), T_("t"), T_("sac"), T_("s1m"))
    assert torch.equal(x0b, T_("x0_back"))


def test_trainable_spec_condition(golden_dir):
    """condition='trainable_spec' (model/diffwave.py:600-606, :656-658): oracle vs the reference run."""
    g = np.load(os.path.join(golden_dir, "trainable_spec.npz"))
    hp = json.loads(str(g["hp"]))
    p = R.synthetic_params(hp, seed=int(g["seed"]))
    assert abs(float(sum(v.double().abs().sum().item() for v in p.values())) - float(g["wsum"])) < 1e-6 * float(g["wsum"])
    T_ = lambda k: torch.from_numpy(np.asarray(g[k]))
    x, wav, z = T_("x"), T_("wav"), T_("z")
    t = torch.tensor(5).repeat(x.shape[0])
    with torch.no_grad():
        x0_u, spec_u = R.forward(p, hp, x, torch.zeros_like(wav), t, sampling=True)
        sch = R.schedule(hp["beta_start"], hp["beta_end"], hp["timesteps"])
        spec_c = R.frontend(wav, hp, x.shape[2])
        cf = R.reverse_step(p, hp, sch, "cfdg_ddpm_x0", x, spec_c, 5, z, 0.5)
        ge = R.reverse_step(p, hp, sch, "generation_ddpm_x0", x, None, 5, z, 0.0)
    assert torch.equal(spec_u[0], T_("spec_u"))          # the reference returns it 2-D (n_mels, T)
    assert float((x0_u - T_("x0_u")).abs().max()) <= 2e-5
    assert float((cf - T_("cfdg_t5")).abs().max()) <= 2e-5
    assert float((ge - T_("generation_t5")).abs().max()) <= 2e-5


def test_forward_with_per_sample_steps(golden_dir):
    """forward(x_t, waveform, diffusion_step (B,)) with different steps per sample vs the reference run."""
    g = np.load(os.path.join(golden_dir, "forward_steps.npz"))
    hp = json.loads(str(g["hp"]))
    p = R.synthetic_params(hp, seed=int(g["seed"]))
    T_ = lambda k: torch.from_numpy(np.asarray(g[k]))
    with torch.no_grad():
        x0_c, _ = R.forward(p, hp, T_("x"), T_("wav"), T_("t"))
        x0_u, _ = R.forward(p, hp, T_("x"), torch.zeros_like(T_("wav")), T_("t"), sampling=True)
    assert float((x0_c - T_("x0_c")).abs().max()) <= 2e-5
    assert float((x0_u - T_("x0_u")).abs().max()) <= 2e-5


def test_framewise_normalisation(golden_dir):
    """norm_args[2] = 'framewise' (model/utils.py:11-19) vs the reference run (random, sine and silent clips)."""
    g = np.load(os.path.join(golden_dir, "framewise.npz"))
    hp = json.loads(str(g["hp"]))
    p = R.synthetic_params(hp, seed=int(g["seed"]))
    T_ = lambda k: torch.from_numpy(np.asarray(g[k]))
    with torch.no_grad():
        x0, spec = R.forward(p, hp, T_("x"), T_("wav"), torch.tensor(3).repeat(4))
        spec_t = R.frontend(T_("wav"), hp, int(g["T"]), inpainting_t=[4, 9])
    assert float((spec - T_("spec")).abs().max()) <= 2e-5 and float((spec_t - T_("spec_t")).abs().max()) <= 2e-5
    assert float((x0 - T_("x0")).abs().max()) <= 2e-5
