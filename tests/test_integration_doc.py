"""INTEGRATION.md section 2 is a program: the reference-side ctypes binding a maintainer would paste into
model/diffwave.py.  This test extracts that Python block VERBATIM, executes it in a child process against a stand-in
for the reference's host class (what the block reads: state_dict(), hparams, the schedule vectors, diffusion_embedding
.embedding, the mel_layer buffers - the package's Engine class and library loader are never used in that process), runs its
predict_step, and compares the roll bit for bit with the facade's predict_step on the same inputs (same Philox seed).
A header change that breaks the documented binding fails here."""
import os
import re
import subprocess
import sys
import textwrap

import pytest
import torch

from oracle import diffroll_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def doc_block():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. Binding the C-ABI"):text.index("## 3. Multi-GPU")]
    blocks = re.findall(r"```python\n(.*?)```", sec, re.S)
    assert len(blocks) == 1, "INTEGRATION.md section 2 must hold exactly one python block"
    return blocks[0]


HOST = r'''
import json, os, sys
import torch
sys.path.insert(0, ROOT)
from oracle import diffroll_ref as R                      # weights / tables of the stand-in only
from diffroll_amd.schedule import make_schedule, build_embedding
from diffroll_amd.frontend_tables import frontend_tables

class AttrDict(dict):
    __getattr__ = dict.__getitem__

HP = json.loads(HP_JSON)

class _Holder:                                            # attribute container (diffusion_embedding, mel_layer.*)
    pass

class SpecRollDiffusion:
    """Stand-in for task/diffusion.py::SpecRollDiffusion + the ctor of model/diffwave.py::ClassifierFreeDiffRoll: what
    the documented binding reads from `self`."""
    def __init__(self):
        hp = HP
        self.hparams = AttrDict(
            residual_channels=hp["residual_channels"], residual_layers=hp["residual_layers"], kernel_size=hp["kernel_size"],
            dilation_base=hp["dilation_base"], dilation_bound=hp["dilation_bound"], n_mels=hp["n_mels"],
            timesteps=hp["timesteps"], beta_start=hp["beta_start"], beta_end=hp["beta_end"], inpainting_t=None, inpainting_f=None,
            spec_args=AttrDict(sample_rate=hp["sample_rate"], n_fft=hp["n_fft"], hop_length=hp["hop_length"],
                               f_min=hp["f_min"], f_max=hp["f_max"]),
            sampling=AttrDict(type="cfdg_ddpm_x0", w=0.5))
        self._params = R.synthetic_params(hp, seed=SEED)
        for k, v in make_schedule(hp["beta_start"], hp["beta_end"], hp["timesteps"]).items():
            setattr(self, k, v)                           # betas, alphas, sqrt_recip_alphas, ... (task/diffusion.py:239-256)
        self.diffusion_embedding = _Holder()
        self.diffusion_embedding.embedding = build_embedding(hp["timesteps"])
        win, _, fb = frontend_tables(hp["n_fft"], hp["f_min"], hp["f_max"], hp["n_mels"], hp["sample_rate"])
        self.mel_layer = _Holder(); self.mel_layer.spectrogram = _Holder(); self.mel_layer.mel_scale = _Holder()
        self.mel_layer.spectrogram.window = win
        self.mel_layer.mel_scale.fb = fb

    def state_dict(self):
        sd = dict(self._params)
        sd["diffusion_embedding.embedding"] = self.diffusion_embedding.embedding      # (non-persistent in the reference; skipped by name)
        sd["mel_layer.spectrogram.window"] = self.mel_layer.spectrogram.window
        sd["mel_layer.mel_scale.fb"] = self.mel_layer.mel_scale.fb
        return sd
'''

DRIVER = r'''
from diffroll_amd import _cabi
assert _cabi._lib is None                                  # the package's own loader / Engine class are not involved
torch.cuda.set_device(0)
m = ClassifierFreeDiffRoll()
g = torch.Generator().manual_seed(INPUT_SEED)
B, T = 3, 40
wav = (0.1 * torch.randn(B, T * 512, generator=g)).cuda()
x_T = torch.randn(B, 1, T, 88, generator=g).cuda()
roll = m.predict_step((x_T.clone(), wav), 5)
torch.cuda.synchronize()
assert _cabi._lib is None
assert "libdiffroll_amd.so" in open("/proc/self/maps").read()
torch.save(roll.cpu(), OUT)
print("DOC_BINDING_OK", tuple(roll.shape))
'''


def test_the_documented_block_is_valid_python_and_names_the_header_symbols():
    """CPU: the block parses, and every dr_* symbol it calls is declared in include/diffroll_amd.h."""
    block = doc_block()
    compile(block, "INTEGRATION.md#2", "exec")
    header = open(os.path.join(ROOT, "include", "diffroll_amd.h")).read()
    used = set(re.findall(r"_lib\.(dr_\w+)", block))
    assert {"dr_create", "dr_set_param", "dr_set_tables", "dr_set_frontend_tables", "dr_commit", "dr_frontend",
            "dr_sample_checked", "dr_abi_version", "dr_last_error"} <= used
    for name in used:
        assert re.search(r"\b" + name + r"\s*\(", header), name
    assert "_Cfg(8," not in block and "dr_abi_version()" in block          # the ABI number is asked, not hard-coded


@pytest.mark.gpu
def test_the_documented_binding_runs_and_matches_the_facade_bit_for_bit(tmp_path):
    import json
    from test_gpu_parity import make_model
    hp = dict(R.DEFAULT_HP)
    hp.update(residual_channels=64, residual_layers=4, kernel_size=9, timesteps=12)
    out = str(tmp_path / "roll.pt")
    script = "\n".join([
        f"ROOT = {ROOT!r}", f"HP_JSON = {json.dumps(json.dumps(hp))}", "SEED = 21", "INPUT_SEED = 8", f"OUT = {out!r}",
        textwrap.dedent(HOST), doc_block(), textwrap.dedent(DRIVER)])
    path = str(tmp_path / "doc_binding.py")
    open(path, "w").write(script)
    env = dict(os.environ)
    libdir = os.path.join(ROOT, "diffroll_amd", "lib")
    env["LD_LIBRARY_PATH"] = libdir + os.pathsep + env.get("LD_LIBRARY_PATH", "")      # C.CDLL("libdiffroll_amd.so") as documented
    env.pop("DR_TEST_TUNE", None)
    r = subprocess.run([sys.executable, path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DOC_BINDING_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    got = torch.load(out)
    # the facade on the same weights / inputs / Philox seed (predict_step: seed = batch_idx)
    m = make_model(hp, R.synthetic_params(hp, seed=21), sampler="cfdg_ddpm_x0", w=0.5)
    g = torch.Generator().manual_seed(8)
    B, T = 3, 40
    wav = 0.1 * torch.randn(B, T * 512, generator=g)
    x_T = torch.randn(B, 1, T, 88, generator=g)
    want = m.predict_step((x_T, wav), 5).cpu()
    assert got.shape == want.shape == (B, 1, T, 88)
    assert torch.equal(got, want)
