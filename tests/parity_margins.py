#!/usr/bin/env python3
"""Print the observed HIP-vs-oracle differences behind the tolerances of tests/test_gpu_parity.py
(a script, not a test: `python tests/parity_margins.py` on an MI355X; it lives under tests/ because it uses the
oracle, which only test code may import)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import diffroll_ref as R                      # noqa: E402  (checker only)
from tests.test_gpu_parity import make_model             # noqa: E402


def main():
    for prec in ("f32", "bf16x3"):
        hp = dict(R.DEFAULT_HP)
        hp.update(kernel_size=9, timesteps=200)
        p = R.synthetic_params(hp, seed=0)
        m = make_model(hp, p, sampler="cfdg_ddpm_x0", w=0.5, precision=prec)
        torch.manual_seed(0)
        B, Tn = 2, 125
        wav = 0.1 * torch.randn(B, Tn * 512)
        x = torch.randn(B, 1, Tn, 88)
        t = torch.tensor(117).repeat(B)
        with torch.no_grad():
            ref_c, ref_spec = R.forward(p, hp, x, wav, t)
        x0_c, spec = m(x, wav, t)
        print(f"[{prec}] one evaluation (C=512, L=15, k=9, T=125): max|x0 - oracle| = {float((x0_c.cpu() - ref_c).abs().max()):.2e}, "
              f"max|spec - oracle| = {float((spec.cpu() - ref_spec).abs().max()):.2e}, |x0|max = {float(ref_c.abs().max()):.2f}")
        hp1 = dict(hp)
        hp1["timesteps"] = 50
        p1 = R.synthetic_params(hp1, seed=0)
        m1 = make_model(hp1, p1, sampler="cfdg_ddpm_x0", w=0.5, precision=prec)
        wav1 = 0.1 * torch.randn(1, Tn * 512)
        x1 = torch.randn(1, 1, Tn, 88)
        nz = torch.randn(50, 1, 1, Tn, 88)
        with torch.no_grad():
            ref = R.sample_chain(p1, hp1, "cfdg_ddpm_x0", x1, wav1, nz, w=0.5)
        roll, _ = m1.sample(x1, wav1, noise=nz)
        d = (roll.cpu() - ref).abs()
        print(f"[{prec}] config-1 chain (50 steps, cfdg w=0.5): max|roll - oracle| = {float(d.max()):.2e}, "
              f"thresholded (>0.5) frames differing: {int(((roll.cpu() > 0.5) != (ref > 0.5)).sum())}")


if __name__ == "__main__":
    main()
