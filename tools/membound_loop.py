#!/usr/bin/env python3
"""Target of `rocprofv3 --kernel-trace`: every HBM-bound kernel of the path at one geometry, REPS times each -
reflect_pad / stft_power / minmax / normalize (dr_frontend), update_kernel (dr_step with the tail kernel off),
noise_mix_kernel (dr_q_sample), note_runs_kernel (dr_note_runs), frame_counts_kernel (dr_frame_counts).  A narrow network
(C = 64, 2 layers) carries the calls: none of these kernels' bytes depend on the network width.
    python tools/membound_loop.py --size cfg2|cfg5|cfg7|large [--reps 20]
Prints the geometry as one JSON line (tools/membound_report.py joins it with the kernel trace)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SIZES = {   # per-GPU shapes of the BASELINE configurations (bench.CONFIGS) and one far beyond them
    "cfg2": dict(B=16, L=64000, sampler="cfdg_ddpm_x0", what="BASELINE config 2 / 4 per GPU: 16 guided clips x 125 frames"),
    "cfg3": dict(B=16, L=64000, sampler="generation_ddpm_x0", what="BASELINE config 3 per GPU: 16 generated rolls x 125 frames"),
    "cfg5": dict(B=4, L=327680, sampler="cfdg_ddpm_x0", what="BASELINE config 5 per GPU: 4 guided clips x 640 frames"),
    "cfg7": dict(B=16, L=327680, sampler="generation_ddpm_x0", what="config 7 (config 3 at the reference's 640-frame length): 16 rolls x 640 frames"),
    "large": dict(B=64, L=327680, sampler="cfdg_ddpm_x0", what="64 guided clips x 640 frames (15x config 5): where the curves flatten"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="cfg2", choices=sorted(SIZES))
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    sz = SIZES[args.size]
    dev = torch.device("cuda", 0)
    hp = dict(bench.HP)
    hp.update(residual_channels=64, residual_layers=2, timesteps=200)
    m = bench.build_model(dev, hp=hp, sampler=sz["sampler"])
    eng = m.engine
    eng.set_option("fused_tail", 0)           # the update as its own kernel (configs 5-7 run it that way)
    B, L = sz["B"], sz["L"]
    T = L // 512
    g = torch.Generator().manual_seed(3)
    wav = (0.1 * torch.randn(B, L, generator=g)).to(dev)
    x = torch.randn(B, 1, T, 88, generator=g).to(dev)
    label = (torch.rand(B, T, 88, generator=g) > 0.9).float().to(dev)
    from diffroll_amd import q_sample
    sch = {k: getattr(m, k) for k in ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod")}
    tt = torch.randint(0, 200, (B,), generator=g)
    roll = x
    for i in range(args.reps):
        m._fe_key = None
        roll, _ = m.reverse_diffusion(x, wav, 199 - i, noise=None)       # front-end (re-run every rep) + one step: update_kernel with Philox
        q_sample(x, tt, sch["sqrt_alphas_cumprod"], sch["sqrt_one_minus_alphas_cumprod"], roll)
        eng.note_runs(roll[:, 0].contiguous(), 0.5)
        eng.frame_counts(roll[:, 0].contiguous(), label, 0.5)
    torch.cuda.synchronize()
    n = B * T * 88
    TF = T + 1
    pad = 1024
    Lp = (L + 2 * pad + 3) & ~3
    guided = sz["sampler"] != "generation_ddpm_x0"
    geo = {
        "size": args.size, "what": sz["what"], "B": B, "L": L, "T": T, "reps": args.reps,
        # ALGORITHMIC bytes per launch: every operand once
        "bytes": {
            "reflect_pad4_kernel": 4 * B * L + 4 * B * Lp,
            "stft_power_kernel": 4 * B * Lp + 4 * B * TF * 1088,                  # the padded clip once (frames overlap 4x) + the power rows
            "minmax_kernel": 16 * B * 58 * TF,                                    # 229 mel rows = 58 planes of 4
            "normalize_kernel": 16 * B * 58 * T + 16 * B * 58 * T + 4 * B * 229 * T,   # read log-mel, write P4 spec + the plain spec handed back
            "update_kernel": 4 * n * (4 if guided else 3),                        # read x, x0 (cond [, uncond]), write x; Philox noise: no bytes
            "noise_mix_kernel": 4 * n * 3,
            "note_runs_kernel": 4 * n * 2,
            "frame_counts_kernel": 4 * n * 2,
        },
    }
    print("MEMBOUND_GEOMETRY " + json.dumps(geo))


if __name__ == "__main__":
    main()
