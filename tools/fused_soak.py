#!/usr/bin/env python3
"""Soak of the fused residual-stack kernel at the bench geometry: N captured 200-step chains (N x 200 persistent launches,
29 group barriers each) with the same seed must give bit-identical rolls, and no barrier may ever time out.
    python tools/fused_soak.py [--chains 12] [--config 2]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import tuning_env  # noqa: E402

tuning_env.install()        # DR_TEST_TUNE="tune.stack_fl=2,..." pins engine options for this process
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=12)
    ap.add_argument("--config", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = bench.CONFIGS[args.config]
    hp = dict(bench.HP)
    hp.update(kernel_size=cfg["k"], timesteps=cfg["S"])
    T = cfg["L"] // 512
    m = bench.build_model(dev, hp=hp, sampler=cfg["sampler"])
    g = torch.Generator().manual_seed(9)
    wav = (0.1 * torch.randn(cfg["B"], cfg["L"], generator=g)).to(dev)
    x = torch.randn(cfg["B"], 1, T, 88, generator=g).to(dev)
    ref, _ = m.sample(x, wav, seed=1)
    bad = 0
    for i in range(args.chains):
        m._fe_key = None
        out, _ = m.sample(x, wav, seed=1)
        bad += int(not torch.equal(out, ref))
    flag, _ = m.engine.stack_status()
    m.engine.set_option("fused_stack", 0)
    un, _ = m.sample(x, wav, seed=1)
    # (the per-phase launches of this comparison are the planner's NATURAL choice - other tile widths, split-K: another fp32
    # accumulation order, a few ulp; bit-identity against the per-phase TWINS of each flavour is tests/test_gpu_fused.py)
    print(f"config {args.config}: {args.chains} chains ({args.chains * cfg['S']} fused launches): mismatching chains {bad}, "
          f"barrier time-outs {flag}, max |fused - natural per-phase launches| {float((un - ref).abs().max()):.2e}, "
          f"fused launches captured {m.engine.stack_launches}")
    return 1 if (bad or flag) else 0


if __name__ == "__main__":
    sys.exit(main())
