#!/usr/bin/env python3
"""N guided reverse steps at a BASELINE config's per-GPU geometry (eager launches: in-proj, layer-0 conv, the fused
residual-stack kernel, skip / output projection, update) - the target of the rocprofv3 counter passes.
    python tools/step_loop.py [--config 2] [--iters 10] [--fused 1]"""
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
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--fused", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = bench.CONFIGS[args.config]
    hp = dict(bench.HP)
    hp.update(kernel_size=cfg["k"], timesteps=cfg["S"])
    T = cfg["L"] // 512
    m = bench.build_model(dev, hp=hp, sampler=cfg["sampler"],
                          inpainting_t=[T // 4, T // 2] if cfg["sampler"] == "inpainting_ddpm_x0" else None)
    m.engine.set_option("fused_stack", args.fused)
    g = torch.Generator().manual_seed(5)
    wav = (0.1 * torch.randn(cfg["B"], cfg["L"], generator=g)).to(dev)
    x = torch.randn(cfg["B"], 1, T, 88, generator=g).to(dev)
    z = torch.randn(cfg["B"], 1, T, 88, generator=g).to(dev)
    for i in range(args.iters):
        x, _ = m.reverse_diffusion(x, wav, cfg["S"] - 1 - (i % cfg["S"]), noise=z)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(x).all())
    print("step_loop done", args.iters)


if __name__ == "__main__":
    main()
