#!/usr/bin/env python3
"""Soak: alternate shapes, samplers, precisions and seeds on one engine; every roll finite, device memory stable."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import tuning_env  # noqa: E402

tuning_env.install()        # DR_TEST_TUNE="tune.stack_fl=2,..." pins engine options for this process
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    hp = dict(bench.HP)
    hp["timesteps"] = 20
    models = {s: bench.build_model(dev, hp=hp, sampler=s, w=0.5) for s in
              ("cfdg_ddpm_x0", "generation_ddpm_x0", "inpainting_ddpm_x0", "ddim_x0")}
    models["inpainting_ddpm_x0"].hparams.inpainting_t = [10, 40]
    g = torch.Generator().manual_seed(0)
    free0 = None
    t0 = time.perf_counter()
    for it in range(40):
        s = list(models)[it % 4]
        B = (1, 3, 16, 7)[(it // 4) % 4]
        T = (125, 97, 640, 200)[(it // 3) % 4]
        m = models[s]
        m.precision = "bf16x3" if it % 5 == 4 else "f32"
        wav = (0.1 * torch.randn(B, T * 512, generator=g)).to(dev)
        x = torch.randn(B, 1, T, 88, generator=g).to(dev)
        roll, _ = m.sample(x, wav, seed=it)
        assert roll.shape == (B, 1, T, 88) and bool(torch.isfinite(roll).all()), (it, s, B, T)
        torch.cuda.synchronize()
        if it == 19:
            free0 = torch.cuda.mem_get_info()[0]
    free1 = torch.cuda.mem_get_info()[0]
    print(f"40 chains ok in {time.perf_counter() - t0:.1f} s; free memory after 20: {free0 >> 20} MiB, after 40: {free1 >> 20} MiB")
    assert abs(free0 - free1) < (256 << 20), "device memory drifts"


if __name__ == "__main__":
    main()
