#!/usr/bin/env python3
"""Repeatability soak for the split-K path (cross-XCD hand-over through the workspace): small launches, every
chain run twice with the same seed must agree bit for bit - a stale or torn read of a partial would not.
    python tools/determinism_soak.py [iterations]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import tuning_env  # noqa: E402

tuning_env.install()        # DR_TEST_TUNE="tune.stack_fl=2,..." pins engine options for this process
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda", 0)
    hp = dict(bench.HP)
    hp["timesteps"] = 10
    models = [bench.build_model(dev, hp=hp, sampler=s, w=0.5) for s in ("cfdg_ddpm_x0", "generation_ddpm_x0")]
    g = torch.Generator().manual_seed(1)
    t0 = time.perf_counter()
    launches = 0
    for it in range(n):
        m = models[it % 2]
        B = (1, 2, 3, 1)[it % 4]
        T = (33, 125, 200, 640)[(it // 4) % 4]
        m.precision = "bf16x3" if it % 7 == 6 else "f32"
        wav = (0.1 * torch.randn(B, T * 512, generator=g)).to(dev)
        x = torch.randn(B, 1, T, 88, generator=g).to(dev)
        a, _ = m.sample(x, wav, seed=it)
        b, _ = m.sample(x, wav, seed=it)
        assert torch.equal(a, b), (it, B, T, float((a - b).abs().max()))
        assert bool(torch.isfinite(a).all())
        launches += 2 * 10 * 33
    torch.cuda.synchronize()
    print(f"{n} shape/seed combinations, {launches} kernel launches, every chain bitwise repeatable ({time.perf_counter() - t0:.0f} s)")


if __name__ == "__main__":
    main()
