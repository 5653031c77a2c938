#!/usr/bin/env python3
"""Chain time of small guided batches (the launches the under-filled kernel choices apply to), for A/B runs under
different tune.* options:    DR_TEST_TUNE=tune.pwk=0 python tools/lab/small_batch_ab.py [--batches 1,2,3,4,6] [--T 125] [--steps 50]"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,2,3,4,6")
    ap.add_argument("--T", type=int, default=125)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--sampler", default="cfdg_ddpm_x0")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    hp = dict(bench.HP)
    hp.update(timesteps=args.steps)
    m = bench.build_model(dev, hp=hp, sampler=args.sampler)
    g = torch.Generator().manual_seed(3)
    for B in [int(v) for v in args.batches.split(",")]:
        wav = (0.1 * torch.randn(B, args.T * 512, generator=g)).to(dev)
        x = torch.randn(B, 1, args.T, 88, generator=g).to(dev)
        m.sample(x, wav, seed=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            m._fe_key = None
            m.sample(x, wav, seed=0)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        print(f"B={B} T={args.T} {args.sampler}: {ms:.2f} ms per {args.steps}-step chain ({ms / args.steps * 1e3:.0f} us per step)", flush=True)


if __name__ == "__main__":
    main()
