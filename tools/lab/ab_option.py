#!/usr/bin/env python3
"""Interleaved A/B of one engine option on whole captured chains:  python tools/lab/ab_option.py fused_stack_warm 0 1 [--config 2] [--rounds 4]"""
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
    ap.add_argument("option")
    ap.add_argument("values", nargs="+", type=int)
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = bench.CONFIGS[args.config]
    hp = dict(bench.HP)
    hp.update(kernel_size=cfg["k"], timesteps=cfg["S"])
    T = cfg["L"] // 512
    m = bench.build_model(dev, hp=hp, sampler=cfg["sampler"])
    eng = m.engine
    g = torch.Generator().manual_seed(5)
    wav = (0.1 * torch.randn(cfg["B"], cfg["L"], generator=g)).to(dev)
    x = torch.randn(cfg["B"], 1, T, 88, generator=g).to(dev)
    res = {v: [] for v in args.values}
    for r in range(args.rounds):
        for v in args.values:
            eng.set_option(args.option, v)
            m.sample(x, wav, seed=0)                  # capture
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                m.sample(x, wav, seed=0)
            torch.cuda.synchronize()
            res[v].append(1e3 * (time.perf_counter() - t0) / 2)
    for v in args.values:
        xs = sorted(res[v])
        print(f"{args.option}={v}: median {xs[len(xs) // 2]:.2f} ms, min {xs[0]:.2f}, max {xs[-1]:.2f}  ({['%.1f' % t for t in res[v]]})")


if __name__ == "__main__":
    main()
