#!/usr/bin/env python3
"""In-kernel tick marks of the tail kernel (block 0, s_memtime = shader cycles) at a BASELINE config's geometry, and
rocprof-free wall times of the step's launches: python tools/tail_ticks.py [--config 2]"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = bench.CONFIGS[args.config]
    hp = dict(bench.HP)
    hp.update(kernel_size=cfg["k"], timesteps=8)
    T = cfg["L"] // 512
    m = bench.build_model(dev, hp=hp, sampler=cfg["sampler"])
    eng = m.engine
    g = torch.Generator().manual_seed(5)
    wav = (0.1 * torch.randn(cfg["B"], cfg["L"], generator=g)).to(dev)
    x = torch.randn(cfg["B"], 1, T, 88, generator=g).to(dev)
    m.sample(x, wav, seed=0, use_graph=False)
    eng.set_option("stack_ticks", 1)
    m.sample(x, wav, seed=0, use_graph=False)
    flag, ticks = eng.stack_status(128)
    tk = ticks[112:120]
    names = ["T1 skip projection", "group barrier", "T2 output projection", "pair barrier", "T3 update + input projection",
             "pair barrier", "T4 shared first-layer conv"]
    print(f"tail kernel, block 0, shader cycles (2.4 GHz: 2400 cycles = 1 us); timed_out={flag}")
    for i, n in enumerate(names):
        if tk[i + 1] and tk[i]:
            d = tk[i + 1] - tk[i]
            print(f"  {n:32s} {d:8d} cycles  {d / 2400:7.2f} us")
    if tk[7] and tk[0]:
        print(f"  {'total':32s} {tk[7] - tk[0]:8d} cycles  {(tk[7] - tk[0]) / 2400:7.2f} us")
    eng.set_option("stack_ticks", 0)
    for tail in (1, 0):
        eng.set_option("fused_tail", tail)
        m.sample(x, wav, seed=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m.sample(x, wav, seed=0, check=False)
        torch.cuda.synchronize()
        print(f"fused_tail={tail}: {1e3 * (time.perf_counter() - t0) / 20 / 8:.4f} ms per reverse step (8-step chains, graph)")


if __name__ == "__main__":
    main()
