#!/usr/bin/env python3
"""Fused residual-stack kernel vs one launch per phase: bitwise comparison of a guided step / generation step at
BASELINE config 2 / 3 geometry, then chain timing of both (and both block mappings), plus the in-kernel phase
tick marks.    python tools/stack_check.py [--quick]"""
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
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--level", type=int, default=1, help="fused_stack option value (2 = forced, e.g. with DR_TEST_TUNE=tune.stack_fl=2)")
    ap.add_argument("--batch", type=int, default=0, help="override the configuration's batch (clips per GPU)")
    ap.add_argument("--k", type=int, default=0, help="override the configuration's kernel size (taps per 32-channel chunk)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = bench.CONFIGS[args.config]
    hp = dict(bench.HP)
    hp.update(kernel_size=args.k or cfg["k"], timesteps=cfg["S"])
    T = cfg["L"] // 512
    m = bench.build_model(dev, hp=hp, sampler=cfg["sampler"])
    eng = m.engine
    g = torch.Generator().manual_seed(5)
    B = args.batch or cfg["B"]
    wav = (0.1 * torch.randn(B, cfg["L"], generator=g)).to(dev)
    x = torch.randn(B, 1, T, 88, generator=g).to(dev)
    z = torch.randn(B, 1, T, 88, generator=g).to(dev)

    def step(t=150):
        out, _ = m.reverse_diffusion(x, wav, t, noise=z)
        return out

    eng.set_option("fused_stack", 0)
    ref = step()
    torch.cuda.synchronize()
    for xcd in (1, 0):
        eng.set_option("fused_stack", args.level)
        eng.set_option("fused_stack_xcd", xcd)
        for rep in range(3):
            out = step()
            flag, _ = eng.stack_status()
            same = bool(torch.equal(out, ref))
            d = float((out - ref).abs().max())
            print(f"fused (xcd mapping {xcd}) rep {rep}: timed_out={flag} bitwise_equal={same} max|diff|={d:.3e}", flush=True)
            if flag or not same:
                bad = (out != ref).nonzero()
                print("  first mismatches:", bad[:5].tolist(), "count", len(bad))
                if flag:
                    return 1
    if args.quick:
        return 0
    # the same body tick marks from the per-phase launches, inside a real step (weights cold, as in the chain)
    eng.set_option("fused_stack", 0)
    eng.set_option("stack_ticks", 1)
    step()
    flag, ticks = eng.stack_status(128)
    print(f"unfused: block 0 body ticks: last conv: K loop {ticks[64]}, body {ticks[65]}; a 1x1: K loop {ticks[96]}, body {ticks[97]}, "
          f"first 2 steps done at {ticks[98]}, before RMW request {ticks[99]}")
    eng.set_option("stack_ticks", 0)
    eng.set_option("fused_stack", args.level)
    # phase ticks (shader-clock cycles of block 0, barrier waits included)
    for xcd in (1, 0):
        eng.set_option("fused_stack_xcd", xcd)
        eng.set_option("stack_ticks", 1)
        step()
        flag, ticks = eng.stack_status(128)
        print(f"xcd={xcd}: block 0 body ticks: last conv: K loop {ticks[64]}, body {ticks[65]}; a 1x1: K loop {ticks[96]}, body {ticks[97]}, first 2 steps done at {ticks[98]}, before RMW request {ticks[99]}")
        cs = [t for t in ticks[66:80] if t]            # chunk-start marks of block 0's last conv phase (first 14 chunks)
        if len(cs) > 1:
            print(f"xcd={xcd}: last conv phase, block 0: first chunk opens at {cs[0]}, chunk durations {[b - a for a, b in zip(cs[:-1], cs[1:])]}")
        wall = ticks[121] - ticks[120]                 # constant 100 MHz counter over the same span as the phase marks
        ticks = ticks[:2 * hp["residual_layers"] + 2]
        nz = [t for t in ticks[:-1] if t]
        if wall > 0 and len(nz) > 1:
            print(f"xcd={xcd}: block 0 ran {nz[-1] - nz[0]} shader cycles in {wall / 100:.1f} us = {(nz[-1] - nz[0]) / wall * 100:.1f} MHz")
        if len(nz) > 1:
            d = [b - a for a, b in zip(nz[:-1], nz[1:])]
            tail, d = d[-1], d[:-1]                 # the last mark pair brackets the write-back of the resident tile
            cut = 2 * min(d)                        # conv phases are ~8x the 1x1 phases
            conv = [v for v in d if v > cut]
            pw = [v for v in d[1:] if v <= cut]     # (the first phase also waits for the resident tile's load)
            print(f"xcd={xcd}: phase ticks (shader cycles, block 0, barrier waits included): {len(conv)} conv phases mean "
                  f"{sum(conv) / max(len(conv), 1):.0f}, {len(pw)} 1x1 phases mean {sum(pw) / max(len(pw), 1):.0f} (first phase {d[0]}), "
                  f"tail {tail}, total {nz[-1] - nz[0]}; per phase {d}", flush=True)
        eng.set_option("stack_ticks", 0)

    def chain_ms(n=2):
        m._fe_key = None
        m.sample(x, wav, seed=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            m._fe_key = None
            r, _ = m.sample(x, wav, seed=0)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n, r

    eng.set_option("fused_stack", 0)
    t_un, r_un = chain_ms()
    for xcd, warm in ((1, 1), (1, 0), (0, 1)):
        eng.set_option("fused_stack", args.level)
        eng.set_option("fused_stack_xcd", xcd)
        eng.set_option("fused_stack_warm", warm)
        t_f, r_f = chain_ms()
        flag, _ = eng.stack_status()
        print(f"chain: unfused {t_un:.1f} ms, fused(xcd={xcd}, warm={warm}) {t_f:.1f} ms ({100 * (t_un - t_f) / t_un:+.2f} %), "
              f"bitwise equal rolls: {bool(torch.equal(r_un, r_f))}, timed_out={flag}", flush=True)
    eng.set_option("fused_stack_warm", 0)
    return 0


if __name__ == "__main__":
    sys.exit(main())
