#!/usr/bin/env python3
"""Whole-chain time of the five BASELINE.json configs at their PER-GPU shapes on one MI355X
(graph-captured chain incl. front-end, Philox noise, random-init weights).
    python tools/lab/config_bench.py [--only 2,5]"""
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

CONFIGS = {
    1: dict(name="cfg1 k=9 S=50 B=1 T=125 cfdg w=0.5", k=9, S=50, B=1, T=125, sampler="cfdg_ddpm_x0", evals=2),
    2: dict(name="cfg2 k=9 S=200 B=16 T=125 cfdg w=0.5", k=9, S=200, B=16, T=125, sampler="cfdg_ddpm_x0", evals=2),
    3: dict(name="cfg3/GPU k=9 S=200 B=16 T=125 generation", k=9, S=200, B=16, T=125, sampler="generation_ddpm_x0", evals=1),
    4: dict(name="cfg4/GPU k=9 S=200 B=16 T=125 inpainting w=0.5", k=9, S=200, B=16, T=125, sampler="inpainting_ddpm_x0", evals=2),
    5: dict(name="cfg5/GPU k=15 S=200 B=4 T=640 cfdg w=0.5", k=15, S=200, B=4, T=640, sampler="cfdg_ddpm_x0", evals=2),
    6: dict(name="ref-default k=9 S=200 B=16 T=640 cfdg w=0.5", k=9, S=200, B=16, T=640, sampler="cfdg_ddpm_x0", evals=2),
}


def flops_per_frame_eval(k, C=512, L=15):
    return 2 * 88 * C + L * (2 * C * 2 * C * k + 2 * C * 2 * C) + 2 * C * C + 2 * C * 88


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="1,2,3,4,5,6")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for cid in [int(v) for v in args.only.split(",")]:
        c = CONFIGS[cid]
        hp = dict(bench.HP)
        hp.update(kernel_size=c["k"], timesteps=c["S"])
        m = bench.build_model(dev, hp=hp, sampler=c["sampler"], w=0.5)
        if c["sampler"] == "inpainting_ddpm_x0":
            m.hparams.inpainting_t = [c["T"] // 4, c["T"] // 2]
        L = c["T"] * 512
        g = torch.Generator().manual_seed(cid)
        wav = (0.1 * torch.randn(c["B"], L, generator=g)).to(dev)
        x = torch.randn(c["B"], 1, c["T"], 88, generator=g).to(dev)

        def run():
            m._fe_key = None
            roll, _ = m.sample(x, wav, seed=0)
            return roll.cpu()

        run()
        torch.cuda.synchronize()
        n = 2 if c["S"] * c["B"] * c["T"] > 100000 else 5
        t0 = time.perf_counter()
        for _ in range(n):
            out = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        assert bool(torch.isfinite(out).all())
        fl = flops_per_frame_eval(c["k"]) * c["B"] * c["T"] * c["evals"] * c["S"]
        print(f"{c['name']:52s} {1e3 * dt:9.1f} ms/chain  {c['B'] * c['T'] / dt:9.1f} frames/s  "
              f"{fl / dt / 1e12:6.1f} TFLOP/s whole-chain ({100 * fl / dt / 157.3e12:4.1f}% of fp32 MFMA peak)", flush=True)
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
