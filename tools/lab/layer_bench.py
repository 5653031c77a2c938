#!/usr/bin/env python3
"""Micro-benchmark of the fused dilated-conv+gate kernel (dr_bench_layer) on one MI355X.
    python tools/lab/layer_bench.py [--k 9] [--B 16] [--T 125] [--iters 50] [--layers 0,1,2,3]
Prints per-layer mean launch time (HIP events) and achieved TFLOP/s."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import tuning_env  # noqa: E402

tuning_env.install()        # DR_TEST_TUNE="tune.stack_fl=2,..." pins engine options for this process
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=9)
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--T", type=int, default=125)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--layers", default="0,1,2,3")
    ap.add_argument("--uncond", action="store_true", help="all samples unconditional (generation)")
    ap.add_argument("--pointwise", action="store_true", help="time the 1x1 output projection kernel instead")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3"])
    ap.add_argument("--cycle", action="store_true",
                    help="time the layers round-robin (every launch streams a different layer's weights and conditioner "
                         "from HBM, as inside the sampling chain) instead of one L2-hot layer at a time")
    args = ap.parse_args()
    hp = dict(bench.HP)
    hp["kernel_size"] = args.k
    dev = torch.device("cuda", 0)
    m = bench.build_model(dev, hp=hp)
    m.precision = args.precision
    eng = m.engine
    L = args.T * hp["hop_length"]
    wav = (0.1 * torch.randn(args.B, L)).to(dev)
    NB = args.B if args.uncond else 2 * args.B
    n_cond = 0 if args.uncond else args.B
    if n_cond:
        eng.frontend(wav, args.T)
    # fill EVERY sample slot of the hidden state with real activations (one real evaluation of the same batch
    # geometry): MFMA power - and with it the clock - depends on the data, half-zero inputs flatter the kernel
    x = torch.randn(args.B, args.T, 88, device=dev)
    if n_cond:
        eng.step("cfdg_ddpm_x0", x.clone(), None, 100, w=0.5)
    else:
        eng.forward(x, 100, uncond=True)
    flops = 2.0 * 512 * 1024 * args.k * NB * args.T
    if args.pointwise and not args.cycle:
        flops = 2.0 * 512 * 1024 * NB * args.T
        for layer in [int(v) for v in args.layers.split(",")]:
            for _ in range(5):
                eng.bench_pointwise(layer, NB, args.T)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                eng.bench_pointwise(layer, NB, args.T)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            lt, bt = eng.debug_ticks()
            nmfma = 64 * (hp["residual_channels"] // 32)
            print(f"1x1 layer {layer:2d} NB={NB} T={args.T}: {us:8.2f} us  {flops / us / 1e6:7.2f} TFLOP/s | block0 ticks: "
                  f"loop {lt} total {bt} ({bt / 2.33e3:.1f} us at 2.33 GHz), loop ticks/MFMA {lt / nmfma:.1f}", flush=True)
        return
    if args.cycle:
        layers = [int(v) for v in args.layers.split(",")]
        for layer in layers:
            eng.bench_layer(layer, NB, args.T, 100, n_cond)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            for layer in layers:
                if args.pointwise:            # 1x1 -> conv -> 1x1 ... as in the chain (time is per PAIR then)
                    eng.bench_pointwise(layer, NB, args.T)
                eng.bench_layer(layer, NB, args.T, 100, n_cond)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (args.iters * len(layers))
        lt, bt = eng.debug_ticks()
        print(f"round-robin over layers {layers} k={args.k} NB={NB} T={args.T}: {us:8.2f} us per launch  "
              f"{flops / us / 1e6:7.2f} TFLOP/s | last conv launch, block 0: loop {lt} total {bt} ticks", flush=True)
        return
    for layer in [int(v) for v in args.layers.split(",")]:
        for _ in range(5):
            eng.bench_layer(layer, NB, args.T, 100, n_cond)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            eng.bench_layer(layer, NB, args.T, 100, n_cond)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        lt, bt = eng.debug_ticks()
        nmfma = 64 * (hp["residual_channels"] // 32) * args.k          # per wave (NI=2)
        print(f"layer {layer:2d} k={args.k} NB={NB} T={args.T}: {us:8.2f} us  {flops / us / 1e6:7.2f} TFLOP/s "
              f"({100 * flops / us / 1e6 / 157.3:5.1f}% of fp32 MFMA peak) | block0 ticks: loop {lt} total {bt} "
              f"-> clock >= {bt / us / 1e3:.3f} GHz, loop ticks/MFMA {lt / nmfma:.1f}", flush=True)


if __name__ == "__main__":
    main()
