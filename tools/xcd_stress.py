#!/usr/bin/env python3
"""Cross-XCD hand-off stress of the fused residual stack: deep (15-layer) full-width net, groups of 32 blocks (4 frame
tiles per clip) - block mapping 1 (a group inside one XCD) is the reference, mapping 0 (a group spread over all XCDs:
every hand-off write-through + L1-bypassing / invalidated loads) must reproduce it bit for bit, repeatedly.
    python tools/xcd_stress.py [--T 500] [--B 4] [--reps 6] [--sampler cfdg_ddpm_x0]      (DR_TEST_TUNE=tune.stack_fl=2 pins the 128-frame flavour)"""
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
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--k", type=int, default=9)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--sampler", default="cfdg_ddpm_x0")
    ap.add_argument("--layers", type=int, default=15)
    ap.add_argument("--chain", type=int, default=0, help="> 0: whole captured chains of that many steps (tail kernel incl. the "
                                                         "next step's input projection / first-layer conv) instead of single steps")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    hp = dict(bench.HP)
    hp.update(kernel_size=args.k, timesteps=args.chain if args.chain > 0 else 200, residual_layers=args.layers)
    m = bench.build_model(dev, hp=hp, sampler=args.sampler)
    eng = m.engine
    g = torch.Generator().manual_seed(11)
    wav = (0.1 * torch.randn(args.B, args.T * 512, generator=g)).to(dev)
    x = torch.randn(args.B, 1, args.T, 88, generator=g).to(dev)
    z = torch.randn(args.B, 1, args.T, 88, generator=g).to(dev)
    eng.set_option("fused_stack", 2)
    eng.set_option("fused_stack_xcd", 1)
    if args.chain > 0:
        chain_noise = torch.randn(args.chain, args.B, 1, args.T, 88, generator=g).to(dev)

        def run():
            return m.sample(x, wav, noise=chain_noise)[0]
    else:
        def run():
            return m.reverse_diffusion(x, wav, 150, noise=z)[0]
    eng.profile_enable(True)
    ref = m.reverse_diffusion(x, wav, min(150, hp["timesteps"] - 1), noise=z)[0]
    _, _, _, kname = eng.profile_read_ex()
    eng.profile_enable(False)
    ref = run()
    again = run()
    print(f"kernel {kname.split(' ')[0]}; mapping 1 repeatable: {bool(torch.equal(ref, again))}")
    eng.set_option("fused_stack_xcd", 0)
    bad = 0
    for r in range(args.reps):
        out = run()
        flag, _ = eng.stack_status()
        eq = bool(torch.equal(out, ref))
        d = float((out - ref).abs().max())
        nz = (out != ref).nonzero()
        where = ""
        if len(nz):
            fr = nz[:, 2]
            where = f" mismatches {len(nz)}: samples {sorted(set(nz[:, 0].tolist()))}, frames {int(fr.min())}..{int(fr.max())}"
        print(f"mapping 0 rep {r}: timed_out={flag} equal={eq} max|diff|={d:.3e}{where}", flush=True)
        bad += (not eq)
    import time
    for mapping in (1, 0):
        eng.set_option("fused_stack_xcd", mapping)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        print(f"mapping {mapping}: {1e3 * (time.perf_counter() - t0) / 5:.3f} ms per {'chain' if args.chain else 'step'}")
    print("RESULT", "FAIL" if bad else "ok", bad, "of", args.reps)


if __name__ == "__main__":
    main()
