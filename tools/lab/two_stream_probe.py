#!/usr/bin/env python3
"""Feasibility probe: run the conditional and the unconditional half of a classifier-free-guidance chain as two
independent chains on two HIP streams (two engines) and compare with the batched 2B chain on one stream."""
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
    B, T = 16, 125
    g = torch.Generator().manual_seed(0)
    wav = (0.1 * torch.randn(B, T * 512, generator=g)).to(dev)
    x = torch.randn(B, 1, T, 88, generator=g).to(dev)
    m2 = bench.build_model(dev, sampler="cfdg_ddpm_x0", w=0.5)
    mc = bench.build_model(dev, sampler="ddpm_x0", w=0.0)
    mu = bench.build_model(dev, sampler="generation_ddpm_x0", w=0.0)

    def timed(fn, n=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    print("cfdg (2B batched, one stream): %.1f ms" % timed(lambda: m2.sample(x, wav, seed=0)))
    print("ddpm_x0 alone: %.1f ms" % timed(lambda: mc.sample(x, wav, seed=0)))
    print("generation alone: %.1f ms" % timed(lambda: mu.sample(x, wav, seed=0)))
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def both():
        with torch.cuda.stream(s1):
            mc.sample(x, wav, seed=0)
        with torch.cuda.stream(s2):
            mu.sample(x, wav, seed=0)

    print("ddpm_x0 || generation on two streams, one host thread: %.1f ms" % timed(both))

    # hipGraphLaunch of a 6800-node graph blocks the calling thread for ~half the chain (the queue holds ~4k
    # packets), so a single host thread launches the second graph late: give each stream its own thread
    import threading

    def run(model, stream):
        with torch.cuda.stream(stream):
            model.sample(x, wav, seed=0)

    def both_threads():
        ts = [threading.Thread(target=run, args=(mc, s1)), threading.Thread(target=run, args=(mu, s2))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    print("ddpm_x0 || generation on two streams, two host threads: %.1f ms" % timed(both_threads))


if __name__ == "__main__":
    main()
