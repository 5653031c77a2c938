#!/usr/bin/env python3
"""Per-kernel time of one eager chain step set (torch profiler-free): runs N reverse steps at config 2 and
prints wall per step; use under rocprofv3 --kernel-trace for per-kernel numbers."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import tuning_env  # noqa: E402

tuning_env.install()        # DR_TEST_TUNE="tune.stack_fl=2,..." pins engine options for this process
import bench
dev = torch.device("cuda", 0)
m = bench.build_model(dev)
m.precision = sys.argv[1] if len(sys.argv) > 1 else "f32"
T = bench.L_SAMPLES // 512
wav = (0.1 * torch.randn(bench.B_LOCAL, bench.L_SAMPLES)).to(dev)
x = torch.randn(bench.B_LOCAL, 1, T, 88).to(dev)
m.sample(x, wav, seed=0)
torch.cuda.synchronize()
t0 = time.perf_counter()
m.sample(x, wav, seed=0)
torch.cuda.synchronize()
print(f"chain: {1e3 * (time.perf_counter() - t0):.1f} ms")
