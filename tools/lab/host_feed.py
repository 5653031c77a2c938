#!/usr/bin/env python3
"""Host side of a graph-replayed chain: how long does the launching thread sit in dr_sample (hipGraphLaunch feeds the
queue as the GPU drains it), how much CPU does one chain cost, and does that change when 8 processes do it at once on
one host (the 8-rank layout; on this 1-GPU box the 8 processes time-slice the GPU, so only HOST numbers are
meaningful in that leg).
    python tools/lab/host_feed.py [--config 2] [--procs 8] [--chains 3]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import tuning_env  # noqa: E402

tuning_env.install()        # DR_TEST_TUNE="tune.stack_fl=2,..." pins engine options for this process


def worker(config, chains, fused, tag):
    import torch
    import bench
    dev = torch.device("cuda", 0)
    cfg = bench.CONFIGS[config]
    hp = dict(bench.HP)
    hp.update(kernel_size=cfg["k"], timesteps=cfg["S"])
    T = cfg["L"] // 512
    m = bench.build_model(dev, hp=hp, sampler=cfg["sampler"])
    eng = m.engine
    eng.set_option("fused_stack", fused)
    g = torch.Generator().manual_seed(5)
    wav = (0.1 * torch.randn(cfg["B"], cfg["L"], generator=g)).to(dev)
    x = torch.randn(cfg["B"], 1, T, 88, generator=g).to(dev)
    m.sample(x, wav, seed=0)                     # capture
    torch.cuda.synchronize()
    # rendezvous file: start all processes' timed legs together
    t_call, t_total, cpu = [], [], []
    for _ in range(chains):
        c0 = time.process_time()
        t0 = time.perf_counter()
        m.sample(x, wav, seed=0)                 # front-end cached: the call is copy-in + hipGraphLaunch + copy-out
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        c1 = time.process_time()
        t_call.append(1e3 * (t1 - t0)); t_total.append(1e3 * (t2 - t0)); cpu.append(1e3 * (c1 - c0))
    med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
    print("HOSTFEED " + json.dumps({"tag": tag, "config": config, "fused": fused, "launch_call_ms": round(med(t_call), 2),
                                    "chain_wall_ms": round(med(t_total), 2), "process_cpu_ms_per_chain": round(med(cpu), 2)}),
          flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--chains", type=int, default=3)
    ap.add_argument("--worker", default=None)
    ap.add_argument("--fused", type=int, default=1)
    args = ap.parse_args()
    if args.worker:
        worker(args.config, args.chains, args.fused, args.worker)
        return
    for fused in (1, 0):
        for procs in (1, args.procs):
            cmd = [sys.executable, os.path.abspath(__file__), "--config", str(args.config), "--chains", str(args.chains),
                   "--fused", str(fused)]
            ps = [subprocess.Popen(cmd + ["--worker", f"p{procs}.{i}"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                  for i in range(procs)]
            rows = []
            for p in ps:
                out, _ = p.communicate(timeout=900)
                rows += [json.loads(ln[len("HOSTFEED "):]) for ln in out.splitlines() if ln.startswith("HOSTFEED ")]
            if rows:
                avg = lambda k: round(sum(r[k] for r in rows) / len(rows), 2)      # noqa: E731
                print(f"config {args.config} fused={fused} processes={procs}: hipGraphLaunch call {avg('launch_call_ms')} ms, "
                      f"chain wall {avg('chain_wall_ms')} ms, host CPU per chain {avg('process_cpu_ms_per_chain')} ms "
                      f"(mean over {len(rows)} processes)", flush=True)


if __name__ == "__main__":
    main()
