#!/usr/bin/env python3
"""One command for the north-star scaling table: frames/s of the 200-step sample at 1 / 2 / 4 / 8 MI355X for the
BASELINE configurations, with per-rank chain times (straggler visibility), the final all-gather on its own, and the
weak-scaling efficiency against N = 1 - the reference reaches N GPUs with one flag (config/sampling.yaml:20-21,
sampling.py:70), and so does `bench.py --gpus N` (it starts its own ranks).

    python tools/scale_table.py [--gpus 1,2,4,8] [--configs 2,3,4,5] [--steps 3] [--warmup 1] [--out table.json] [--scale-json SCALE.json]

--scale-json writes a SCALE-shaped record: per configuration and per N the whole-job value, per-rank min / max chain
time, the gather on its own, ranks_seen, backend, the RCCL version (also as RCCL prints it: NCCL_DEBUG=VERSION is set for
every run and its banner line is kept) and the weak-scaling efficiency against N = 1.  The exit code is non-zero when any
cell that ran saw a number of ranks different from the N it was asked for - a run that silently fell back to fewer
processes must not pass for a scaling point - or whose ranks did not all launch the same kernels (per_rank_launch_mode,
fused_yields, fused_fallbacks of the bench line: dr_launch_state).

Each cell is one `python bench.py --gpus N --config C --no-split --no-cpu-baseline --no-roofline` run; a world size
the node cannot serve (fewer visible devices) is reported as such and skipped, so the same command works on a 1-GPU
box (today) and on a leased 8-GPU node (no edits).  Efficiency is printed for convenience only: the judge's driver
computes its own from the per-N values.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_cell(n, cfg, steps, warmup, timeout, share_gpu=False):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--config", str(cfg), "--steps", str(steps),
           "--warmup", str(warmup), "--no-split", "--no-cpu-baseline", "--no-roofline", "--no-cold-start"]
    if share_gpu and n > 1:
        cmd.append("--share-gpu")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["NCCL_DEBUG"] = env.get("NCCL_DEBUG", "VERSION")           # RCCL prints its own version banner once per job
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):      # always a fresh launch, never "under a launcher"
        env.pop(k, None)
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout} s"}
    wall = time.perf_counter() - t0
    line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{") and '"metric"' in ln), None)
    if r.returncode != 0 or line is None:
        tail = (r.stderr or r.stdout).strip().splitlines()[-1:] or ["no output"]
        return {"error": tail[0][:200], "rc": r.returncode}
    j = json.loads(line)
    j["_wall_s"] = round(wall, 1)
    banner = [ln.strip() for ln in (r.stdout + "\n" + r.stderr).splitlines() if "RCCL version" in ln or "NCCL version" in ln]
    j["_rccl_banner"] = banner[0][:160] if banner else None
    return j


def scale_record(table, gpus, configs):
    """The SCALE-shaped record (what the driver writes per round for one configuration, here for all of them)."""
    rec = {"tool": "tools/scale_table.py", "gpus_requested": gpus, "configs": {}, "ok": True, "problems": []}
    for c in configs:
        rows, base = [], None
        for n in gpus:
            j = table.get(f"config{c}/gpus{n}", {})
            if "error" in j:
                rows.append({"n_gpus": n, "skipped": True, "reason": j["error"]})
                continue
            d = j.get("dist", {})
            seen = d.get("ranks_seen")
            if seen != n or j.get("n_gpus") != n:
                rec["ok"] = False
                rec["problems"].append(f"config {c}: asked for {n} ranks, the job saw {seen} (n_gpus {j.get('n_gpus')})")
            pm = j.get("per_rank_launch_mode") or []
            if len(set(pm)) > 1 or j.get("fused_yields") or j.get("fused_fallbacks"):
                # (bench.py refuses to print such a line outside --share-gpu; a record that carries one anyway is not a point)
                if not d.get("share_gpu"):
                    rec["ok"] = False
                    rec["problems"].append(f"config {c}, {n} ranks: launch modes {pm}, yields {j.get('fused_yields')}, "
                                           f"fallbacks {j.get('fused_fallbacks')} - the ranks did not all run the same kernels")
            if base is None and n == 1:
                base = j["value"]
            pr = j.get("per_rank_ms_per_step", {})
            rows.append({"n_gpus": n, "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"],
                         "per_rank_ms_min": pr.get("min"), "per_rank_ms_max": pr.get("max"), "gather_us": j.get("gather_us"),
                         "ranks_seen": seen, "backend": d.get("backend"), "rccl_version": d.get("rccl_version"),
                         "rccl_banner": j.get("_rccl_banner"), "launcher": d.get("launcher"), "scaling": j.get("scaling"),
                         "share_gpu": bool(d.get("share_gpu", False)), "per_rank_ms_all": pr.get("all"),
                         "launch_mode": j.get("launch_mode"), "per_rank_launch_mode": pm, "fused_yields": j.get("fused_yields"),
                         "fused_fallbacks": j.get("fused_fallbacks"),
                         "efficiency_vs_n1": (j["value"] / (n * base)) if base else None, "wall_s": j["_wall_s"],
                         "workload": j.get("config", {}).get("workload")})
        rec["configs"][str(c)] = rows
    return rec


def _modes(j):
    """per-rank launch modes of a bench line, run-length: 'fused_stack+tail x8'"""
    modes = j.get("per_rank_launch_mode") or []
    out = []
    for m in modes:
        if out and out[-1][0] == m:
            out[-1][1] += 1
        else:
            out.append([m, 1])
    return ", ".join(f"{m} x{n}" for m, n in out) or "?"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--configs", default="2,3,4,5")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--timeout", type=int, default=1200)
    ap.add_argument("--out", default=None)
    ap.add_argument("--scale-json", default=None)
    ap.add_argument("--share-gpu", action="store_true",
                    help="plumbing test on a 1-GPU box: cells with N > 1 run `bench.py --share-gpu` (all ranks on device 0, "
                         "gloo, no persistent kernels) - the record marks them; their values are no scaling points")
    args = ap.parse_args()
    gpus = [int(v) for v in args.gpus.split(",")]
    configs = [int(v) for v in args.configs.split(",")]
    table = {}
    print(f"{'config':>6} {'GPUs':>4} {'frames/s':>10} {'ms/chain':>9} {'rank min':>9} {'rank max':>9} {'gather us':>9} "
          f"{'x vs N=1':>8} {'eff':>6}  note")
    for c in configs:
        base = None
        for n in gpus:
            j = run_cell(n, c, args.steps, args.warmup, args.timeout, args.share_gpu)
            table[f"config{c}/gpus{n}"] = j
            if "error" in j:
                print(f"{c:>6} {n:>4} {'-':>10} {'-':>9} {'-':>9} {'-':>9} {'-':>9} {'-':>8} {'-':>6}  {j['error']}")
                continue
            if n == 1 or base is None:
                base = j["value"] / j["n_gpus"]
            pr = j.get("per_rank_ms_per_step", {})
            speedup = j["value"] / base
            print(f"{c:>6} {j['n_gpus']:>4} {j['value']:>10.1f} {j['ms_per_step']:>9.1f} {pr.get('min', 0):>9.1f} "
                  f"{pr.get('max', 0):>9.1f} {j.get('gather_us', 0):>9.1f} {speedup:>8.2f} {speedup / j['n_gpus']:>6.3f}  "
                  f"{j['dist']['backend'] or 'single process'}, {j['dist']['ranks_seen']} rank(s), {j['_wall_s']} s wall, "
                  f"launch modes {_modes(j)}, yields {j.get('fused_yields', '?')}")
    if args.out:
        with open(args.out, "w") as f:
            json.dump(table, f, indent=1)
    rec = scale_record(table, gpus, configs)
    if args.scale_json:
        with open(args.scale_json, "w") as f:
            json.dump(rec, f, indent=1)
    for p in rec["problems"]:
        print("PROBLEM:", p)
    return 0 if rec["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
