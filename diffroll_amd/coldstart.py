"""Time-to-first-roll of a one-shot process (what the reference ships: sampling.py:22-73 = load checkpoint -> ONE
trainer.predict).  Run in a FRESH process:

    python -m diffroll_amd.coldstart --config {1,2,...} [--json]

Prints the split a user feels: interpreter + torch import, library load, engine creation (dr_create + host tables), weight
hand-over (dr_set_param copies), dr_commit (host packing / uploads / device-built tables, from dr_cold_times), the first
sample (front-end + capture + instantiate + chain) and the steady-state chain.  Synthetic weights of the configuration's
architecture (no checkpoint exists in this image); the checkpoint read itself is torch.load and is not ours."""
import argparse
import json
import os
import sys
import time

T0 = time.perf_counter()


def proc_age_s():
    """Seconds since the process was created (includes interpreter start-up), from /proc."""
    try:
        with open("/proc/self/stat") as f:
            start_ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return up - start_ticks / os.sysconf("SC_CLK_TCK")
    except Exception:       # noqa: BLE001
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--steady", type=int, default=3)
    ap.add_argument("--tune", action="append", default=[], metavar="NAME=VALUE",
                    help="dr_set_option applied right after dr_create (e.g. tune.pack_threads=1, tune.s3_eager=1: the 'before' "
                         "of the cold-start report)")
    args = ap.parse_args()
    out = {"config": args.config}
    out["interpreter_s"] = (proc_age_s() or 0.0) - (time.perf_counter() - T0) if proc_age_s() is not None else None
    t = time.perf_counter()
    import torch
    out["import_torch_s"] = time.perf_counter() - t
    t = time.perf_counter()
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from diffroll_amd import _cabi
    _cabi.load_library()
    out["import_package_s"] = time.perf_counter() - t
    t = time.perf_counter()
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    out["hip_init_s"] = time.perf_counter() - t
    cfg = bench.CONFIGS[args.config]
    hp = dict(bench.HP)
    hp.update(kernel_size=cfg["k"], timesteps=cfg["S"])
    t = time.perf_counter()
    m = bench.build_model(dev, hp=hp, sampler=cfg["sampler"])       # host-side module + synthetic weights (stands in for torch.load)
    out["host_model_s"] = time.perf_counter() - t
    from diffroll_amd.engine import Engine
    marks = {}
    orig_init, orig_load = Engine.__init__, Engine.load_params

    def timed_init(self, *a, **k):
        t0 = time.perf_counter()
        orig_init(self, *a, **k)
        marks["create_s"] = time.perf_counter() - t0
        for item in args.tune:
            name, value = item.split("=", 1)
            self.set_option(name, int(value))

    def timed_load(self, params):
        t0 = time.perf_counter()
        orig_load(self, params)
        torch.cuda.synchronize()
        marks["load_params_s"] = time.perf_counter() - t0

    Engine.__init__, Engine.load_params = timed_init, timed_load
    eng = m.engine
    Engine.__init__, Engine.load_params = orig_init, orig_load
    out["create_s"] = marks["create_s"]
    ct = eng.cold_times()
    out["commit_s"] = {"total_incl_param_copies": marks["load_params_s"], "pack": ct[0], "upload": ct[1], "tables": ct[2]}
    T = cfg["L"] // 512
    g = torch.Generator().manual_seed(0)
    wav = (0.1 * torch.randn(cfg["B"], cfg["L"], generator=g)).to(dev)
    x = torch.randn(cfg["B"], 1, T, 88, generator=g).to(dev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    roll, _ = m.sample(x, wav, seed=0)
    roll_host = roll.cpu()
    out["first_sample_s"] = time.perf_counter() - t
    ct = eng.cold_times()
    out["capture_instantiate_s"] = ct[3]
    out["graph_nodes"] = int(ct[4])
    out["process_start_to_first_roll_s"] = proc_age_s()
    ts = []
    for _ in range(args.steady):
        m._fe_key = None
        torch.cuda.synchronize()
        t = time.perf_counter()
        roll, _ = m.sample(x, wav, seed=0)
        roll.cpu()
        ts.append(time.perf_counter() - t)
    out["steady_ms"] = 1e3 * min(ts)
    assert bool(torch.isfinite(roll_host).all())
    if args.json:
        print("COLD_START " + json.dumps(out))
    else:
        for k, v in out.items():
            print(f"{k:32s} {v}")


if __name__ == "__main__":
    main()
