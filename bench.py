#!/usr/bin/env python3
"""Headline benchmark: piano-roll frames/s for a 200-step reverse-diffusion sample.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1..5}]

``--gpus N`` with N > 1 starts the N ranks itself (re-executes under ``python -m torch.distributed.run``, one
process per GPU, RCCL); the same script also runs under an external launcher:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads = BASELINE.json `configs` at their PER-GPU shape; default --config 2, the configuration
the metric is quoted on: ClassifierFreeDiffRoll k=9, C=512, 15 layers, cfdg_ddpm_x0 w=0.5, 200 steps, batch 16 per
GPU of 4 s / 16 kHz synthetic clips (L=64000 -> T=125 frames), fp32, random-init weights.  One "step" of this
benchmark = one whole sample of the local batch: front-end (STFT/mel/normalise/conditioner projections) + the
hipGraph-captured reverse chain + the RCCL gather of the finished rolls + the device->host copy on rank 0.
Inputs are resident in HBM when the timed region starts.  Multi-GPU is weak scaling: every rank runs its own
clips, the only collective is the final all-gather.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel: algorithmic FLOPs per launch / mean launch
duration measured with HIP events on the launch stream in an extra, event-instrumented eager pass of the same
chain run right after the timed region.  `cpu_baseline` is the CPU oracle (a port of the reference math, torch
CPU fp32) timed on this box's host cores on a bounded sample.  `launch_mode` / `per_rank_launch_mode` / `fused_yields` /
`fused_fallbacks` say what every rank's engine launched inside the timed region: a region in which any rank yielded to
per-phase launches or healed a fused time-out is discarded by all ranks and repeated (`attempts`, `discarded_attempts`), at
most twice - then there is no line and the exit code is not 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HP = dict(residual_channels=512, residual_layers=15, kernel_size=9, dilation_base=2, dilation_bound=4,
          n_mels=229, timesteps=200, beta_start=1e-4, beta_end=0.02, sample_rate=16000, n_fft=2048,
          hop_length=512, f_min=0.0, f_max=8000.0)
W_CFG = 0.5
PEAK_MFMA_F32_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: fp32 matrix peak
PEAK_HBM_GBPS = 8000.0

# BASELINE.json configs[0..4] at their per-GPU shape (SURVEY.md 8d "synthetic inputs").  evals = network
# evaluations per reverse step (2 = conditional + unconditional under classifier-free guidance).
CONFIGS = {
    1: dict(k=9, S=50, B=1, L=64000, sampler="cfdg_ddpm_x0", evals=2, gpus_named=1,
            what="configs[0]: k=9, 50 steps, batch 1, 4 s clip, cfdg_ddpm_x0 w=0.5 (the reference's CPU-runnable case)"),
    2: dict(k=9, S=200, B=16, L=64000, sampler="cfdg_ddpm_x0", evals=2, gpus_named=1,
            what="configs[1]: k=9 transcription, cfdg_ddpm_x0 w=0.5, 200 steps, batch 16 per GPU, 4 s @16 kHz clips"),
    3: dict(k=9, S=200, B=16, L=64000, sampler="generation_ddpm_x0", evals=1, gpus_named=8,
            what="configs[2]: k=9 unconditional generation (spec = -1), 200 steps, batch 128 over 8 GPUs = 16 per GPU"),
    4: dict(k=9, S=200, B=16, L=64000, sampler="inpainting_ddpm_x0", evals=2, gpus_named=4,
            what="configs[3]: k=9 inpainting (spectrogram frames [T/4, T/2) masked), w=0.5, 200 steps, batch 64 over "
                 "4 GPUs = 16 per GPU, hipGraph-captured chain"),
    5: dict(k=15, S=200, B=4, L=327680, sampler="cfdg_ddpm_x0", evals=2, gpus_named=8,
            what="configs[4]: k=15, 640-frame segments, cfdg_ddpm_x0 w=0.5, 200 steps, batch 32 over 8 GPUs = 4 per GPU"),
    # the reference's own shipping geometry: sampling.py:27 draws x_T = randn(S, 1, 640, 88) (20.48 s segments) and
    # config/sampling.yaml:11 sets batch_size 4; SURVEY.md 8d asks for config 3 "also at T=640"
    6: dict(k=9, S=200, B=4, L=327680, sampler="cfdg_ddpm_x0", evals=2, gpus_named=1,
            what="reference shipping geometry (sampling.py:27, config/sampling.yaml:11): k=9 transcription, cfdg_ddpm_x0 "
                 "w=0.5, 200 steps, batch 4 of 640-frame (20.48 s) segments"),
    7: dict(k=9, S=200, B=16, L=327680, sampler="generation_ddpm_x0", evals=1, gpus_named=8,
            what="configs[2] at the reference's own generation length (sampling.py:27,45; SURVEY.md 8d): k=9 unconditional "
                 "generation, 200 steps, 640-frame rolls, batch 16 per GPU"),
}
# module-level aliases of the default workload (tools/ import them)
B_LOCAL = CONFIGS[2]["B"]
L_SAMPLES = CONFIGS[2]["L"]
SAMPLER = CONFIGS[2]["sampler"]


def build_model(device, hp=HP, sampler=SAMPLER, w=W_CFG, seed=0, inpainting_t=None):
    from diffroll_amd import ClassifierFreeDiffRoll
    torch.manual_seed(seed)
    m = ClassifierFreeDiffRoll(
        residual_channels=hp["residual_channels"], unconditional=False, condition="fixed",
        n_mels=hp["n_mels"], norm_args=[0, 1, "imagewise"], residual_layers=hp["residual_layers"],
        kernel_size=hp["kernel_size"], dilation_base=hp["dilation_base"], dilation_bound=hp["dilation_bound"],
        spec_args=dict(sample_rate=hp["sample_rate"], n_fft=hp["n_fft"], hop_length=hp["hop_length"],
                       n_mels=hp["n_mels"], f_min=hp["f_min"], f_max=hp["f_max"], center=True,
                       normalized=True, pad_mode="reflect"),
        spec_dropout=0.1, inpainting_t=inpainting_t, timesteps=hp["timesteps"], beta_start=hp["beta_start"],
        beta_end=hp["beta_end"], training={"mode": "x_0"}, sampling={"type": sampler, "w": w}, device=device)
    # the reference zero-initialises the output projection (model/diffwave.py:630): re-draw it so the
    # synthetic network is input dependent
    torch.nn.init.normal_(m.output_projection.weight, 0.0, 0.02)
    m._dirty = True
    return m


def csrc_digest():
    """sha256 over the kernel sources: counter profiles under profiles/ are stamped with it, so a stale
    profile is detected instead of silently reported."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "diffroll_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_tag):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, corrected as MI355X_MICROARCH.md prescribes; tools/refresh_profiles.sh).
    Counters cannot be collected from inside the timed process, so the value comes from the newest committed
    profile - and is reported only when that profile was taken from THESE kernel sources (csrc digest) for THIS
    kernel; otherwise null with the reason."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                j = json.load(f)
        except Exception:       # noqa: BLE001
            continue
        if j.get("kernel_tag") != kernel_tag:
            continue
        if j.get("csrc_digest") != csrc_digest():
            return None, f"{os.path.basename(path)} was taken from other kernel sources (digest {j.get('csrc_digest')})"
        return j["hbm_bytes_per_launch"], os.path.basename(path)
    return None, "no committed PMC profile for this kernel"


def membound_fractions(config):
    """the HBM-roof fraction of every memory-bound kernel of the path at this configuration's
    geometry, from the newest committed kernel-trace profile (tools/membound.sh -> profiles/r*_membound_kernels.json;
    algorithmic bytes per launch / rocprofv3 average duration / 8 TB/s).  At the BASELINE sizes these kernels move
    1-17 MB per launch and a launch costs 2-4 us: they are launch-latency-bound there ('large' = where they flatten)."""
    import glob
    size = {2: "cfg2", 4: "cfg2", 3: "cfg3", 5: "cfg5", 6: "cfg5", 7: "cfg7"}.get(config)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_membound_kernels.json")), reverse=True):
        try:
            with open(path) as f:
                j = json.load(f)
        except Exception:       # noqa: BLE001
            continue
        out = {"source": os.path.basename(path), "peak_gbps": j.get("peak_gbps"),
               "current_sources": j.get("csrc_digest") == csrc_digest()}
        for key in (size, "large"):
            if key and key in j.get("sizes", {}):
                out[key] = {k: {"avg_us": v["avg_us"], "mb": round(v["bytes"] / 1e6, 2), "frac": v["frac"]}
                            for k, v in j["sizes"][key]["kernels"].items()}
        return out
    return None


def flops_per_frame_eval(k, C=512, Lr=15):
    """algorithmic FLOPs per frame per network evaluation (conditioner / embedding hoisted)."""
    return 2 * 88 * C + Lr * (2 * C * 2 * C * k + 2 * C * 2 * C) + 2 * C * C + 2 * C * 88


def executed_flops_per_chain(cfg, T, C=512, Lr=15):
    """FLOPs the engine actually EXECUTES for one chain of the local batch: the algorithmic count (two full
    evaluations per guided step) minus what it legitimately never computes - the first residual layer's dilated conv
    of the unconditional half (same x_t: contracted once per (conditional, unconditional) pair) and the residual half
    of the last layer's 1x1 (model/diffwave.py:678-682 only reads the skip sum after the loop)."""
    B, S, k, ev = cfg["B"], cfg["S"], cfg["k"], cfg["evals"]
    algo = flops_per_frame_eval(k, C, Lr) * B * T * ev * S
    shared_conv = (2 * C * 2 * C * k) * B * T * S if ev == 2 else 0
    unused_res = (2 * C * C) * B * T * ev * S
    return algo - shared_conv - unused_res


def chain_bytes(cfg, T):
    """layer-granular compulsory HBM bytes of one chain on one GPU."""
    C, Lr, k, M = HP["residual_channels"], HP["residual_layers"], cfg["k"], 88
    w_bytes = 4 * (M * C + C + Lr * (2 * C * C * k + 2 * C + 2 * C * C + 2 * C) + C * C + C + C * M + M)
    a_uncond = M * 4 + Lr * (C * 4 + C * 4 + 2 * C * 4) + C * 4 + M * 4      # read h, write h, skip RMW per layer
    a_cond = a_uncond + Lr * 2 * C * 4
    per_frame = {"cfdg_ddpm_x0": a_cond + a_uncond, "inpainting_ddpm_x0": a_cond + a_uncond,
                 "generation_ddpm_x0": a_uncond}[cfg["sampler"]]
    return cfg["S"] * (cfg["evals"] * w_bytes + cfg["B"] * T * (per_frame + 3 * M * 4))


def cold_start(configs=(1, 2)):
    """Time-to-first-roll of a one-shot process (the reference's product: sampling.py:22-73 loads a checkpoint and runs ONE
    trainer.predict): `python -m diffroll_amd.coldstart --config N` in a FRESH process per configuration - engine creation,
    dr_commit (packing / uploads / tables), the first sample (front-end + graph capture + instantiate + chain) and the steady
    chain next to them.  Runs after the timed region, on the same device (this process is idle meanwhile)."""
    import subprocess
    out = {}
    for c in configs:
        try:
            r = subprocess.run([sys.executable, "-m", "diffroll_amd.coldstart", "--config", str(c), "--json"], cwd=ROOT,
                               capture_output=True, text=True, timeout=300)
            line = next(ln for ln in r.stdout.splitlines() if ln.startswith("COLD_START "))
            j = json.loads(line[len("COLD_START "):])
            out[f"config{c}"] = {
                "create_s": round(j["create_s"], 3),
                "commit_s": round(j["commit_s"]["total_incl_param_copies"], 3),
                "commit_split_s": {k: round(v, 3) for k, v in j["commit_s"].items() if k != "total_incl_param_copies"},
                "first_sample_s": round(j["first_sample_s"], 3),
                "capture_instantiate_s": round(j["capture_instantiate_s"], 3), "graph_nodes": j["graph_nodes"],
                "steady_ms": round(j["steady_ms"], 2),
                "import_torch_s": round(j["import_torch_s"], 2), "hip_init_s": round(j["hip_init_s"], 2),
                "host_model_s": round(j["host_model_s"], 2),
                "process_start_to_first_roll_s": round(j["process_start_to_first_roll_s"], 2) if j.get("process_start_to_first_roll_s") else None,
            }
        except Exception as ex:       # noqa: BLE001 - reported, never allowed to fail the measurement
            out[f"config{c}"] = {"error": f"{type(ex).__name__}: {ex}"[:200]}
    return out


def cpu_baseline(model, cfg, hp, budget_s=12.0, max_steps=40):
    """The oracle (CPU port of the reference arithmetic) on this box's host cores, bounded sample."""
    from oracle import diffroll_ref as R           # checker / baseline only - never the product path
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    B, Ls, sampler = cfg["B"], cfg["L"], cfg["sampler"]
    T = Ls // hp["hop_length"]
    S = hp["timesteps"]
    g = torch.Generator().manual_seed(1)
    wav = 0.1 * torch.randn(B, Ls, generator=g)
    x = torch.randn(B, 1, T, 88, generator=g)
    sch = R.schedule(hp["beta_start"], hp["beta_end"], S)
    table = R.build_embedding(S)
    default_threads = torch.get_num_threads()
    with torch.no_grad():
        spec = R.frontend(wav, hp, T)
        # Be fair to the CPU: torch's default thread count (one per hardware thread it sees) can be far from the best
        # one inside a container with a CPU quota (measured on the GPU box: 0.23 s per step on 16 threads, 1.95 s on
        # 128, 118 s on 256).  Short ascending sweep, one step per candidate, stop once it clearly gets worse.
        z0 = torch.randn(x.shape, generator=g)
        best_n, best_t = default_threads, float("inf")
        for cand in sorted({c for c in (4, 8, 16, 32, 64, 128, default_threads) if c <= (os.cpu_count() or 1)}):
            torch.set_num_threads(cand)
            t0 = time.perf_counter()
            R.reverse_step(params, hp, sch, sampler, x, spec, S - 1, z0, W_CFG, table)
            dt_c = time.perf_counter() - t0
            if dt_c < best_t:
                best_n, best_t = cand, dt_c
            elif dt_c > 1.5 * best_t:
                break
        torch.set_num_threads(best_n)
        cores = best_n
        t0 = time.perf_counter()
        spec = R.frontend(wav, hp, T)
        t_front = time.perf_counter() - t0
        n = 0
        t0 = time.perf_counter()
        while n < min(max_steps, S) and (n == 0 or time.perf_counter() - t0 < budget_s):
            t_index = S - 1 - n
            z = torch.randn(x.shape, generator=g)
            x = R.reverse_step(params, hp, sch, sampler, x, spec, t_index, z, W_CFG, table)
            n += 1
        t_steps = time.perf_counter() - t0
        # BASELINE.md section 3 also asks for the single-thread figure: one reverse step of ONE clip, scaled to the
        # batch (the per-clip work is independent), extrapolated like the multi-thread figure
        torch.set_num_threads(1)
        try:
            t0 = time.perf_counter()
            R.reverse_step(params, hp, sch, sampler, x[:1], spec[:1], S - 1, z[:1], W_CFG, table)
            t_one = time.perf_counter() - t0
            # ... and the N = all figure (every hardware thread the box reports): the same one-clip step (the whole batch at
            # 256 threads inside this container's CPU quota took 118 s per step when it was tried - the sweep above exists
            # because of that), scaled to the batch like the single-thread figure
            n_all = os.cpu_count() or default_threads
            torch.set_num_threads(n_all)
            t0 = time.perf_counter()
            R.reverse_step(params, hp, sch, sampler, x[:1], spec[:1], S - 1, z[:1], W_CFG, table)
            t_all = time.perf_counter() - t0
        finally:
            torch.set_num_threads(default_threads)
    per_step = t_steps / n
    total = t_front + per_step * S
    model_name = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model_name = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": round(B * T / total, 3), "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": f"{n} of {S} reverse steps ({cfg['evals']} network evaluation(s) each) at B={B},T={T} + one "
                  f"front-end, {t_steps + t_front:.1f} s of CPU work, extrapolated to the {S}-step chain",
        "os_cpu_count": os.cpu_count(), "cpu_model": model_name, "s_per_step": round(per_step, 4),
        "threads_note": f"thread count chosen by a short ascending sweep over 4..128 (os.cpu_count() = {os.cpu_count()})",
        "single_thread": {"value": round(T / (t_one * S), 3), "unit": "frames/s", "cores": 1,
                          "sample": f"1 reverse step of 1 clip ({t_one:.1f} s), extrapolated to {S} steps"},
        "all_threads": {"value": round(T / (t_all * S), 3), "unit": "frames/s", "cores": n_all,
                        "sample": f"1 reverse step of 1 clip ({t_all:.1f} s) with torch.set_num_threads(os.cpu_count() = {n_all}), "
                                  f"extrapolated to {S} steps (BASELINE.md section 3: N = all)"},
    }


def rank_state(before, after, expect_fused, mode_code):
    """One rank's verdict words for a timed region, from dr_launch_state before and after it: [fused time-outs healed inside
    it, yields inside it, launch mode code, fusing throughout (or never meant to: an A/B run with the fused launches switched
    off on purpose), yields since engine creation]."""
    fusing = (not expect_fused) or (before["fused_enabled"] != 0 and after["fused_enabled"] != 0)
    return [after["fallbacks"] - before["fallbacks"], after["yields"] - before["yields"], mode_code[after["mode"]], int(fusing),
            after["yields"]]


def degraded_ranks(states, share_gpu):
    """[(rank, words)] of the ranks whose timed region is no measurement of the fused engine (collective verdict: one of them
    is enough to discard the attempt on every rank).  --share-gpu runs ask for per-phase launches themselves: exempt."""
    if share_gpu:
        return []
    return [(r, sx) for r, sx in enumerate(states) if sx[0] or sx[1] or not sx[3]]


def _modes():
    from diffroll_amd import _cabi
    return dict(_cabi.MODES)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json config (1-based) at its per-GPU shape; default 2 = the headline workload")
    ap.add_argument("--dist", action="store_true",
                    help="plain single-process run: still create a 1-rank RCCL group (exercises init / all-gather / barrier)")
    ap.add_argument("--gather-check", action="store_true",
                    help="multi-rank runs: also create a C-ABI communicator (dr_comm_create) and compare dr_gather with "
                         "torch's all-gather (always done in 1-rank groups)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--accumulation", choices=("auto", "blocked", "single_chain"), default="auto",
                    help="accumulation order of the dilated conv: auto = blocked = one fp32 chain per 32-channel chunk in every fp32 "
                         "flavour (default); single_chain = 128-frame blocks keep one chain over all of K (-0.5 %%; DESIGN.md 2)")
    ap.add_argument("--no-cold-start", action="store_true", help="skip the time-to-first-roll subprocesses (configs 1 and 2)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-split", action="store_true", help="skip the extra bf16x3 split-precision measurement")
    ap.add_argument("--share-gpu", action="store_true",
                    help="PLUMBING TEST of the N > 1 path on a 1-GPU box: all ranks bind device 0 over a gloo group with one "
                         "launch per phase (no persistent kernels); the line carries dist.share_gpu = true and is no measurement")
    args = ap.parse_args()
    if args.share_gpu:
        os.environ["DR_BENCH_SHARE_GPU"] = "1"
    # A/B runs pin engine options for the process through DR_TEST_TUNE (tools/tuning_env.py); the line says so ("tuning")
    from tools import tuning_env
    forced_options = tuning_env.install()

    from diffroll_amd import launch
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not launch.under_launcher() and args.gpus > 1:
        # `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU over RCCL)
        sys.exit(launch.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    rank, world, local_rank = launch.rank_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the engine has no CPU fallback)")
    if launch.share_gpu():
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # host-side torch ops here are tiny; keep N ranks from each spawning one CPU thread per hardware thread
    torch.set_num_threads(min(16, torch.get_num_threads()))
    dist = launch.init_process_group(device, force_single=args.dist)
    cdev = launch.collective_device(dist, device)

    from diffroll_amd.distributed import gather_rolls

    cfg = CONFIGS[args.config]
    hp = dict(HP)
    hp.update(kernel_size=cfg["k"], timesteps=cfg["S"])
    B, Ls, S, sampler = cfg["B"], cfg["L"], cfg["S"], cfg["sampler"]
    T = Ls // hp["hop_length"]
    inp_t = [T // 4, T // 2] if sampler == "inpainting_ddpm_x0" else None
    model = build_model(device, hp=hp, sampler=sampler, inpainting_t=inp_t)
    model.accumulation = args.accumulation
    # synthetic inputs, resident in HBM; global sample index = rank * B + b
    g = torch.Generator().manual_seed(1000 + rank)
    wav = (0.1 * torch.randn(B, Ls, generator=g)).to(device)
    x_T = torch.randn(B, 1, T, 88, generator=g).to(device)
    model.engine   # create + commit (weight packing / upload) outside the timed region
    if launch.share_gpu():
        # several processes time-share the one device: persistent kernels assume all their workgroups resident
        model.engine.set_option("fused_stack", 0)
        model.engine.set_option("fused_tail", 0)

    def one_step():
        model._fe_key = None                                   # front-end is part of every sample
        roll, _ = model.sample(x_T, wav, seed=0, first_sample=rank * B)   # on-device Philox noise
        full = gather_rolls(roll)                              # RCCL all-gather (no-op without a process group)
        if rank == 0:
            return full.cpu()
        return None

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n):
        sync()
        t0 = time.perf_counter()
        o = None
        for _ in range(n):
            o = one_step()
        sync()
        dt_ = time.perf_counter() - t0
        tt = torch.tensor([dt_], device=cdev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item()), o

    fake_kfd = os.environ.get("DR_BENCH_FAKE_KFD")
    if fake_kfd:
        # test hook (tests/test_gpu_sharding.py): show the engine a KFD process list with a busy co-tenant that does not
        # exist, so that it yields - and this script must then refuse to print a line
        from diffroll_amd import _cabi
        _cabi.load_library().dr_debug_kfd_root(fake_kfd.encode())
    # What did the engine actually launch?  A fused launch that timed out and was healed (fallbacks), or an engine that
    # YIELDED to per-phase launches because it believed the GPU shared (yields: at creation, in the warm-up or in the timed
    # region), still returns the right rolls - but the time is then no measurement of the engine this line describes.  The
    # verdict is collective (one rank's yield bends the max-over-ranks time of everybody): every rank learns it.  An
    # attempt whose timed region saw a time-out or a yield on ANY rank, or started on an engine that was not fusing, is
    # DISCARDED by all ranks together and repeated (a yielded engine re-arms after two clean looks at the process list:
    # the repeat's warm-up takes them, >= 250 ms apart) - at most MAX_ATTEMPTS times; then no rank prints and every rank
    # exits non-zero.  --share-gpu runs ask for per-phase launches themselves and are exempt.
    MAX_ATTEMPTS = 3
    mode_code = {v: k for k, v in _modes().items()}
    st0 = model.engine.launch_state()
    # (an engine whose fused launches were switched off on purpose - an A/B run, DR_TEST_TUNE=fused_stack=0 - is not "degraded")
    expect_fused = st0["fused_enabled"] != 0 or st0["yields"] > 0
    discarded = []
    for attempt in range(1, MAX_ATTEMPTS + 1):
        for _ in range(args.warmup if attempt == 1 else max(args.warmup, 3)):
            if attempt > 1:
                time.sleep(0.3)
            one_step()
        st1 = model.engine.launch_state()
        dt, out = timed(args.steps)
        st = model.engine.launch_state()
        mine = torch.tensor(rank_state(st1, st, expect_fused, mode_code), device=cdev, dtype=torch.int64)
        if dist is not None:
            allst = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allst, mine)
            states = [[int(v) for v in t.tolist()] for t in allst]
        else:
            states = [[int(v) for v in mine.tolist()]]
        modes = [_modes()[sx[2]] for sx in states]
        bad = degraded_ranks(states, launch.share_gpu())
        if not bad:
            break
        why = "; ".join(f"rank {r}: {sx[0]} fused time-out(s) and {sx[1]} yield(s) to per-phase launches in the timed region, "
                        f"{'fusing' if sx[3] else 'NOT fusing (yielded earlier)'}, {sx[4]} yield(s) since creation" for r, sx in bad)
        discarded.append({"attempt": attempt, "ms_per_step": round(1e3 * dt / args.steps, 3), "per_rank_launch_mode": modes,
                          "ranks": {str(r): {"timeouts": sx[0], "yields": sx[1], "fusing": bool(sx[3])} for r, sx in bad}})
        if rank == 0:
            print(f"[bench] attempt {attempt} of {MAX_ATTEMPTS} discarded - {why}", file=sys.stderr, flush=True)
        if fake_kfd and os.environ.get("DR_BENCH_FAKE_KFD_THEN"):
            # (test hook: the made-up co-tenant stops computing after the first discarded attempt)
            _cabi.load_library().dr_debug_kfd_root(os.environ["DR_BENCH_FAKE_KFD_THEN"].encode())
        if attempt == MAX_ATTEMPTS:
            raise SystemExit(f"rank {rank}: no benchmark line - {why} (is something else using the GPU(s)?  see dr_launch_state "
                             "/ csrc/tenants.h)")

    frames = world * B * T * args.steps
    result = {
        "metric": "piano-roll frames/sec (200-step sample)" if S == 200 else f"piano-roll frames/sec ({S}-step sample)",
        "value": round(frames / dt, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE {cfg['what']} (T={T}); ClassifierFreeDiffRoll C=512 L=15, random-init "
                               "weights, Philox noise, front-end + chain + gather + D2H per step",
                   "baseline_config": args.config, "batch_per_gpu": B, "frames_per_clip": T, "diffusion_steps": S,
                   "kernel_size": cfg["k"], "sampler": sampler, "w": W_CFG if cfg["evals"] == 2 else None,
                   "inpainting_t": inp_t, "parallelism": f"batch-shard x{world}", "graph": True,
                   "conv_accumulation": "blocked" if args.accumulation == "auto" else args.accumulation},
        "dist": launch.dist_info(dist),
        # per rank: fused time-outs healed inside the timed region / yields since engine creation (both 0 or there is no
        # line, except under --share-gpu) and how the residual layers were launched (dr_launch_state)
        "fused_fallbacks": sum(sx[0] for sx in states), "fused_yields": sum(sx[1] for sx in states),
        "launch_mode": modes[0] if len(set(modes)) == 1 else "mixed", "per_rank_launch_mode": modes,
        # attempts: 1 unless timed regions were discarded (and why); yields_since_creation: per rank, warm-ups included
        "attempts": len(discarded) + 1, "discarded_attempts": discarded, "yields_since_creation": [sx[4] for sx in states],
    }
    if forced_options:
        result["tuning"] = dict(forced_options)      # NOT the shipping configuration: an A/B run
    # straggler visibility for the scaling table: every rank's own time over the same K steps (no barrier inside),
    # and the final all-gather on its own (HIP events around 10 back-to-back gathers of the finished rolls)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    mine = torch.tensor([1e3 * (time.perf_counter() - t0) / args.steps], device=cdev, dtype=torch.float64)
    if dist is not None:
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(v.item()) for v in allr]
    else:
        per_rank = [float(mine.item())]
    roll_l, _ = model.sample(x_T, wav, seed=0, first_sample=rank * B)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gather_rolls(roll_l)
    sync()
    e0.record()
    for _ in range(10):
        gather_rolls(roll_l)
    e1.record()
    torch.cuda.synchronize()
    result["per_rank_ms_per_step"] = {"min": round(min(per_rank), 3), "max": round(max(per_rank), 3),
                                      "all": [round(v, 3) for v in per_rank]}
    result["gather_us"] = round(1e3 * e0.elapsed_time(e1) / 10, 2) if dist is not None else 0.0
    if rank == 0:
        assert out is not None and bool(torch.isfinite(out).all()) and out.shape == (world * B, 1, T, 88)
        # the metric also asks for the HBM-roofline fraction: SURVEY.md 8(d) algorithmic bytes of one chain per GPU
        # (weights streamed once per evaluation + layer-granular activation traffic + the update) over the chain time
        cb = chain_bytes(cfg, T)
        gbps = cb / (dt / args.steps) / 1e9
        fl = flops_per_frame_eval(cfg["k"]) * B * T * cfg["evals"] * S
        result["hbm_roofline"] = {"algorithmic_bytes_per_chain_per_gpu": cb, "achieved_gbps_per_gpu": round(gbps, 1),
                                  "peak_gbps": PEAK_HBM_GBPS, "frac": round(gbps / PEAK_HBM_GBPS, 4),
                                  "note": "the step is MFMA-bound in fp32 (see roofline); reported because the metric names it"}
        fx = executed_flops_per_chain(cfg, T)
        per = dt / args.steps
        result["whole_chain"] = {
            "algorithmic_tflop_per_chain_per_gpu": round(fl / 1e12, 2),
            "algorithmic_tflops_per_gpu": round(fl / per / 1e12, 2),
            "algorithmic_frac_of_fp32_mfma_peak": round(fl / per / 1e12 / PEAK_MFMA_F32_TFLOPS, 4),
            "executed_tflop_per_chain_per_gpu": round(fx / 1e12, 2),
            "executed_tflops_per_gpu": round(fx / per / 1e12, 2),
            "executed_frac_of_fp32_mfma_peak": round(fx / per / 1e12 / PEAK_MFMA_F32_TFLOPS, 4),
            "note": "algorithmic = SURVEY.md 8d count (two full evaluations per guided step); executed = minus the shared "
                    "first-layer contraction and the never-read residual half of the last 1x1 (work the engine does not do)"}

    if rank == 0 and not args.no_roofline:
        eng = model.engine
        eng.profile_enable(True)
        model._fe_key = None
        model.sample(x_T, wav, seed=0, first_sample=0, use_graph=False)
        launches, ms, flops, kname = eng.profile_read_ex(reset=True)
        eng.profile_enable(False)
        # the timed kernel is the engine's dominant one for this launch geometry: the fused residual-stack kernel
        # (all residual layers of one evaluation in one persistent launch) or, where that does not apply, the
        # dilated conv + gate kernel.  flops = SURVEY.md 8(d) per-frame figures x the frames each launch processed.
        avg_s = ms * 1e-3 / max(launches, 1)
        achieved = flops / max(ms * 1e-3, 1e-12) / 1e12
        tag = "stack" if kname.startswith("stack_kernel") else "conv_gate"
        traffic, traffic_src = pmc_traffic(f"{tag}:config{args.config}")
        result["roofline"] = {
            "bound": "mfma", "kernel": kname,
            "achieved": round(achieved, 2), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_MFMA_F32_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "launches": launches, "avg_launch_us": round(avg_s * 1e6, 2),
            "flops_per_launch": flops / max(launches, 1),
            "share_of_step_time": round((ms * 1e-3) / (dt / args.steps), 4),
        }
    if rank == 0:
        result["memory_bound_kernels"] = membound_fractions(args.config)
    if not args.no_split:
        # Opt-in split-bf16 precision (DR_PRECISION_BF16X3: every fp32 operand split exactly into three bf16
        # pieces, six piece products on the bf16 MFMA, fp32 accumulation - fp32-level error).
        # Reported NEXT TO the headline, never as `value`: same workload, same seeds, same timing protocol.
        model.precision = "bf16x3"
        one_step()
        dt3, out3 = timed(args.steps)
        model.precision = "f32"
        model.engine
        if rank == 0:
            result["split_bf16x3"] = {
                "value": round(frames / dt3, 2), "unit": "frames/s", "ms_per_step": round(1e3 * dt3 / args.steps, 3),
                "max_abs_diff_vs_f32_roll": float((out3 - out).abs().max()),
                "thresholded_frames_differing": int(((out3 > 0.5) != (out > 0.5)).sum()),
                "note": "opt-in precision mode; identical inputs and Philox noise as the headline run",
            }
    if dist is not None and (args.gather_check or world == 1):
        # (multi-rank runs: only with --gather-check - a second communicator is created collectively, and the headline
        # measurement must never depend on it)
        # the same collective through the C-ABI (dr_comm_create / dr_gather: RCCL via dlopen, no torch.distributed in
        # the data path) - run once next to the timed region and compared with torch's all-gather; reported, never
        # allowed to fail the measurement
        check = {}
        try:
            from diffroll_amd.distributed import NativeComm
            comm = NativeComm(device)
            roll, _ = model.sample(x_T, wav, seed=0, first_sample=rank * B)
            a = gather_rolls(roll)
            b = gather_rolls(roll, comm=comm)
            torch.cuda.synchronize()
            check = {"ok": bool(torch.equal(a, b)), "ranks": comm.world_size, "rccl_version_code": comm.rccl_version()}
            comm.close()
        except Exception as ex:       # noqa: BLE001
            check = {"ok": False, "error": f"{type(ex).__name__}: {ex}"[:300]}
        if rank == 0:
            result["dr_gather_check"] = check
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(model, cfg, hp)
    if rank == 0 and world == 1 and not args.no_cold_start:
        result["cold_start"] = cold_start()
    seen = result["dist"]["ranks_seen"]
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if seen != args.gpus or len(per_rank) != args.gpus:
        # a job that ran with fewer ranks than it was asked for must not pass for a scaling point
        raise SystemExit(f"rank {rank}: --gpus {args.gpus} but the process group had {seen} rank(s) "
                         f"({len(per_rank)} per-rank times)")


if __name__ == "__main__":
    main()
