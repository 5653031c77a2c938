#!/usr/bin/env python3
"""Combine the two rocprofv3 TCC counter passes (FETCH_SIZE; WRITE_SIZE + TCC_HIT/MISS) of tools/step_loop.py into the
stamped traffic record bench.py reads:  python tools/make_traffic_json.py <fetch_dir> <write_dir> <config> <out.json>
Correction as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE (KB) x 2 for wide coalesced reads;
WRITE_SIZE taken as is (uncalibrated)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def counters(d):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch_dir, write_dir, config, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    cf, cw = counters(fetch_dir), counters(write_dir)
    # the dominant kernel of the run: the fused stack where it is used, else the conv + gate kernel (EPI 3) of either
    # MFMA flavour
    import re
    kern = max((k for k in cf if "stack_kernel" in k or re.search(r"gemm_kernel<\d, 1, 3, 0(?:, \d)?>|gemm16_kernel<\d, 1, 3>", k)),
               key=lambda k: sum(cf[k]["FETCH_SIZE"]))
    mean = lambda v: sum(v) / len(v)      # noqa: E731
    fetch_kb = mean(cf[kern]["FETCH_SIZE"])
    write_kb = mean(cw[kern]["WRITE_SIZE"])
    hit, miss = mean(cw[kern]["TCC_HIT_sum"]), mean(cw[kern]["TCC_MISS_sum"])
    cfg = bench.CONFIGS[config]
    C, Lr, k = 512, 15, cfg["k"]
    T = cfg["L"] // 512
    NB = cfg["B"] * cfg["evals"]
    fused = "stack_kernel" in kern
    # Two denominators, both per LAUNCH of that kernel, weights counted ONCE per launch (every evaluation of the launch - the
    # 2B batch of a guided step - streams the same panels):
    #  * algorithmic (layer-granular): weights of the phases it runs + per frame: conv reads hd (2048) +
    #    conditioner (4096, conditional samples) and writes g (2048); the 1x1 reads g (2048), read-modify-writes h (2 x 2048)
    #    and skip (2 x 2048) and writes hd (2048) - what a layer-at-a-time implementation must move;
    #  * fused lower bound (fused launches only): what THIS kernel must move if every exchange between its phases stayed
    #    on chip - weights + conditioner once, the h / skip tile read once and written once (it lives in LDS in between);
    #    g and hd are exchanged between workgroups inside the launch (8 MB each per layer at config 2: L2 / Infinity-Cache
    #    sized) and count as zero.  The counters sit at the L2 -> fabric boundary: what the Infinity Cache serves is in them.
    conv_w, pw_w = 4 * (2 * C * C * k + 2 * C), 4 * (2 * C * C + 2 * C)
    frames, cond_frames = NB * T, cfg["B"] * T if cfg["sampler"] != "generation_ddpm_x0" else 0
    conv_a = frames * (2048 + 2048) + cond_frames * 4096
    pw_a = frames * (2048 + 4096 + 4096 + 2048)
    lower = None
    if fused:
        n_conv = Lr - 1 if cfg["evals"] == 2 else Lr          # layer 0's conv is its own launch under guidance
        w_bytes = n_conv * conv_w + Lr * pw_w
        a_bytes = n_conv * conv_a + Lr * pw_a
        lower = w_bytes + n_conv * cond_frames * 4096 + frames * 2 * (2048 + 2048)
        tag = "stack"
    else:
        w_bytes, a_bytes = conv_w, conv_a          # (launches of layer 0 under guidance contract half the samples: a lower bound there)
        tag = "conv_gate"
    algo = w_bytes + a_bytes
    rec = {
        "kernel": kern[:160], "kernel_tag": f"{tag}:config{config}", "csrc_digest": bench.csrc_digest(),
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes, --kernel-trace) "
                  f"-- python tools/step_loop.py --config {config} --iters 10; means over the launches of that kernel",
        "FETCH_SIZE_KB_per_launch": round(fetch_kb, 1), "WRITE_SIZE_KB_per_launch": round(write_kb, 1),
        "TCC_HIT_per_launch": round(hit, 1), "TCC_MISS_per_launch": round(miss, 1),
        "l2_hit_rate": round(hit / max(hit + miss, 1), 4),
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) "
                      "coalesced reads -> doubled; WRITE_SIZE uncalibrated, taken as is",
        "hbm_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024),
        "algorithmic_bytes_per_launch": int(algo),
        "algorithmic_split": {"weights_once_per_launch": int(w_bytes), "activations_layer_granular": int(a_bytes)},
        "fused_lower_bound_bytes_per_launch": int(lower) if lower is not None else None,
        "counter_boundary": "TCC FETCH / WRITE count requests leaving the XCDs' L2s towards the fabric: reads the 256 MB Infinity "
                            "Cache serves (the same weight panel pulled by all eight XCDs, g / hd exchanged between workgroups) "
                            "are counted although they never reach HBM",
    }
    rec["traffic_over_algorithmic"] = round(rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"], 3)
    if lower:
        rec["traffic_over_fused_lower_bound"] = round(rec["hbm_bytes_per_launch"] / lower, 3)
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
