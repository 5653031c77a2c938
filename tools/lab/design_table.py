#!/usr/bin/env python3
"""Rewrite the numeric cells of DESIGN.md's per-configuration table from profiles/<round>_bench_cfgN.json (the narrative
parts of the rows stay as they are):    python tools/lab/design_table.py [r03]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PREFIX = {1: '| 1: k=9, 50 steps, B=1', 2: '| 2: k=9, 200 steps, B=16', 3: '| 3/GPU: k=9, generation', 4: '| 4/GPU: k=9, inpainting',
          5: '| 5/GPU: k=15, B=4', 6: '| 6: **the reference', 7: '| 7: config 3 at the reference'}


def main():
    rd = sys.argv[1] if len(sys.argv) > 1 else "r03"
    path = os.path.join(ROOT, "DESIGN.md")
    lines = open(path).read().split("\n")
    for c, pre in PREFIX.items():
        j = json.loads(open(os.path.join(ROOT, "profiles", f"{rd}_bench_cfg{c}.json")).read().strip().splitlines()[-1])
        r, w, h = j["roofline"], j["whole_chain"], j["hbm_roofline"]
        idx = [i for i, ln in enumerate(lines) if ln.startswith(pre)]
        assert len(idx) == 1, pre
        cells = lines[idx[0]].split(" | ")
        m = re.match(r"^([0-9.]+)(.*)$", cells[1])
        cells[1] = f"{j['ms_per_step']:.1f}" + m.group(2)
        cells[2] = f"{j['value']:.0f}"
        cells[3] = re.sub(r"(0\.\d{2,3})(?!.*0\.\d{2,3})", f"{r['frac']:.3f}", cells[3], count=1)
        cells[4] = re.sub(r"^[0-9.]+ \([0-9.]+ %\) / [0-9.]+ \([0-9.]+ %\)",
                          f"{w['executed_tflops_per_gpu']:.1f} ({100 * w['executed_frac_of_fp32_mfma_peak']:.1f} %) / "
                          f"{w['algorithmic_tflops_per_gpu']:.1f} ({100 * w['algorithmic_frac_of_fp32_mfma_peak']:.1f} %)", cells[4])
        cells[6] = re.sub(r"^[0-9.]+ \([0-9.]+ %\)", f"{h['achieved_gbps_per_gpu']:.0f} ({100 * h['frac']:.1f} %)", cells[6])
        lines[idx[0]] = " | ".join(cells)
    open(path, "w").write("\n".join(lines))


if __name__ == "__main__":
    main()
