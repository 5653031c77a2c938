#!/usr/bin/env python3
"""Aggregate a DR_PARITY_LOG file (one line per maxdiff() comparison of the GPU suite, plus the trained-regime battery's
records) into the table committed under profiles/:  python tools/lab/margins_summary.py <margins.txt> > profiles/rNN_parity_margins.txt"""
import collections
import re
import statistics
import sys


def main():
    path = sys.argv[1]
    per = collections.defaultdict(list)
    trained = []
    for line in open(path):
        line = line.strip()
        if not line:
            continue
        if line.startswith("trained_regime["):
            trained.append(line)
            continue
        name, val = line.rsplit(" ", 1)
        per[name.split(":")[0]].append(float(val))
    print("# Observed HIP-vs-oracle / HIP-vs-reference-vector differences of the whole GPU suite on one MI355X")
    print("# (DR_PARITY_LOG=<file> python -m pytest tests -m gpu: every maxdiff() comparison, aggregated per test function;")
    print("#  `prop` = the hypothesis-driven property of tests/test_gpu_r3.py).")
    print("# Tolerances: evaluation / step / chain 1e-5 (x the output range where a test says so), normalised log-mel 4e-5,")
    print("# shard geometry 1e-5, FFT power spectrum 2e-6 of the clip's largest bin.")
    print(f"{'test':66s} {'n':>4s} {'max':>10s} {'median':>10s}")
    for name, v in sorted(per.items(), key=lambda kv: -max(kv[1])):
        print(f"{name:66s} {len(v):4d} {max(v):10.2e} {statistics.median(v):10.2e}")
    if trained:
        print()
        print("# Trained-weight regime (tests/test_gpu_r3.py: dilated-conv / conditioner weights x s_conv, 1x1 weights x s_out;")
        print("# 5-layer full-width net): error against a FLOAT64 evaluation of the oracle, next to the fp32 oracle's own error.")
        print("# The bound asserted: err_hip <= b x err_fp32_oracle + 5e-6 x range, b = 2.5 where the dilated conv accumulates\n# in blocks (every flavour the default options select), 6 where a flavour keeps one chain over all of K (DR_BLOCKED=1 =\n# third column 'single_chain': 128-frame blocks, and the 16x16-MFMA conv kernels of the 96 / 160-frame widths).")
        worst = collections.defaultdict(lambda: (0.0, ""))
        for line in trained:
            tag = line.split("]")[0] + "]"
            f = line.split()
            rng, e32, ehip = float(f[2]), float(f[4]), float(f[6])
            key = tag.split(",")[0].replace("trained_regime[", "") + "," + tag.split(",")[1]
            m = re.search(r"(?:acc|conv)=(\w+)", tag)
            key += "," + (m.group(1) if m else "auto")
            ratio = ehip / max(e32, 1e-30)
            if ratio > worst[key][0]:
                worst[key] = (ratio, f"range {rng:.2e} err_fp32_oracle {e32:.2e} err_hip {ehip:.2e}  {tag}")
        print(f"{'precision, scaling, accum.':28s} {'worst err_hip / err_fp32_oracle':>32s}   where")
        for key, (ratio, where) in sorted(worst.items()):
            print(f"{key:28s} {ratio:32.2f}   {where}")
        print(f"# {len(trained)} (geometry, conditional / unconditional) evaluations in all")


if __name__ == "__main__":
    main()
