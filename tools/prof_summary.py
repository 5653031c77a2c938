#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace and/or counter collection) into a small text table.
    python tools/prof_summary.py <dir> <prefix> [title]"""
import collections
import csv
import glob
import os
import sys


def main():
    d, prefix = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else ""
    out = [f"# {title}"] if title else []
    kt = glob.glob(os.path.join(d, "**", f"{prefix}_kernel_trace.csv"), recursive=True)
    if kt:
        rows = list(csv.DictReader(open(kt[0])))
        agg = collections.defaultdict(list)
        for r in rows:
            agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        tot = sum(sum(v) for v in agg.values())
        out.append(f"{'calls':>7} {'total_us':>12} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
        for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            out.append(f"{len(v):7d} {sum(v) / 1e3:12.1f} {sum(v) / len(v) / 1e3:9.2f} {min(v) / 1e3:9.2f} "
                       f"{max(v) / 1e3:9.2f} {100 * sum(v) / tot:6.2f}  {n[:140]}")
    cc = glob.glob(os.path.join(d, "**", f"{prefix}_counter_collection.csv"), recursive=True)
    if cc:
        rows = list(csv.DictReader(open(cc[0])))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out.append("")
        out.append("# counters: mean per dispatch")
        for k, cs in sorted(agg.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
            out.append(f"{k[:140]}")
            for c, v in sorted(cs.items()):
                out.append(f"    {c:28s} n={len(v):6d} mean={sum(v) / len(v):16.1f}")
    print("\n".join(out))


if __name__ == "__main__":
    main()
