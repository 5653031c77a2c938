#!/usr/bin/env python3
"""the HBM-roof fraction of every memory-bound kernel of the path, from rocprofv3 kernel traces of
tools/membound_loop.py at the BASELINE geometries and one far beyond them.
    python tools/membound_report.py <dir with one sub-directory per size> <out.json> > table.txt
(run by tools/membound.sh on the GPU box)."""
import collections
import csv
import glob
import json
import os
import sys

PEAK = 8000.0      # GB/s, MI355X_MICROARCH.md
COPY_CEILING = 0.79


def main():
    root, out_json = sys.argv[1], sys.argv[2]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    rec = {"peak_gbps": PEAK, "csrc_digest": bench.csrc_digest(), "sizes": {}}
    print("# HBM-roof fraction of the memory-bound kernels (SURVEY 8d item 2): ALGORITHMIC bytes per launch / rocprofv3 --kernel-trace")
    print("# average duration / 8 TB/s.  The guide's plain-copy ceiling is 0.79; an empty kernel launch shows as ~2-4 us in the same traces")
    print("# (set_dyn_kernel, one thread), i.e. a kernel moving less than ~10 MB cannot reach 0.3 of the roof whatever it does.")
    for d in sorted(glob.glob(os.path.join(root, "*"))):
        logs = glob.glob(os.path.join(d, "loop.log"))
        kts = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
        if not logs or not kts:
            continue
        line = [ln for ln in open(logs[0]) if ln.startswith("MEMBOUND_GEOMETRY ")]
        if not line:
            continue
        geo = json.loads(line[-1][len("MEMBOUND_GEOMETRY "):])
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(kts[0])):
            agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        floor = [v for k, v in agg.items() if "set_dyn_kernel" in k]
        print(f"\n## {geo['size']}: {geo['what']}")
        print(f"{'kernel':24s} {'launches':>8} {'avg_us':>9} {'min_us':>9} {'MB/launch':>10} {'GB/s':>9} {'frac of 8 TB/s':>14}")
        rows = {}
        for name, nbytes in geo["bytes"].items():
            ks = [v for k, v in agg.items() if name in k]
            if not ks:
                continue
            v = sorted(ks[0])
            v = v[: max(1, len(v) - len(v) // 10)]            # drop the slowest tenth (first-launch effects)
            avg = sum(v) / len(v) / 1e3
            gbps = nbytes / (avg * 1e-6) / 1e9
            rows[name] = {"launches": len(ks[0]), "avg_us": round(avg, 2), "min_us": round(v[0] / 1e3, 2), "bytes": nbytes,
                          "gbps": round(gbps, 1), "frac": round(gbps / PEAK, 4)}
            print(f"{name:24s} {len(ks[0]):8d} {avg:9.2f} {v[0] / 1e3:9.2f} {nbytes / 1e6:10.2f} {gbps:9.1f} {gbps / PEAK:14.3f}")
        rec["sizes"][geo["size"]] = {"what": geo["what"], "B": geo["B"], "T": geo["T"], "kernels": rows}
    json.dump(rec, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
