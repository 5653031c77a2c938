#!/usr/bin/env python3
"""Per-kernel digest of the gfx950 ISA of csrc/*.hip (device code only): the instruction stream of every kernel with
labels renamed in order of appearance and comments stripped, sha1'd - so a refactoring that is meant to leave the hot
kernels' machine code alone can be checked without a GPU.
    python tools/isa_hash.py [--save FILE] [--compare FILE] [-DDR_...]      (exit code 1 when a compared kernel differs)"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffroll_amd", "csrc")


def device_sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and f not in ("pack.hip", "plan.hip", "abi.hip", "debug_abi.hip", "comm.hip"))


def compile_asm(src, extra=()):
    out = os.path.join(tempfile.mkdtemp(), os.path.basename(src) + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
           src, "-o", out] + list(extra)
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def kernels(txt):
    labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", txt, re.M)]
    for i, (pos, name) in enumerate(labels):
        end = txt.find(".end_amdhsa_kernel", pos)
        nxt = labels[i + 1][0] if i + 1 < len(labels) else len(txt)
        if end < 0 or end > nxt:
            continue
        body = txt[pos:txt.find(".section", pos) if 0 <= txt.find(".section", pos) < end else end]
        yield name, body, txt[pos:end]


def digest(body):
    names = {}
    out = []
    for l in body.split("\n")[1:]:
        t = l.split(";")[0].rstrip()
        if not t.strip() or t.strip().startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", t.strip())
            if not m:
                continue
        t = re.sub(r"\.LBB\d+_\d+", lambda m: names.setdefault(m.group(0), f"L{len(names)}"), t.strip())
        out.append(t)
    return hashlib.sha1("\n".join(out).encode()).hexdigest()[:16], sum(1 for t in out if not t.endswith(":"))


def resources(meta):
    g = lambda k: (re.search(r"\." + k + r"\s+(\d+)", meta) or [None, "?"])[1]
    return dict(vgpr=g("amdhsa_next_free_vgpr"), sgpr=g("amdhsa_next_free_sgpr"), scratch=g("amdhsa_private_segment_fixed_size"))


def main():
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    table = {}
    for src in device_sources():
        txt = compile_asm(os.path.join(CSRC, src), extra)
        for name, body, full in kernels(txt):
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            h, n = digest(body)
            table[dem.split("(")[0]] = dict(sha=h, instructions=n, tu=src, **resources(full))
    for k in sorted(table):
        v = table[k]
        print(f"{v['sha']}  {v['instructions']:6d} ins  vgpr {v['vgpr']:>3} scratch {v['scratch']:>4}  {v['tu']:14s} {k}")
    rc = 0
    for i, a in enumerate(sys.argv):
        if a == "--save":
            json.dump(table, open(sys.argv[i + 1], "w"), indent=1, sort_keys=True)
        if a == "--compare":
            old = json.load(open(sys.argv[i + 1]))
            for k in sorted(set(old) | set(table)):
                o, n = old.get(k), table.get(k)
                if o is None or n is None:
                    print(("ADDED   " if o is None else "REMOVED ") + k)
                elif o["sha"] != n["sha"]:
                    print(f"CHANGED {k}: {o['instructions']} -> {n['instructions']} instructions, vgpr {o['vgpr']} -> {n['vgpr']}")
                    rc = 1
            print("compare:", "differences" if rc else "all common kernels identical")
    return rc


if __name__ == "__main__":
    sys.exit(main())
