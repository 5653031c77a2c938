#!/usr/bin/env python3
"""ISA audit of the LDS-DMA hand-over: an `s_barrier` that hands a DMA-staged LDS tile to other waves must be preceded,
in the ISSUING wave, by `s_waitcnt vmcnt(0)` after its last `buffer_load ... lds` - barriers do not drain VMEM
(MI355X_MICROARCH.md "Two waves per SIMD", item 7) and hipcc does not always insert the wait for a `__syncthreads()`
behind LDS-DMA builtins (it treats them as loads without a register result).

For every kernel of csrc/{gemm,stack,tail}.hip: walk the control-flow graph of the gfx950 assembly from every LDS-DMA load and
report any path that reaches an s_barrier (or the end of the kernel) without passing `s_waitcnt vmcnt(0)`.

    python tools/isa_audit.py [--defines -DDR_BOUNDS]        (exit code 1 when a path is found)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAX_DIST = 100


CSRC = os.path.join(ROOT, "diffroll_amd", "csrc")
DEVICE_UNITS = ("gemm.hip", "stack.hip", "tail.hip")        # the translation units that hold LDS-DMA producers


def compile_asm(extra, units=DEVICE_UNITS):
    """gfx950 assembly of the kernel translation units (device code only), concatenated."""
    txt = []
    for u in units:
        out = os.path.join(tempfile.mkdtemp(), u + ".s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
               os.path.join(CSRC, u), "-o", out] + extra
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        txt.append(open(out).read())
    return "\n".join(txt)


def kernels(txt):
    labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_ZN2dr\w+):", txt, re.M)]
    for i, (pos, name) in enumerate(labels):
        end = txt.find(".end_amdhsa_kernel", pos)
        nxt = labels[i + 1][0] if i + 1 < len(labels) else len(txt)
        if end < 0 or end > nxt:
            continue            # a device function, not a kernel
        yield name, txt[pos:end].split("\n")


def audit(name, lines):
    # basic blocks: split at labels and after branches
    ins = []          # (text) of real instructions, with block labels kept as ('LABEL', name)
    for l in lines[1:]:
        t = l.split(";")[0].rstrip()
        if not t.strip():
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            ins.append(("L", m.group(1)))
        elif t.startswith("\t") and not t.strip().startswith("."):
            ins.append(("I", t.strip()))
    label_at = {v: i for i, (k, v) in enumerate(ins) if k == "L"}

    def succ(i):
        k, t = ins[i]
        if k == "L":
            return [i + 1] if i + 1 < len(ins) else []
        op = t.split()[0]
        if op == "s_endpgm":
            return []
        if op == "s_branch":
            return [label_at[t.split()[1]]]
        if op.startswith("s_cbranch"):
            tgt = t.split()[-1]
            return [label_at[tgt]] + ([i + 1] if i + 1 < len(ins) else [])
        if op in ("s_setpc_b64", "s_swappc_b64"):
            return [i + 1] if op == "s_swappc_b64" and i + 1 < len(ins) else []
        return [i + 1] if i + 1 < len(ins) else []

    # Breadth-first from every LDS-DMA load, at most MAX_DIST instructions along a path: the hand-over barrier of a
    # producer loop sits ~25 instructions behind its DMA issue loop.  (The search is path-INsensitive: without the
    # bound it also finds an infeasible ~210-instruction path through the persistent kernels' phase loop - the "no
    # group barrier" branch followed by "the loop continues", then the CONSUMER role - that ends at the consumers' barrier.)
    findings = []
    dma = [i for i, (k, t) in enumerate(ins) if k == "I" and re.search(r"buffer_load_dword\w*\s.*\blds\b", t)]
    for d in dma:
        dist = {s_: 1 for s_ in succ(d)}
        queue = list(dist)
        while queue:
            i = queue.pop(0)
            k, t = ins[i]
            if k == "I":
                if re.match(r"s_waitcnt\b", t) and re.search(r"vmcnt\(0\)", t):
                    continue                      # this path is covered
                if t.startswith("s_barrier"):
                    findings.append((d, i))
                    continue
            if dist[i] >= MAX_DIST:
                continue
            for n in succ(i):
                if n not in dist:
                    dist[n] = dist[i] + (1 if k == "I" else 0)
                    queue.append(n)
    return len(dma), findings


def main():
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    txt = compile_asm(extra)
    bad = 0
    for name, lines in kernels(txt):
        n, f = audit(name, lines)
        if n == 0:
            continue
        uniq = sorted({b for _, b in f})
        print(f"{name[:72]:72s} LDS-DMA sites {n:2d}  barriers reachable without vmcnt(0): {len(uniq)}")
        bad += len(uniq)
    print("RESULT", "FAIL" if bad else "ok", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
