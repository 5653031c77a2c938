#!/usr/bin/env python3
"""What /sys/class/kfd/kfd shows from inside a lease (run on the GPU box): the input of csrc/tenants.h.  Output of round 5:
profiles/r05_kfd_sysfs_probe.txt."""
import os, time, glob, sys
import torch
torch.zeros(1, device="cuda").add_(1); torch.cuda.synchronize()
print("pid", os.getpid())
base = "/sys/class/kfd/kfd/proc"
t0 = time.perf_counter()
try:
    ents = os.listdir(base)
except Exception as e:
    print("listdir failed", repr(e)); ents = []
print("proc entries", ents, "listdir us", 1e6 * (time.perf_counter() - t0))
for p in ents:
    d = os.path.join(base, p)
    try:
        print(p, sorted(os.listdir(d)))
        qd = os.path.join(d, "queues")
        if os.path.isdir(qd):
            for q in os.listdir(qd):
                vals = {}
                for f in os.listdir(os.path.join(qd, q)):
                    try: vals[f] = open(os.path.join(qd, q, f)).read().strip()
                    except Exception as e: vals[f] = repr(e)
                print("   queue", q, vals)
        for st in glob.glob(os.path.join(d, "stats_*")):
            for f in os.listdir(st):
                t1 = time.perf_counter()
                try: v = open(os.path.join(st, f)).read().strip()
                except Exception as e: v = repr(e)
                print("   ", os.path.basename(st), f, v, "read us", 1e6 * (time.perf_counter() - t1))
    except Exception as e:
        print(p, "err", repr(e))
for n in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*")):
    try:
        gid = open(os.path.join(n, "gpu_id")).read().strip()
        props = dict(l.split() for l in open(os.path.join(n, "properties")).read().strip().splitlines())
        print(n, "gpu_id", gid, {k: props.get(k) for k in ("location_id", "domain", "simd_count", "drm_render_minor")})
    except Exception as e:
        print(n, "err", repr(e))
pr = torch.cuda.get_device_properties(0)
print("hip props", getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None), getattr(pr, "pci_domain_id", None))
# timing of a full scan
t0 = time.perf_counter()
for _ in range(100):
    n = 0
    for p in os.listdir(base):
        qd = os.path.join(base, p, "queues")
        try:
            for q in os.listdir(qd):
                n += 1
                open(os.path.join(qd, q, "gpuid")).read()
        except Exception: pass
print("full scan us", 1e4 * (time.perf_counter() - t0), "queues", n)
# a second process
if len(sys.argv) < 2:
    import subprocess
    r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True)
    print("---- child ----"); print(r.stdout[-3000:]); print(r.stderr[-500:])
