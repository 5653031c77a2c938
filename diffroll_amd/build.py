"""Build the in-tree HIP shared library (gfx950 only): ``python -m diffroll_amd.build``.

hipcc cross-compiles without a GPU; the resulting ``diffroll_amd/lib/libdiffroll_amd.so`` is
git-ignored but travels with the tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdiffroll_amd.so")
# one translation unit per kernel family (a kernel edit rebuilds its unit only; the units compile in parallel)
SOURCES = ["gemm.hip", "stack.hip", "tail.hip", "update.hip", "frontend.hip", "pack.hip", "plan.hip", "abi.hip", "debug_abi.hip", "comm.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("kernels.h", "device_common.h", "gemm_body.h", "persistent.h", "update_quad.h", "engine_state.h", "tenants.h")] + \
          [os.path.join(os.path.dirname(HERE), "include", h) for h in ("diffroll_amd.h", "diffroll_amd_debug.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


VARIANTS = {
    # checker builds (tools/checked_build.sh), written next to the production library and selected with DR_LIB=<path>
    #   bounds: -DDR_BOUNDS - every hand-computed LDS address / in-range buffer offset of the kernels and every tensor
    #           extent of a launch is checked at run time (dr_debug_bounds reports)
    #   asan:   the HOST side (pack / plan / abi / debug_abi / comm .hip: packing, tables, C-ABI marshalling) under AddressSanitizer +
    #           UndefinedBehaviorSanitizer; device code is compiled as usual (-fno-gpu-sanitize)
    "bounds": dict(flags=["-DDR_BOUNDS"], link=[]),
    # test build of the time-out path: the ONLY library that knows the option "stack_fault_test" (the persistent kernels' group
    # barriers can be told to wait for one arrival too many); correct results otherwise.  tests/hook_cases.py runs against it
    "hook": dict(flags=["-DDR_FAULT_HOOK"], link=[]),
    # litmus builds (WRONG on purpose): the hand-over without its vmcnt wait / hand-offs without write-through stores
    "fault1": dict(flags=["-DDR_FAULT=1"], link=[]),
    "fault2": dict(flags=["-DDR_FAULT=2"], link=[]),
    "asan": dict(flags=["-O1", "-g", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-shared-libsan", "-fno-omit-frame-pointer"],
                 link=["-fsanitize=address,undefined", "-shared-libsan"]),
    #   ubsan:  the host side under UndefinedBehaviorSanitizer alone (-fno-sanitize-recover: the first finding aborts).
    #           This is the variant that can run ON THE GPU BOX: the HIP runtime does not initialise under ASan's
    #           allocator (segfault inside hipInit), UBSan does not touch the allocator
    "ubsan": dict(flags=["-O1", "-g", "-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer"],
                  link=["-fsanitize=undefined", "-shared-libsan", "@RPATH:libclang_rt.ubsan_standalone-x86_64.so"]),
}


def variant_path(variant: str) -> str:
    return os.path.join(LIBDIR, f"libdiffroll_amd_{variant}.so")


def build(force: bool = False, verbose: bool = True, variant: str = "") -> str:
    """variant "" = the production library; "bounds" / "asan" = the checker builds (see VARIANTS)."""
    hipcc = _hipcc()
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj" + ("_" + variant if variant else ""))
    os.makedirs(objdir, exist_ok=True)
    extra = VARIANTS[variant] if variant else dict(flags=[], link=[])
    lib = variant_path(variant) if variant else LIB
    objs = []
    relink = force
    jobs = []
    # objects of translation units that no longer exist (or clang-offload-bundler leftovers of an interrupted build) must not
    # linger next to the ones the library is linked from
    wanted = {src.replace(".hip", ".o") for src in SOURCES}
    for name in os.listdir(objdir):
        if name not in wanted:
            os.remove(os.path.join(objdir, name))
            relink = True
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            jobs.append([hipcc] + FLAGS + extra["flags"] + ["-c", s, "-o", o])
        objs.append(o)
    if jobs:
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
        relink = True
    if relink or _stale(lib, objs):
        link = []
        for f in extra["link"]:
            if f.startswith("@RPATH:"):      # the directory of a compiler runtime library, as an rpath of the output
                rt = subprocess.check_output([hipcc, "-print-file-name=" + f[len("@RPATH:"):]], text=True).strip()
                link.append("-Wl,-rpath," + os.path.dirname(os.path.realpath(rt)))
            else:
                link.append(f)
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + link + objs + ["-ldl", "-o", lib]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


def asan_runtime() -> str:
    """The shared ASan runtime a python process must LD_PRELOAD before it dlopens the "asan" variant."""
    out = subprocess.check_output([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    return out


if __name__ == "__main__":
    var = ""
    for a in sys.argv[1:]:
        if a.startswith("--variant="):
            var = a.split("=", 1)[1]
    print(build(force="--force" in sys.argv, variant=var))
