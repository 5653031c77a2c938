"""CPU-only checks of the C-ABI boundary: the library builds, loads, and exports exactly the symbols
include/diffroll_amd.h declares; without a GPU every entry point fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def lib():
    from diffroll_amd.build import build
    from diffroll_amd import _cabi
    build(verbose=False)
    return _cabi.load_library()


def header_functions(name="diffroll_amd.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dr_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    """Both headers: include/diffroll_amd.h is the boundary (<= 30 functions, SURVEY.md 8b asks for about nine), and
    include/diffroll_amd_debug.h the lab (measurement / checker / test entry points of the same library).  The ctypes
    binding declares exactly these, the library exports them, and nothing ELSE that starts with dr_."""
    import subprocess
    from diffroll_amd import _cabi
    declared = header_functions()
    assert declared, "no functions parsed from the header"
    assert sorted(_cabi.EXPORTS) == declared
    assert len(declared) <= 30, len(declared)
    lab = header_functions("diffroll_amd_debug.h")
    assert sorted(_cabi.DEBUG_EXPORTS) == lab and not set(lab) & set(declared)
    for name in declared + lab:
        assert hasattr(lib, name), f"{name} declared in a header but not exported"
    # the boundary header holds no lab vocabulary
    text = open(os.path.join(ROOT, "include", "diffroll_amd.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for word in ("dr_debug_", "dr_bench_", "dr_profile_", "dr_stack_status", "dr_cold_times"):
        assert word not in code, word                     # (comments may point at the lab header)
    assert "stack_fault_test" not in text
    nm = subprocess.run(["nm", "-D", "--defined-only", _cabi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in nm.splitlines() if ln.split()[-1].startswith("dr_")})
    assert exported == sorted(declared + lab), set(exported) ^ set(declared + lab)


def test_production_library_has_no_fault_injection():
    """the test hook of the time-out path ("stack_fault_test": barriers that wait for one arrival too
    many) is compiled into the "hook" variant only (-DDR_FAULT_HOOK) - the shipped library does not contain the string."""
    import subprocess
    from diffroll_amd import build
    strs = subprocess.run(["strings", build.build(verbose=False)], capture_output=True, text=True, check=True).stdout
    assert "stack_fault_test" not in strs
    assert "-DDR_FAULT_HOOK" in build.VARIANTS["hook"]["flags"]
    hook = build.variant_path("hook")
    if os.path.exists(hook):      # (built by __graft_entry__.build(); a CPU box without it skips the positive half)
        assert "stack_fault_test" in subprocess.run(["strings", hook], capture_output=True, text=True, check=True).stdout


def test_launch_info_struct_layout_matches_header():
    from diffroll_amd import _cabi
    text = open(os.path.join(ROOT, "include", "diffroll_amd.h")).read()
    body = re.search(r"typedef struct dr_launch_info \{(.*?)\} dr_launch_info;", text, flags=re.S).group(1)
    fields = re.findall(r"\b(int32_t|int64_t)\s+(\w+);", body)
    assert [f[1] for f in fields] == [f[0] for f in _cabi.DrLaunchInfo._fields_]
    for (ctype, _), (_, pyt) in zip(fields, _cabi.DrLaunchInfo._fields_):
        assert pyt is (C.c_int32 if ctype == "int32_t" else C.c_int64)
    modes = dict(re.findall(r"(DR_MODE_\w+) = (\d)", text))
    assert {int(v) for v in modes.values()} == set(_cabi.MODES) and _cabi.MODES[int(modes["DR_MODE_FUSED_STACK_TAIL"])] == "fused_stack+tail"


def test_abi_version(lib):
    from diffroll_amd import _cabi
    assert lib.dr_abi_version() == _cabi.DR_ABI_VERSION
    text = open(os.path.join(ROOT, "include", "diffroll_amd.h")).read()
    assert int(re.search(r"#define DR_ABI_VERSION (\d+)", text).group(1)) == _cabi.DR_ABI_VERSION


def test_config_struct_layout_matches_header():
    from diffroll_amd import _cabi
    text = open(os.path.join(ROOT, "include", "diffroll_amd.h")).read()
    body = re.search(r"typedef struct dr_config \{(.*?)\} dr_config;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(int32_t|float)\s+(\w+);", body)
    assert [f[1] for f in fields] == [f[0] for f in _cabi.DrConfig._fields_]
    for (ctype, _), (_, pyt) in zip(fields, _cabi.DrConfig._fields_):
        assert pyt is (C.c_int32 if ctype == "int32_t" else C.c_float)
    assert C.sizeof(_cabi.DrConfig) == 4 * len(fields)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly(lib):
    from diffroll_amd import _cabi
    cfg = _cabi.DrConfig(abi_version=_cabi.DR_ABI_VERSION, device=0, residual_channels=64, residual_layers=2,
                         kernel_size=3, dilation_base=2, dilation_bound=4, n_mels=229, timesteps=8,
                         sample_rate=16000, n_fft=2048, hop_length=512, f_min=0.0, f_max=8000.0,
                         beta_start=1e-4, beta_end=0.02)
    h = C.c_void_p()
    rc = lib.dr_create(C.byref(h), C.byref(cfg))
    assert rc == _cabi.DR_EHIP and not h.value
    assert b"no CPU fallback" in lib.dr_last_error(None)
    # wrong ABI version / bad config are rejected before touching the device
    cfg.abi_version = 999
    assert lib.dr_create(C.byref(h), C.byref(cfg)) == _cabi.DR_EINVAL
    # python wrapper raises
    from diffroll_amd import ClassifierFreeDiffRoll, EngineError
    m = ClassifierFreeDiffRoll(64, False, "fixed", 229, [0, 1, "imagewise"], residual_layers=2, kernel_size=3,
                               dilation_base=2, sampling={"type": "cfdg_ddpm_x0", "w": 0.5},
                               spec_args=dict(sample_rate=16000, n_fft=2048, hop_length=512, n_mels=229, f_min=0,
                                              f_max=8000, center=True, normalized=True, pad_mode="reflect"))
    with pytest.raises(EngineError):
        m.engine
    with pytest.raises(EngineError):
        m(torch.zeros(1, 1, 8, 88), torch.zeros(1, 4096), torch.zeros(1, dtype=torch.long))


def test_header_is_plain_c_and_links(lib, tmp_path):
    """include/diffroll_amd.h is the whole boundary: it must compile as C99 (no C++, no torch/HIP types) and a
    plain C program must link against the library and call it (version query and a failing create: no GPU here)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    from diffroll_amd import _cabi
    src = tmp_path / "cabi_probe.c"
    src.write_text(r"""
        #include <stdio.h>
        #include <string.h>
        #include "diffroll_amd.h"
        int main(void) {
            dr_config cfg;
            memset(&cfg, 0, sizeof cfg);
            cfg.abi_version = DR_ABI_VERSION + 1000;          /* wrong on purpose */
            dr_engine* e = NULL;
            int rc = dr_create(&e, &cfg);
            printf("abi %d create_rc %d err '%s'\n", dr_abi_version(), rc, dr_last_error(NULL));
            return (dr_abi_version() == DR_ABI_VERSION && rc != 0 && e == NULL) ? 0 : 1;
        }
    """)
    exe = tmp_path / "cabi_probe"
    libdir = os.path.dirname(_cabi.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", libdir, "-ldiffroll_amd", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "ABI version mismatch" in r.stdout


def test_comm_entry_points_fail_cleanly_without_a_gpu():
    """The RCCL half of the ABI: librccl is found by dlopen (version readable), and creating a communicator on a box
    with no HIP device is an error code + message, not a crash."""
    import ctypes as C
    from diffroll_amd import _cabi
    import torch
    lib = _cabi.load_library()
    v = C.c_int(0)
    rc = lib.dr_comm_info(None, None, None, C.byref(v))        # no communicator: the version of the loaded librccl alone
    assert (rc == 0 and v.value > 20000) or (rc != 0 and lib.dr_comm_last_error())
    n = C.c_int(0)
    assert lib.dr_comm_info(None, C.byref(n), None, None) == _cabi.DR_EINVAL
    if not torch.cuda.is_available():
        h = C.c_void_p()
        assert lib.dr_comm_create(C.byref(h), b"\0" * 128, 1, 0, 0) == _cabi.DR_EINVAL
        assert b"out of range" in lib.dr_comm_last_error() and not h.value
    assert lib.dr_comm_create(None, None, 1, 0, 0) == _cabi.DR_EINVAL
    assert lib.dr_gather(None, None, None, None, 1, 1, None) == _cabi.DR_EINVAL


def test_round3_entry_points_reject_bad_calls_without_a_gpu(lib):
    """dr_finish / dr_sample_checked / the counters and the checker hooks: NULL handles are an error code, never a
    crash; the production library says that it is not a checker build."""
    from diffroll_amd import _cabi
    assert lib.dr_finish(None, None) == _cabi.DR_EINVAL
    assert lib.dr_sample_checked(None, 1, None, None, 1, 1, 0.5, 0, 0, 1, None, None) == _cabi.DR_EINVAL
    info = _cabi.DrLaunchInfo()
    assert lib.dr_launch_state(None, C.byref(info)) == _cabi.DR_EINVAL and lib.dr_launch_state(None, None) == _cabi.DR_EINVAL
    assert lib.dr_set_option(None, b"fused_stack", 1) == _cabi.DR_EINVAL and lib.dr_debug_set_option(None, b"tune.tile", 0) == _cabi.DR_EINVAL
    assert lib.dr_debug_stft_power(None, None, 1, 4096, None, None) == _cabi.DR_EINVAL
    out = (C.c_int64 * 4)()
    if "bounds" not in os.path.basename(_cabi.LIB_PATH):
        assert lib.dr_debug_bounds(out, 0) == _cabi.DR_ESTATE and b"checker build" in lib.dr_last_error(None)
        assert _cabi.bounds_violations() is None
    assert lib.dr_debug_bounds(None, 0) == _cabi.DR_EINVAL
    assert _cabi.DR_ETIMEOUT == -6
    text = open(os.path.join(ROOT, "include", "diffroll_amd.h")).read()
    assert re.search(r"DR_ETIMEOUT\s*=\s*-6", text)


def test_checker_build_variants_are_declared():
    """tools/checked_build.sh drives diffroll_amd.build's variants: the -DDR_BOUNDS kernels and the ASan/UBSan host."""
    from diffroll_amd import build
    assert {"bounds", "asan", "ubsan", "hook"} <= set(build.VARIANTS)          # (+ measurement builds)
    assert "-DDR_BOUNDS" in build.VARIANTS["bounds"]["flags"]
    assert any("-fsanitize=address" in f for f in build.VARIANTS["asan"]["flags"])
    assert build.variant_path("bounds").endswith("libdiffroll_amd_bounds.so")
    csrc = os.path.join(ROOT, "diffroll_amd", "csrc")
    src = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h")))
    assert src.count("DR_CHECK_LDS(") >= 12 and "check_gemm_extents" in src      # the instrumentation is there
    for unit in ("gemm", "stack", "tail"):                                       # ... in every unit that carries checks
        assert f"DR_BOUNDS_TU({unit})" in src


def test_library_reads_no_environment_variable():
    """the A/B knobs are dr_set_option names ("tune.*", csrc/kernels.h Tuning) - the shipped library
    neither imports getenv nor carries a DR_* string, and no source under csrc/ calls getenv."""
    import subprocess
    from diffroll_amd import build
    lib = build.build(verbose=False)
    und = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und
    strs = subprocess.run(["strings", lib], capture_output=True, text=True, check=True).stdout.splitlines()
    assert [s for s in strs if s.startswith("DR_")] == []
    csrc = os.path.join(os.path.dirname(os.path.abspath(build.__file__)), "csrc")
    for name in os.listdir(csrc):
        assert "getenv" not in open(os.path.join(csrc, name)).read(), name


def test_tuning_env_hook_parses_and_merges():
    from tools import tuning_env
    assert tuning_env.parse("fused_stack=0, tune.tile=3202") == {"fused_stack": 0, "tune.tile": 3202}
    env = tuning_env.env_with({"DR_TEST_TUNE": "fused_tail=0"}, tune__stack_fl=2, blocked_accumulation=1)
    assert tuning_env.parse(env["DR_TEST_TUNE"]) == {"fused_tail": 0, "tune.stack_fl": 2, "blocked_accumulation": 1}


def _fake_kfd(root, procs, nodes):
    """A /sys/class/kfd/kfd look-alike: procs = {pid: {"queues": [gpuid, ...], "occ": {gpuid: cus}}}, nodes = [(gpu_id,
    location_id, domain)]."""
    for i, (gid, loc, dom) in enumerate(nodes):
        nd = os.path.join(root, "topology", "nodes", str(i))
        os.makedirs(nd)
        open(os.path.join(nd, "gpu_id"), "w").write(f"{gid}\n")
        open(os.path.join(nd, "properties"), "w").write(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {loc}\ndomain {dom}\ndrm_render_minor 128\n")
    for pid, p in procs.items():
        pd = os.path.join(root, "proc", str(pid))
        os.makedirs(os.path.join(pd, "queues"))
        open(os.path.join(pd, "pasid"), "w").write("32770\n")
        for n, gid in enumerate(p.get("queues", [])):
            qd = os.path.join(pd, "queues", str(n))
            os.makedirs(qd)
            open(os.path.join(qd, "gpuid"), "w").write(f"{gid}\n")
            open(os.path.join(qd, "type"), "w").write("0\n")
        for gid, cus in p.get("occ", {}).items():
            sd = os.path.join(pd, f"stats_{gid}")
            os.makedirs(sd)
            open(os.path.join(sd, "cu_occupancy"), "w").write(f"{cus}\n")


def test_co_tenant_scan_on_a_fake_kfd_tree(tmp_path):
    """csrc/tenants.h through dr_debug_tenants: the GPU is found by PCI address, processes are counted by the queues
    they hold on THAT GPU (host daemons without queues and tenants of other GPUs do not count), and the busy CUs are
    theirs.  The layout is what an MI355X node's /sys/class/kfd/kfd shows (profiles/r05_kfd_sysfs_probe.txt)."""
    from diffroll_amd import _cabi
    lib = _cabi.load_library()
    root = str(tmp_path / "kfd")
    _fake_kfd(root, procs={
        1347236: {"queues": [28206, 28206], "occ": {28206: 0}},            # "us": queues on our GPU, idle
        227209: {"queues": [], "occ": {28206: 0, 25266: 0}},               # a host daemon: contexts everywhere, no queue
        1321268: {"queues": [25266, 25266, 25266], "occ": {25266: 77}},    # a busy tenant of ANOTHER GPU
    }, nodes=[(0, 0, 0), (25266, 0x1500, 0), (28206, 0x5A00, 0)])
    out = (C.c_int64 * 4)()
    assert lib.dr_debug_tenants(root.encode(), 0, 0x5A, 0, out) == 0
    assert list(out) == [28206, 1, 0, 1]                                    # our GPU: one holder, nothing busy
    _fake_kfd(root, procs={1350000: {"queues": [28206], "occ": {28206: 133}}}, nodes=[])
    assert lib.dr_debug_tenants(root.encode(), 0, 0x5A, 0, out) == 0
    assert list(out) == [28206, 2, 133, 1]                                  # a second process computing on it: the engine yields
    assert lib.dr_debug_tenants(root.encode(), 0, 0x15, 0, out) == 0 and list(out) == [25266, 1, 77, 1]
    assert lib.dr_debug_tenants(root.encode(), 0, 0x77, 0, out) == 0 and out[0] == -1        # a GPU the tree does not list
    assert lib.dr_debug_tenants(str(tmp_path / "absent").encode(), 0, 0x5A, 0, out) == 0 and out[0] == -1
