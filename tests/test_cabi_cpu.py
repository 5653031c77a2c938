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


def header_functions():
    text = open(os.path.join(ROOT, "include", "diffroll_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dr_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree(lib):
    from diffroll_amd import _cabi
    declared = header_functions()
    assert declared, "no functions parsed from the header"
    assert sorted(_cabi.EXPORTS) == declared
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


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
    rc = lib.dr_rccl_version(C.byref(v))
    assert (rc == 0 and v.value > 20000) or (rc != 0 and lib.dr_comm_last_error())
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
    n = C.c_int64(7)
    assert lib.dr_stack_fallbacks(None, C.byref(n)) == _cabi.DR_EINVAL and lib.dr_tail_launches(None, C.byref(n)) == _cabi.DR_EINVAL
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
    assert {"bounds", "asan", "ubsan"} <= set(build.VARIANTS)          # (+ measurement builds)
    assert "-DDR_BOUNDS" in build.VARIANTS["bounds"]["flags"]
    assert any("-fsanitize=address" in f for f in build.VARIANTS["asan"]["flags"])
    assert build.variant_path("bounds").endswith("libdiffroll_amd_bounds.so")
    csrc = os.path.join(ROOT, "diffroll_amd", "csrc")
    src = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h")))
    assert src.count("DR_CHECK_LDS(") >= 12 and "check_gemm_extents" in src      # the instrumentation is there
    for unit in ("gemm", "stack", "tail"):                                       # ... in every unit that carries checks
        assert f"DR_BOUNDS_TU({unit})" in src
