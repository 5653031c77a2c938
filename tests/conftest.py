import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:          # test modules share helpers (make_model of test_gpu_parity)
    sys.path.insert(0, HERE)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle is torch-CPU: torch's default thread count inside the GPU box's container (128) is ~8x slower than
    # 16 for these shapes (bench.py cpu_baseline measures it) - cap it so the parity suite spends its time on the GPU
    import torch
    torch.set_num_threads(min(16, torch.get_num_threads()))
    from tools import tuning_env          # DR_TEST_TUNE="fused_stack=0,...": a forced-mode run of the suite
    tuning_env.install()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """Checker runs (tools/checked_build.sh: DR_LIB points at the -DDR_BOUNDS library): the session fails if any LDS /
    buffer-offset / tensor-extent check fired in this process (child processes report their own: tests/fused_cases.py)."""
    import os
    if not os.environ.get("DR_BOUNDS_REPORT"):
        return
    from diffroll_amd import _cabi
    if _cabi._lib is None:
        return
    v = _cabi.bounds_violations()
    print(f"\n[DR_BOUNDS] {'production library (no checks)' if v is None else 'violations (code, detail, detail, count) = ' + str(v)}")
    if v is not None and v[3] != 0:
        session.exitstatus = 1
