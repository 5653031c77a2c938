#!/bin/bash
# round 3: the driver's round-end sequence on the final sources + the checker builds
set -u
O=gpurun_out/r3f; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f $O/margins.txt
DR_PARITY_LOG=$PWD/$O/margins.txt timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > $O/pytest.log
echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/checked_build.sh gpu $O > $O/checked.out 2>&1; tail -4 $O/checked_bounds.log
RT=$(python -c "from diffroll_amd.build import asan_runtime; print(asan_runtime())")
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0:allocator_may_return_null=1 DR_LIB=$PWD/diffroll_amd/lib/libdiffroll_amd_asan.so timeout 900 python -m pytest -q -m gpu -p no:cacheprovider "tests/test_gpu_parity.py::test_forward_golden" "tests/test_gpu_parity.py::test_frontend_golden" "tests/test_gpu_parity.py::test_steps_and_chain_golden" "tests/test_gpu_parity.py::test_load_from_checkpoint_end_to_end" > $O/asan_gpu.txt 2>&1; echo "asan pytest rc=$?"; tail -6 $O/asan_gpu.txt
