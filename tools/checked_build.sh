#!/bin/bash
# Checker pass over the kernels and the host library.
#   tools/checked_build.sh build            compile the two checker builds (works without a GPU):
#                                             diffroll_amd/lib/libdiffroll_amd_bounds.so  csrc with -DDR_BOUNDS: every LDS
#                                               address / in-range buffer offset / tensor extent checked at run time
#                                             diffroll_amd/lib/libdiffroll_amd_asan.so    host side under ASan + UBSan
#                                             diffroll_amd/lib/libdiffroll_amd_ubsan.so   host side under UBSan alone
#   tools/checked_build.sh cpu  [outdir]    the CPU suite against the ASan/UBSan host library (no GPU needed)
#   tools/checked_build.sh gpu  [outdir]    ON THE GPU BOX: fused-kernel cases, ragged-shape / random-geometry sweeps and the
#                                           round-3 battery against the DR_BOUNDS build (dr_debug_bounds must report 0), then
#                                           a GPU subset against the UBSan host library with glibc's heap checks on
#                                           (ASan and the HIP runtime do not start together)
# Logs: <outdir>/checked_{cpu,bounds,ubsan}.log (copy the summaries into profiles/).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
MODE=${1:-build}
O=${2:-gpurun_out/checked}
mkdir -p "$O"
LIBB=$R/diffroll_amd/lib/libdiffroll_amd_bounds.so
LIBA=$R/diffroll_amd/lib/libdiffroll_amd_asan.so
LIBU=$R/diffroll_amd/lib/libdiffroll_amd_ubsan.so
case "$MODE" in
build)
  python -m diffroll_amd.build --variant=bounds | tail -1
  python -m diffroll_amd.build --variant=asan | tail -1
  python -m diffroll_amd.build --variant=ubsan | tail -1
  ;;
cpu)
  RT=$(python -c "from diffroll_amd.build import asan_runtime; print(asan_runtime())")
  LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    DR_LIB=$LIBA timeout 1800 python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15 | tee "$O/checked_cpu.log"
  ;;
gpu)
  export HSA_ENABLE_IPC_MODE_LEGACY=0
  DR_LIB=$LIBB DR_BOUNDS_REPORT=1 timeout 2400 python -m pytest -q -m gpu -p no:cacheprovider \
    tests/test_gpu_fused.py tests/test_gpu_r3.py \
    "tests/test_gpu_parity.py::test_ragged_shapes_vs_oracle" "tests/test_gpu_parity.py::test_random_configurations_vs_oracle" \
    "tests/test_gpu_parity.py::test_random_full_width_geometries_vs_oracle" "tests/test_gpu_parity.py::test_flexible_width_tiles_vs_oracle" \
    "tests/test_gpu_parity.py::test_forward_golden" "tests/test_gpu_parity.py::test_steps_and_chain_golden" \
    "tests/test_gpu_parity.py::test_config1_chain_vs_oracle" "tests/test_gpu_parity.py::test_config5_shape_step_vs_oracle" \
    "tests/test_gpu_parity.py::test_bf16x3_forward_golden" "tests/test_gpu_parity.py::test_frontend_golden" \
    2>&1 | tail -25 | tee "$O/checked_bounds.log"
  # host side on the GPU: UBSan (first finding aborts) + glibc's heap consistency checks.  (ASan cannot be used here:
  # the HIP runtime segfaults in hipInit under ASan's allocator - the ASan build covers the CPU suite, mode `cpu`.)
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 MALLOC_CHECK_=3 MALLOC_PERTURB_=165 \
    DR_LIB=$LIBU timeout 1800 python -m pytest -q -m gpu -p no:cacheprovider \
    "tests/test_gpu_parity.py::test_forward_golden" "tests/test_gpu_parity.py::test_steps_and_chain_golden" \
    "tests/test_gpu_parity.py::test_frontend_golden" "tests/test_gpu_parity.py::test_load_from_checkpoint_end_to_end" \
    "tests/test_gpu_parity.py::test_ragged_shapes_vs_oracle" "tests/test_gpu_parity.py::test_random_chains_vs_oracle" \
    "tests/test_gpu_r3.py::test_two_engines_on_two_streams_both_match_the_oracle" \
    "tests/test_gpu_r3.py::test_config2_real_batch_200_step_chain_vs_oracle" tests/test_gpu_sharding.py \
    2>&1 | tail -15 | tee "$O/checked_ubsan.log"
  ;;
*) echo "usage: $0 build|cpu|gpu [outdir]"; exit 2;;
esac
