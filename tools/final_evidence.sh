#!/bin/bash
# The evidence that is taken on the FINAL sources of a round, after tools/refresh_profiles.sh - run ON THE GPU BOX:
#     gpurun --timeout 4500 -- 'bash tools/final_evidence.sh r06 gpurun_out/final'
# (1) the whole GPU suite with every maxdiff() logged -> <rd>_parity_margins.txt; (2) the suite again with every engine's
# defaults changed (per-phase launches everywhere / the 160-frame flavour excluded); (3) the repeatability soaks and litmus
# builds; (4) the checker builds (tools/checked_build.sh build must have run where hipcc is); (5) phase ticks of the
# 160-frame stack against its pinned per-phase twins.
set -u
RD=${1:-r06}
O=${2:-gpurun_out/final}
mkdir -p "$O"
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f "$O/margins_raw.txt"
DR_PARITY_LOG=$PWD/$O/margins_raw.txt timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > "$O/gpu_suite.log" 2>&1; echo "suite rc=$?"
tail -3 "$O/gpu_suite.log"
python tools/lab/margins_summary.py "$O/margins_raw.txt" > "$O/${RD}_parity_margins.txt" 2>&1
{
echo "# the GPU suite with every engine's DEFAULTS changed (tools/tuning_env.py), one MI355X, round-6 final sources"
for tune in fused_stack=0 tune.stack_fl=-5; do
  echo "##### DR_TEST_TUNE=$tune"
  DR_TEST_TUNE=$tune timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR"
done
} > "$O/${RD}_forced_mode_suites.txt" 2>&1
cat "$O/${RD}_forced_mode_suites.txt"
bash tools/gpu_soak.sh $RD > "$O/soak.log" 2>&1; cp gpurun_out/soak/${RD}_soak.txt "$O/" 2>/dev/null
bash tools/checked_build.sh gpu "$O/checked" > "$O/checked.log" 2>&1
cp "$O/checked/checked_bounds.log" "$O/${RD}_checked_bounds_gpu.log" 2>/dev/null; cp "$O/checked/checked_ubsan.log" "$O/${RD}_checked_ubsan_gpu.log" 2>/dev/null
for c in 5 6; do
  { echo "# DR_TEST_TUNE=tune.ksplit_max=1,tune.tile=3205,tune.pw_nw=5 python tools/stack_check.py --config $c"
    DR_TEST_TUNE=tune.ksplit_max=1,tune.tile=3205,tune.pw_nw=5 timeout 600 python tools/stack_check.py --config $c 2>&1 | grep -v "rep [12]" | grep -v amdgpu.ids; } > "$O/${RD}_stack_phase_ticks_cfg$c.txt"
done
head -30 "$O/${RD}_parity_margins.txt"; tail -5 "$O/${RD}_checked_bounds_gpu.log" "$O/${RD}_checked_ubsan_gpu.log"; grep -c "bitwise_equal=True" "$O"/${RD}_stack_phase_ticks_cfg*.txt
