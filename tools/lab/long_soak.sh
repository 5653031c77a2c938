#!/bin/bash
# Round 6: a LONG repeatability soak of the persistent kernels' cross-workgroup hand-offs (same seed -> bitwise equal rolls, no
# barrier time-out) at every fused flavour, then the forced-mode suite with per-phase launches everywhere - ON THE GPU BOX:
#     gpurun --timeout 3000 -- 'bash tools/lab/long_soak.sh gpurun_out/long'
set -u
O=${1:-gpurun_out/long}; mkdir -p "$O"
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
echo "# long repeatability soak, round-6 final sources, one MI355X: tools/fused_soak.py (every chain of a configuration must reproduce the first one bit for bit)"
for spec in "2 500" "3 500" "6 250" "5 120" "7 100"; do
  set -- $spec
  echo "## tools/fused_soak.py --chains $2 --config $1"
  timeout 1500 python tools/fused_soak.py --chains $2 --config $1 2>&1 | grep -v amdgpu.ids | tail -2
done
} > "$O/r06_long_soak.txt" 2>&1
cat "$O/r06_long_soak.txt"
{
echo "# the GPU suite with every engine's DEFAULTS changed (tools/tuning_env.py), one MI355X, round-6 final sources"
echo "##### DR_TEST_TUNE=fused_stack=0"
DR_TEST_TUNE=fused_stack=0 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR"
} > "$O/forced_fused_stack0.txt" 2>&1
cat "$O/forced_fused_stack0.txt"
