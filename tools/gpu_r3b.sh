#!/bin/bash
# round 3, second GPU pass: checker builds on the GPU, config-3 phase ticks, 160-frame fused flavour at the 640-frame
# geometries, scaling table at N = 1
set -u
O=gpurun_out/r3b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/checked_build.sh gpu $O > $O/checked.out 2>&1
tail -4 $O/checked_bounds.log; tail -4 $O/checked_asan.log
timeout 300 python -m pytest -q -m gpu tests/test_gpu_r3.py -k "trained_regime_guided" 2>&1 | tail -3
timeout 600 python tools/stack_check.py --config 3 2>&1 | grep -v "rep [12]" | tail -12
for c in 6 7; do
  DR_STACK_FL=5 timeout 600 python tools/ab_option.py fused_stack 1 0 --config $c --rounds 2 2>&1 | tail -2
done
timeout 900 python tools/scale_table.py --gpus 1,2 --configs 2,3 --steps 2 --out $O/scale_table_n1.json 2>&1 | tail -6
