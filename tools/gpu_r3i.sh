#!/bin/bash
# round 3: where does the 160-frame fused flavour lose its time at 640 frames (configs 6 and 5)?
set -u
O=gpurun_out/r3i; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for c in 6 5; do
  DR_STACK_FL=5 timeout 900 python tools/stack_check.py --config $c --level 2 > $O/stack_check_fl5_cfg$c.txt 2>&1
  tail -12 $O/stack_check_fl5_cfg$c.txt
done
