#!/bin/bash
# round 3: the shader clock the fused launches of config 2 / 3 really run at (wall_clock64 next to the phase marks)
set -u
O=gpurun_out/r3k; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for c in 2 3; do
  timeout 900 python tools/stack_check.py --config $c > $O/stack_check_cfg$c.txt 2>&1
  grep -E "MHz|chain:" $O/stack_check_cfg$c.txt
done
