#!/bin/bash
# round 3: is the lower effective clock of config 3's launch a matter of its geometry (64-frame blocks) or of its data
# (unconditional generation)?  Swap them: guided sampling at 8 clips (16 evaluations -> 64-frame blocks) and
# generation at 32 clips (32 evaluations -> 128-frame blocks).
set -u
O=gpurun_out/r3o; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== guided, 8 clips (64-frame blocks)"; timeout 600 python tools/stack_check.py --config 2 --batch 8 2>&1 | tee $O/guided_b8.txt | grep -E "MHz|phase ticks" | cut -c1-200
echo "== generation, 32 clips (128-frame blocks)"; timeout 600 python tools/stack_check.py --config 3 --batch 32 2>&1 | tee $O/generation_b32.txt | grep -E "MHz|phase ticks" | cut -c1-200
