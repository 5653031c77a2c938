#!/bin/bash
# round 3: is the fused stack the right choice at exactly half a chip of blocks (8 evaluations x 125 frames)?
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 1 0; do echo "== DR_STACK=$v guided"; DR_STACK=$v timeout 600 python tools/small_batch_ab.py --batches 4,5,6,8 2>&1 | grep "B="; done
for v in 1 0; do echo "== DR_STACK=$v generation"; DR_STACK=$v timeout 600 python tools/small_batch_ab.py --batches 8,10,12 --sampler generation_ddpm_x0 2>&1 | grep "B="; done
