#!/bin/bash
# round 3: part-filled launches after the split-K cost model (more blocks than CUs allowed) and the 80 % fused threshold
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== new default, guided"; timeout 900 python tools/small_batch_ab.py --batches 1,2,3,4,5,6,7,8,10,12,13,14,16 2>&1 | grep "B="
echo "== old rules (DR_KSPLIT_BLOCKS=256), guided"; DR_KSPLIT_BLOCKS=256 timeout 900 python tools/small_batch_ab.py --batches 3,5,6,10,12,13 2>&1 | grep "B="
echo "== new default, generation"; timeout 900 python tools/small_batch_ab.py --batches 6,10,12,14,20,24 --sampler generation_ddpm_x0 2>&1 | grep "B="
echo "== new default, guided T=640"; timeout 900 python tools/small_batch_ab.py --batches 1,2,3 --T 640 2>&1 | grep "B="
echo "== old rules, guided T=640"; DR_KSPLIT_BLOCKS=256 timeout 900 python tools/small_batch_ab.py --batches 1,2,3 --T 640 2>&1 | grep "B="
