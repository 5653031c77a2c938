#!/bin/bash
# round 3: is the fused kernel still the right choice at 81-94 % fill now that the per-phase path splits beyond one round?
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 1 0; do echo "== DR_STACK=$v guided"; DR_STACK=$v timeout 900 python tools/small_batch_ab.py --batches 7,13,16,17,18,20,22,24,28,32 2>&1 | grep "B="; done
