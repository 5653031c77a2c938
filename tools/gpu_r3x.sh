#!/bin/bash
# round 3: split-K beyond one resident round (the ticket reduction needs no co-residency) for part-filled launches
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== default"; timeout 600 python tools/small_batch_ab.py --batches 3,5,6,7,8 2>&1 | grep "B="
for mb in 512 1024; do for km in 4 8 16; do
echo "== DR_STACK=0 DR_KSPLIT_BLOCKS=$mb DR_KSPLIT_MAX=$km"; DR_STACK=0 DR_KSPLIT_BLOCKS=$mb DR_KSPLIT_MAX=$km timeout 600 python tools/small_batch_ab.py --batches 3,5,6,7,8 2>&1 | grep "B="
done; done
