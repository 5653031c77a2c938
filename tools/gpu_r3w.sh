#!/bin/bash
# round 3: after the fused-stack threshold fix (strictly more than half a chip): pwk on / off at 8 evaluations, whole suite
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r3w
for v in 1 0; do echo "== DR_PWK=$v"; DR_PWK=$v timeout 600 python tools/small_batch_ab.py --batches 3,4,5 2>&1 | grep "B="; done
rm -f gpurun_out/r3w/margins.txt
DR_PARITY_LOG=$PWD/gpurun_out/r3w/margins.txt timeout 2700 python -m pytest tests -q -m gpu --maxfail=8 2>&1 | tail -6
