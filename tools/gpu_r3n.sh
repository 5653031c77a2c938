#!/bin/bash
# round 3: whole GPU suite on the current build (observed margins logged)
set -u
O=gpurun_out/r3n; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f $O/margins.txt
DR_PARITY_LOG=$PWD/$O/margins.txt timeout 2700 python -m pytest tests -q -m gpu --maxfail=5 --durations=8 2>&1 | tail -25
