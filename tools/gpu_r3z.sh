#!/bin/bash
# round 3, final: whole GPU suite on the final sources (observed margins logged)
set -u
O=gpurun_out/r3z; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f $O/margins.txt
DR_PARITY_LOG=$PWD/$O/margins.txt timeout 2700 python -m pytest tests -q -m gpu --maxfail=5 --durations=5 2>&1 | tail -14
