#!/bin/bash
# round 3: the whole GPU suite in the default configuration and with the non-default engine modes forced
set -u
O=gpurun_out/r3h; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { echo "== $1"; shift; env "$@" timeout 2400 python -m pytest tests -q -m gpu --maxfail=5 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -6; }
rm -f $O/margins.txt
run "default" DR_PARITY_LOG=$PWD/$O/margins.txt
run "spread block mapping (every hand-off cross-XCD)" DR_STACK_XCD=0
run "tail fusion off" DR_TAIL=0
run "per-phase launches only" DR_STACK=0
