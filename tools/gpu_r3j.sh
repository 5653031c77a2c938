#!/bin/bash
# round 3: socket power and shader clock while the fp32 chain of config 2 / 3 runs (is config 3's lower clock a power limit?)
set -u
O=gpurun_out/r3j; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for c in 2 3; do echo "== config $c"; bash tools/power_probe.sh f32 $c 2>&1 | tee -a $O/power_cfg$c.txt; done
echo "== config 3, spread mapping (weight panel per XCD)"; DR_STACK_XCD=0 bash tools/power_probe.sh f32 3 2>&1 | tee $O/power_cfg3_xcd0.txt
