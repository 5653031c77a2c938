#!/bin/bash
# round 3: (a) is config 3's lower effective clock tied to the weight-fragment traffic?  (measurement build without the
# conv's A loads: wrong results, same MFMA stream)  (b) the 160-frame fused flavour with its 1x1 on eight waves
set -u
O=gpurun_out/r3m; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for c in 3 2; do
  DR_LIB=$PWD/diffroll_amd/lib/libdiffroll_amd_ablate1.so timeout 600 python tools/stack_check.py --config $c > $O/stack_check_ablate1_cfg$c.txt 2>&1
  echo "== config $c, no A loads"; grep -E "MHz|^chain" $O/stack_check_ablate1_cfg$c.txt | cut -c1-200
done
for c in 6; do
  DR_STACK_FL=5 DR_LIB=$PWD/diffroll_amd/lib/libdiffroll_amd_pw8.so timeout 900 python tools/stack_check.py --config $c --level 2 > $O/stack_check_fl5_pw8_cfg$c.txt 2>&1
  echo "== config $c FL=5 pw8"; grep -E "MHz|^chain|phase ticks" $O/stack_check_fl5_pw8_cfg$c.txt | cut -c1-260
done
