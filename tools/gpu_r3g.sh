#!/bin/bash
set -u
O=gpurun_out/r3g; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f $O/margins.txt
DR_PARITY_LOG=$PWD/$O/margins.txt timeout 2400 python -m pytest tests -q -m gpu --maxfail=5 2>&1 | tail -12 > $O/pytest.log
echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
bash tools/checked_build.sh gpu $O > $O/checked.out 2>&1; tail -4 $O/checked_bounds.log; tail -5 $O/checked_ubsan.log
