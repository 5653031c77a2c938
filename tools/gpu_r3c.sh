#!/bin/bash
# round 3, third GPU pass: the fused step (tail kernel) - bitwise vs per-phase launches, parity suite, A/B timing
set -u
O=gpurun_out/r3c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_fused.py -q -m gpu -x 2>&1 | tail -15
rm -f $O/margins.txt
DR_PARITY_LOG=$PWD/$O/margins.txt timeout 2700 python -m pytest tests -q -m gpu --maxfail=6 --deselect tests/test_gpu_fused.py 2>&1 | tail -30 > $O/pytest.log
tail -12 $O/pytest.log
sort -k2 -g -r $O/margins.txt | grep -v trained | head -4
for c in 2 3 4; do
  timeout 600 python tools/ab_option.py fused_tail 1 0 --config $c --rounds 3 2>&1 | tail -2
done
timeout 600 python bench.py --no-split --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench rc=$?"; cat $O/bench_cfg2.json | cut -c1-1500
