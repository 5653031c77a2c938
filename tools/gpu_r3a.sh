#!/bin/bash
# round 3, first GPU pass: whole GPU suite (margins logged), bench lines of configs 2 / 3 / 6 / 7, block-mapping A/B at
# config 3, fused vs per-phase at the 640-frame geometries
set -u
O=gpurun_out/r3a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f $O/margins.txt
DR_PARITY_LOG=$PWD/$O/margins.txt timeout 2700 python -m pytest tests -q -m gpu --maxfail=10 --durations=15 2>&1 | tail -80 > $O/pytest.log
echo "pytest rc=$?" >> $O/pytest.log
tail -45 $O/pytest.log
sort -k2 -g -r $O/margins.txt | head -5
grep trained_regime $O/margins.txt | sort -t' ' -k7 -g -r | head -12
for c in 2 3 6 7; do
  timeout 900 python bench.py --config $c --no-split --no-cpu-baseline > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "bench cfg$c rc=$?"
  python - $O/bench_cfg$c.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j["roofline"]
    print(sys.argv[1].split("/")[-1], j["value"], j["ms_per_step"], r["kernel"][:60], r["frac"], r["avg_launch_us"], j["whole_chain"]["executed_frac_of_fp32_mfma_peak"], j["whole_chain"]["algorithmic_frac_of_fp32_mfma_peak"], j["per_rank_ms_per_step"], j["gather_us"])
except Exception as e:
    print("ERR", sys.argv[1], e)
PY
done
timeout 600 python tools/ab_option.py fused_stack_xcd 1 0 --config 3 --rounds 3 2>&1 | tail -3
timeout 600 python tools/ab_option.py fused_stack 1 0 2 --config 6 --rounds 2 2>&1 | tail -4
timeout 600 python tools/ab_option.py fused_stack 1 0 2 --config 7 --rounds 2 2>&1 | tail -4
