#!/bin/bash
# A/B of the 160-frame fused residual stack (stack_kernel<5>, round 6) against the per-phase launches it replaces, at the
# 640-frame geometries (bench configs 5 / 6 / 7) - run ON THE GPU BOX:
#     gpurun --timeout 1500 -- 'bash tools/lab/stack160_ab.sh gpurun_out/r06_stack160_ab.txt'
# tune.stack_fl = -5 excludes the flavour (the planner then picks what round 5 ran: per-phase launches); 0 = automatic.
set -u
OUT=${1:-gpurun_out/r06_stack160_ab.txt}
mkdir -p "$(dirname "$OUT")"
{
echo "# stack_kernel<5> (128 rows x 160 frames per block, fused residual stack for 640-frame geometries) vs per-phase launches"
echo "# (1) alternating whole captured chains in ONE process: python tools/lab/ab_option.py tune.stack_fl 0 -5 --config N --rounds 4"
for c in 5 6 7; do
  echo "## config $c"
  timeout 600 python tools/lab/ab_option.py tune.stack_fl 0 -5 --config $c --rounds 4 2>&1 | grep -v "^\[diffroll_amd\]"
done
echo "# (2) bench lines (fresh process each): value, ms per chain, launch mode, dominant kernel + frac, whole chain executed"
for c in 5 6 7; do
  for fl in 0 -5; do
    DR_TEST_TUNE=tune.stack_fl=$fl timeout 600 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-split --no-cold-start 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{') and '\"metric\"' in ln:
        j=json.loads(ln); r=j.get('roofline',{}); w=j.get('whole_chain',{})
        print('config $c stack_fl=$fl: %.1f frames/s, %.2f ms/chain, mode %s, yields %s, kernel %s: %.2f us x %d launches, frac %.4f, share %.3f; whole chain executed %.4f algorithmic %.4f' % (j['value'], j['ms_per_step'], j['launch_mode'], j['fused_yields'], r.get('kernel','?')[:28], r.get('avg_launch_us',0), r.get('launches',0), r.get('frac',0), r.get('share_of_step_time',0), w.get('executed_frac_of_fp32_mfma_peak',0), w.get('algorithmic_frac_of_fp32_mfma_peak',0)))
"
  done
done
} > "$OUT" 2>&1
cat "$OUT"
