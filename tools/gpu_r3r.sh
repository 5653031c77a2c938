#!/bin/bash
# round 3: the in-block K-split 1x1 kernel for under-filled launches (pwk_kernel): whole suite + config 1 A/B
set -u
O=gpurun_out/r3r; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2700 python -m pytest tests -q -m gpu --maxfail=8 2>&1 | tail -15
for rep in 1 2; do for v in 1 0; do
  DR_PWK=$v timeout 300 python bench.py --config 1 --no-cpu-baseline --no-split > $O/bench_cfg1_pwk${v}_$rep.json 2>$O/err.txt
  python - $O/bench_cfg1_pwk${v}_$rep.json $v <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); r = j["roofline"]
print(f"config 1 DR_PWK={sys.argv[2]}: {j['ms_per_step']:.2f} ms/chain, {j['value']:.0f} frames/s, executed frac {j['whole_chain']['executed_frac_of_fp32_mfma_peak']}")
PY
done; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/kt1 -o bench -- python $OLDPWD/bench.py --config 1 --steps 5 --warmup 2 --no-cpu-baseline --no-split --no-roofline > $OLDPWD/$O/kt1.log 2>&1
cd $OLDPWD
python tools/prof_summary.py $O/kt1 bench "config 1 with pwk_kernel" | head -14
rm -rf $O/kt1
