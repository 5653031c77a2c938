#!/bin/bash
# round 3: A/B of the fused kernel's 1x1 phases on all eight waves (measurement build libdiffroll_amd_pw8.so)
set -u
O=gpurun_out/r3l; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
PW8=$PWD/diffroll_amd/lib/libdiffroll_amd_pw8.so
echo "== bitwise tests on the pw8 build"
DR_LIB=$PW8 timeout 1200 python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | tail -3
for rep in 1 2; do for c in 2 3; do
  for lib in base pw8; do
    if [ $lib = pw8 ]; then export DR_LIB=$PW8; else unset DR_LIB; fi
    timeout 600 python bench.py --config $c --no-cpu-baseline --no-split > $O/bench_cfg${c}_${lib}_$rep.json 2>$O/err.txt
    python - $O/bench_cfg${c}_${lib}_$rep.json $c $lib <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); r = j["roofline"]
print(f"config {sys.argv[2]} {sys.argv[3]:>5}: {j['ms_per_step']:.1f} ms/chain, {r['kernel'][:16]} {r['avg_launch_us']:.1f} us, frac {r['frac']}")
PY
  done; done; done
unset DR_LIB
for c in 2 3; do DR_LIB=$PW8 timeout 600 python tools/stack_check.py --config $c > $O/stack_check_pw8_cfg$c.txt 2>&1; grep -E "phase ticks|MHz|a 1x1" $O/stack_check_pw8_cfg$c.txt | cut -c1-330; done
