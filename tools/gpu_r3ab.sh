#!/bin/bash
# round 3: tile choice aware of the multi-round split-K: part-filled launches at 640 / 320 frames, and the 640-frame bench configurations
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r3ab; mkdir -p $O
echo "== T=640 guided"; timeout 600 python tools/small_batch_ab.py --batches 1,2,3,4 --T 640 2>&1 | grep "B="
echo "== T=320 guided"; timeout 600 python tools/small_batch_ab.py --batches 1,2,3,5,6 --T 320 2>&1 | grep "B="
echo "== T=320 guided, old rules"; DR_KSPLIT_BLOCKS=256 timeout 600 python tools/small_batch_ab.py --batches 1,2,3,5,6 --T 320 2>&1 | grep "B="
echo "== T=125 guided"; timeout 600 python tools/small_batch_ab.py --batches 3,5,6,10,12 2>&1 | grep "B="
for c in 5 6 7; do timeout 600 python bench.py --config $c --no-cpu-baseline --no-split > $O/bench_cfg$c.json 2>$O/err.txt; python - $O/bench_cfg$c.json $c <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); r = j["roofline"]
print(f"config {sys.argv[2]}: {j['ms_per_step']:.1f} ms/chain, {r['kernel'][:30]} {r['avg_launch_us']:.1f} us, frac {r['frac']}")
PY
done
