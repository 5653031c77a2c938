#!/bin/bash
# round-2 GPU pass C: full parity suite with the fused kernel as default + the fused-vs-per-phase tests, margins logged
set -u
O=gpurun_out/r2c
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f $O/margins.txt
DR_PARITY_LOG=$PWD/$O/margins.txt timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest.log
echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
sort -k2 -g -r $O/margins.txt | head -25
timeout 600 python bench.py --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench rc=$?"
timeout 600 python bench.py --config 3 --no-split --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench3 rc=$?"
for f in $O/bench_cfg*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], j["value"], j["ms_per_step"], j.get("roofline"), j.get("whole_chain"))
except Exception as e:
    print("ERR", e)
PY
done
