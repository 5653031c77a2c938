#!/bin/bash
# the driver's round-end sequence on one box: GPU test suite (margins logged), smoke, default bench
set -u
O=gpurun_out/full
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f $O/margins.txt
DR_PARITY_LOG=$PWD/$O/margins.txt timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path\|amdgpu.ids" | tail -25 > $O/pytest.log
echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
sort -k2 -g -r $O/margins.txt | head -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/full/bench.json').read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["traffic"], j["roofline"]["traffic_source"], j["whole_chain"], j["cpu_baseline"]["value"], j["split_bf16x3"]["value"])
PY
