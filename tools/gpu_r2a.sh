#!/bin/bash
# round-2 GPU pass A: parity suite with tightened tolerances, the new bench lines, the 1-rank RCCL path, and the
# conv kernel under the frame-tile-per-XCD block mapping (DR_XCD_N=1)
set -u
O=gpurun_out/r2a
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_random_chains_vs_oracle 2>&1 | tail -40 > $O/pytest.log
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench rc=$?"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline --no-split > $O/bench_cfg2_nccl1.json 2> $O/bench_cfg2_nccl1.err; echo "bench nccl1 rc=$?"
for c in 1 3 4 5; do
  timeout 600 python bench.py --config $c --no-split --no-cpu-baseline > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "bench cfg$c rc=$?"
done
timeout 300 python tools/layer_bench.py --layers 1,3 --iters 40 > $O/conv_default.txt 2>&1
DR_XCD_N=1 timeout 300 python tools/layer_bench.py --layers 1,3 --iters 40 > $O/conv_xcdn1.txt 2>&1
timeout 300 python tools/layer_bench.py --layers 1,2,3,4 --iters 20 --cycle > $O/conv_cycle_default.txt 2>&1
DR_XCD_N=1 timeout 300 python tools/layer_bench.py --layers 1,2,3,4 --iters 20 --cycle > $O/conv_cycle_xcdn1.txt 2>&1
cat $O/conv_default.txt $O/conv_xcdn1.txt $O/conv_cycle_default.txt $O/conv_cycle_xcdn1.txt
for f in $O/bench_cfg*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(j["value"], j["ms_per_step"], j.get("roofline",{}).get("frac"), j.get("whole_chain"), j.get("dist"))
except Exception as e:
    print("ERR", e)
PY
done
tail -3 $O/*.err
