#!/bin/bash
# all five bench lines (+ the 1-rank RCCL run) into gpurun_out/benchall: copy to profiles/r02_bench_*.json
set -u
O=gpurun_out/benchall; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for c in 1 2 3 4 5; do
  extra="--no-split --no-cpu-baseline --no-cold-start"; [ $c = 2 ] && extra=""
  timeout 900 python bench.py --config $c $extra > $O/r02_bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "bench cfg$c rc=$?"
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-split --no-cold-start > $O/r02_bench_cfg2_nccl_1rank.json 2> $O/bench_nccl.err; echo "nccl rc=$?"
for f in $O/r02_bench_*.json; do python - "$f" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j["roofline"]
print(sys.argv[1].split("/")[-1], j["value"], j["ms_per_step"], r["frac"], r["traffic"], j["whole_chain"]["frac_of_fp32_mfma_peak"], j["hbm_roofline"]["frac"], j.get("dist",{}).get("backend"))
PY
done
