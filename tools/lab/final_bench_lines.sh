set -u
O=gpurun_out/pass5; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_sharding.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" > $O/sharding.txt; cat $O/sharding.txt
for c in 1 2 3 4 5 6 7; do
  extra="--no-split --no-cpu-baseline --no-cold-start"; [ $c = 2 ] && extra=""
  timeout 900 python bench.py --config $c $extra > "$O/r06_bench_cfg$c.json" 2> "$O/bench_cfg$c.err"; echo "bench cfg$c rc=$?"
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-split --no-cold-start > "$O/r06_bench_cfg2_nccl_1rank.json" 2> "$O/bench_nccl.err"; echo "bench nccl rc=$?"
{
echo "# the GPU suite with every engine's DEFAULTS changed (tools/tuning_env.py), one MI355X, round-6 final sources"
for tune in fused_stack=0 tune.stack_fl=-5; do
  echo "##### DR_TEST_TUNE=$tune"
  DR_TEST_TUNE=$tune timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR"
done
} > "$O/r06_forced_mode_suites.txt" 2>&1
cat "$O/r06_forced_mode_suites.txt"
bash tools/checked_build.sh gpu "$O/checked" > "$O/checked.log" 2>&1
cp "$O/checked/checked_ubsan.log" "$O/r06_checked_ubsan_gpu.log"; tail -4 "$O/r06_checked_ubsan_gpu.log"
for c in 1 2 3 4 5 6 7; do python - "$O/r06_bench_cfg$c.json" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=j.get("roofline",{})
print(sys.argv[1].split("/")[-1], j["value"], j["ms_per_step"], r.get("frac"), j["launch_mode"], j["attempts"], j["fused_yields"], r.get("traffic_source"))
PY
done
