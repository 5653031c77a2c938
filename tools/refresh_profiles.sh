#!/bin/bash
# Regenerate the rocprofv3 evidence under profiles/ - run ON THE GPU BOX:
#     gpurun --timeout 3600 -- 'bash tools/refresh_profiles.sh r06 gpurun_out/prof'
# then copy gpurun_out/prof/<round>_* into profiles/.  Counter passes are separate runs (one TCC-heavy counter set per
# pass) and never combined with tracing other than --kernel-trace; every profiler call is bounded.
set -u
RD=${1:-r06}
R=$PWD
OUT=$R/${2:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PS="python $R/tools/prof_summary.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-split --no-cold-start > "$OUT/bench_kt.log" 2>&1
echo "kernel trace rc=$?"
$PS "$OUT/kt" bench "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split --no-cold-start (MI355X, config 2: TWO launches per reverse step - stack_kernel<FL> = fused residual stack, 14 dilated convs + 15 1x1 per launch; tail_kernel = skip / output projection + combine + update + next input projection + next shared first-layer conv; gemm_kernel<NI, KS, EPI, PREC>: EPI 0 plain 1 relu 2 silu 3 gate(conv) 4 res_skip 5 power 6 log - those rows are the front-end, each chain's first input projection / first-layer conv, and the event-instrumented roofline pass)" > "$OUT/${RD}_kernel_stats.txt"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt1" -o bench -- python "$R/bench.py" --config 1 --steps 5 --warmup 2 --no-cpu-baseline --no-split --no-cold-start --no-roofline > "$OUT/bench_kt1.log" 2>&1
$PS "$OUT/kt1" bench "rocprofv3 --kernel-trace --stats -- python bench.py --config 1 --steps 5 --warmup 2 --no-cpu-baseline --no-split --no-cold-start --no-roofline (MI355X, config 1: ONE 4-s clip, 50 steps, guided: per-phase launches with split-K)" > "$OUT/${RD}_kernel_stats_cfg1.txt"
rm -rf "$OUT/kt1"
# the other BASELINE configurations at their per-GPU shape: share_of_step_time of every bench line is reproducible from these
for c in 3 4 5 6 7; do
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ktc$c" -o bench -- python "$R/bench.py" --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-split --no-cold-start > "$OUT/bench_kt$c.log" 2>&1
$PS "$OUT/ktc$c" bench "rocprofv3 --kernel-trace --stats -- python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-split --no-cold-start (MI355X, bench config $c at its per-GPU shape - 3 / 4 / 5 = the BASELINE configurations, 6 / 7 = the reference's 640-frame shipping geometry; 3 timed-or-warm-up graph chains + the event-instrumented eager roofline pass)" > "$OUT/${RD}_kernel_stats_cfg$c.txt"
rm -rf "$OUT/ktc$c"
done
for cfg in 1 2 3 4 5 6 7; do
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pf$cfg" -o pf -- python "$R/tools/step_loop.py" --config $cfg --iters 10 > "$OUT/pf$cfg.log" 2>&1
echo "fetch cfg$cfg rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pw$cfg" -o pw -- python "$R/tools/step_loop.py" --config $cfg --iters 10 > "$OUT/pw$cfg.log" 2>&1
echo "write cfg$cfg rc=$?"
python "$R/tools/make_traffic_json.py" "$OUT/pf$cfg" "$OUT/pw$cfg" $cfg "$OUT/${RD}_dominant_cfg${cfg}_traffic.json"
done
$PS "$OUT/pf2" pf "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/step_loop.py --config 2 --iters 10" > "$OUT/${RD}_stack_pmc_fetch.txt"
$PS "$OUT/pw2" pw "rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -- python tools/step_loop.py --config 2 --iters 10" > "$OUT/${RD}_stack_pmc_write.txt"
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/pm" -o pm -- python "$R/tools/step_loop.py" --config 2 --iters 10 > "$OUT/pm.log" 2>&1
echo "mfma rc=$?"
$PS "$OUT/pm" pm "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -- python tools/step_loop.py --config 2 --iters 10" > "$OUT/${RD}_stack_pmc_mfma.txt"
# ... and of the 160-frame flavour at the reference's shipping geometry (config 6)
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/pm6" -o pm -- python "$R/tools/step_loop.py" --config 6 --iters 10 > "$OUT/pm6.log" 2>&1
echo "mfma cfg6 rc=$?"
$PS "$OUT/pm6" pm "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -- python tools/step_loop.py --config 6 --iters 10" > "$OUT/${RD}_stack_pmc_mfma_cfg6.txt"
rm -rf "$OUT/pm6"
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/pl" -o pl -- python "$R/tools/step_loop.py" --config 2 --iters 10 > "$OUT/pl.log" 2>&1
echo "lds rc=$?"
$PS "$OUT/pl" pl "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace -- python tools/step_loop.py --config 2 --iters 10" > "$OUT/${RD}_stack_pmc_lds.txt"
cd "$R"
mkdir -p profiles_tmp && cp "$OUT"/${RD}_dominant_cfg*_traffic.json profiles/ 2>/dev/null   # so that bench.py finds the stamped record
for c in 1 2 3 4 5 6 7; do
  extra="--no-split --no-cpu-baseline --no-cold-start"; [ $c = 2 ] && extra=""
  timeout 900 python bench.py --config $c $extra > "$OUT/${RD}_bench_cfg$c.json" 2> "$OUT/bench_cfg$c.err"; echo "bench cfg$c rc=$?"
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-split --no-cold-start > "$OUT/${RD}_bench_cfg2_nccl_1rank.json" 2> "$OUT/bench_nccl.err"; echo "bench nccl rc=$?"
rm -rf "$OUT/kt" "$OUT"/pf[1-7] "$OUT"/pw[1-7] "$OUT/pm" "$OUT/pl" profiles_tmp
timeout 300 python tools/tail_ticks.py --config 2 > "$OUT/${RD}_tail_phase_ticks.txt" 2>&1
# (the per-phase launches of these comparisons are pinned to the fused flavour's TWINS - same conv tile width, no split-K, the
# matching 1x1 - as tests/test_gpu_fused.py does, so that "bitwise_equal" compares like with like; the planner's natural
# per-phase choice differs from any fused flavour by an ulp of accumulation order)
{ echo "# DR_TEST_TUNE=tune.ksplit_max=1,tune.tile=3202,tune.pw_nw=4 python tools/stack_check.py --config 2"
  DR_TEST_TUNE=tune.ksplit_max=1,tune.tile=3202,tune.pw_nw=4 timeout 600 python tools/stack_check.py --config 2 2>&1 | grep -v "rep [12]" | grep -v amdgpu.ids; } > "$OUT/${RD}_stack_phase_ticks.txt"
{ echo "# DR_TEST_TUNE=tune.ksplit_max=1,tune.tile=3201,tune.pw_nw=2 python tools/stack_check.py --config 3"
  DR_TEST_TUNE=tune.ksplit_max=1,tune.tile=3201,tune.pw_nw=2 timeout 600 python tools/stack_check.py --config 3 2>&1 | grep -v "rep [12]" | grep -v amdgpu.ids; } > "$OUT/${RD}_stack_phase_ticks_cfg3.txt"
# (160-frame flavour: the per-phase launches are pinned to its twins - gemm_kernel<5>, no split-K, the five-tile 1x1 - as
# tests/test_gpu_fused.py does, so that "bitwise_equal" compares like with like; the natural per-phase choice is what
# profiles/*_stack160_ab.txt times)
for c in 5 6; do
  { echo "# DR_TEST_TUNE=tune.ksplit_max=1,tune.tile=3205,tune.pw_nw=5 python tools/stack_check.py --config $c"
    DR_TEST_TUNE=tune.ksplit_max=1,tune.tile=3205,tune.pw_nw=5 timeout 600 python tools/stack_check.py --config $c 2>&1 | grep -v "rep [12]" | grep -v amdgpu.ids; } > "$OUT/${RD}_stack_phase_ticks_cfg$c.txt"
done
timeout 900 python tools/scale_table.py --gpus 1,2,4,8 --configs 2,3,4,5 --steps 2 --out "$OUT/${RD}_scale_table.json" --scale-json "$OUT/${RD}_scale.json" > "$OUT/${RD}_scale_table.txt" 2>&1
# plumbing record of the N > 1 path on the one leased GPU (gloo, every rank on device 0, per-phase launches): one launch mode per rank
timeout 900 python tools/scale_table.py --share-gpu --gpus 1,2,4,8 --configs 1 --steps 2 --out "$OUT/${RD}_scale_share_gpu_plumbing.json" > "$OUT/${RD}_scale_share_gpu_plumbing.txt" 2>&1
# time-to-first-roll: round-3 behaviour re-enabled ("before") next to the current build, fresh process each
{ for c in 1 2; do timeout 300 python -m diffroll_amd.coldstart --config $c --json --tune tune.pack_threads=1 --tune tune.s3_eager=1 2>/dev/null | grep COLD_START | sed 's/^COLD_START /{"mode": "before (serial packing, eager split-bf16 packings)", "record": /; s/$/}/'; done
  for c in 1 2; do timeout 300 python -m diffroll_amd.coldstart --config $c --json 2>/dev/null | grep COLD_START | sed 's/^COLD_START /{"mode": "now", "record": /; s/$/}/'; done; } > "$OUT/${RD}_cold_start.json"
head -14 "$OUT/${RD}_kernel_stats.txt"
cat "$OUT"/${RD}_dominant_cfg*_traffic.json
grep -A8 "stack_kernel" "$OUT/${RD}_stack_pmc_mfma.txt" | head -12
for c in 1 2 3 4 5 6 7; do python - "$OUT/${RD}_bench_cfg$c.json" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=j.get("roofline",{})
    w=j.get("whole_chain",{})
    print(sys.argv[1].split("/")[-1], j["value"], j["ms_per_step"], "roofline", r.get("frac"), "traffic", r.get("traffic"), r.get("traffic_source"), "executed", w.get("executed_frac_of_fp32_mfma_peak"), "algorithmic", w.get("algorithmic_frac_of_fp32_mfma_peak"), j.get("hbm_roofline",{}).get("frac"))
except Exception as e:
    print("ERR", sys.argv[1], e)
PY
done
