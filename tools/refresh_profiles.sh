#!/bin/bash
# Regenerate the rocprofv3 evidence under profiles/ - run ON THE GPU BOX:
#     gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh gpurun_out/prof'
# then copy gpurun_out/prof/r01_* into profiles/.  Counter passes are separate runs (one TCC-heavy counter
# set per pass) and never combined with tracing other than --kernel-trace; every profiler call is bounded.
set -u
R=$PWD
OUT=$R/${1:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_kt.log" 2>&1
echo "rc=$?"
python "$R/tools/prof_summary.py" "$OUT/kt" bench "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline (MI355X; gemm_kernel<NI, KS, EPI, PREC>: EPI 0 plain 1 relu 2 silu 3 gate(conv) 4 res_skip(1x1) 5 power 6 log; PREC 1 = split-bf16 rows of the extra split_bf16x3 measurement; pw_kernel<NW> = 1x1 residual/skip GEMM, fp32)" > "$OUT/r01_kernel_stats.txt"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pf" -o pf -- python "$R/tools/layer_bench.py" --iters 10 --layers 1,3 > "$OUT/pf.log" 2>&1
echo "rc=$?"
python "$R/tools/prof_summary.py" "$OUT/pf" pf "rocprofv3 --pmc FETCH_SIZE -- python tools/layer_bench.py --iters 10 --layers 1,3" > "$OUT/r01_conv_pmc_fetch.txt"
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pw" -o pw -- python "$R/tools/layer_bench.py" --iters 10 --layers 1,3 > "$OUT/pw.log" 2>&1
echo "rc=$?"
python "$R/tools/prof_summary.py" "$OUT/pw" pw "rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python tools/layer_bench.py --iters 10 --layers 1,3" > "$OUT/r01_conv_pmc_write.txt"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/pm" -o pm -- python "$R/tools/layer_bench.py" --iters 10 --layers 1,3 > "$OUT/pm.log" 2>&1
echo "rc=$?"
python "$R/tools/prof_summary.py" "$OUT/pm" pm "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -- python tools/layer_bench.py --iters 10 --layers 1,3" > "$OUT/r01_conv_pmc_mfma.txt"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/pl" -o pl -- python "$R/tools/layer_bench.py" --iters 10 --layers 1,3 > "$OUT/pl.log" 2>&1
echo "rc=$?"
python "$R/tools/prof_summary.py" "$OUT/pl" pl "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace -- python tools/layer_bench.py --iters 10 --layers 1,3" > "$OUT/r01_conv_pmc_lds.txt"
cd "$R"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "rc=$?"
timeout 600 python tools/config_bench.py > "$OUT/r01_config_bench.txt" 2>&1
rm -rf "$OUT/kt" "$OUT/pf" "$OUT/pw" "$OUT/pm" "$OUT/pl"
grep -A3 "gemm_kernel<2, 1, 3, 0>" "$OUT/r01_conv_pmc_fetch.txt" "$OUT/r01_conv_pmc_write.txt" "$OUT/r01_conv_pmc_mfma.txt" | grep -v "^--" | head -30
head -12 "$OUT/r01_kernel_stats.txt"
cat "$OUT/bench.json" "$OUT/r01_config_bench.txt"
