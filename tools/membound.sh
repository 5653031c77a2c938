#!/bin/bash
# SURVEY 8(d) item 2 on the GPU box: kernel traces of every memory-bound kernel at the BASELINE geometries.
#     gpurun --timeout 900 -- 'bash tools/membound.sh r05 gpurun_out/membound'
set -u
RD=${1:-r05}
R=$PWD
OUT=$R/${2:-gpurun_out/membound}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for sz in cfg2 cfg3 cfg5 cfg7 large; do
  mkdir -p "$OUT/$sz"
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$sz/kt" -o mb -- python "$R/tools/membound_loop.py" --size $sz --reps 20 > "$OUT/$sz/loop.log" 2>&1
  echo "$sz rc=$?"
done
cd "$R"
python tools/membound_report.py "$OUT" "$OUT/${RD}_membound_kernels.json" > "$OUT/${RD}_membound_kernels.txt"
cat "$OUT/${RD}_membound_kernels.txt"
for sz in cfg2 cfg3 cfg5 cfg7 large; do rm -rf "$OUT/$sz/kt"; done
