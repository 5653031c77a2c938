set -u
R=$PWD; OUT=$R/gpurun_out/prof4; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for cfg in 1 4 5; do
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pf$cfg" -o pf -- python "$R/tools/step_loop.py" --config $cfg --iters 10 > "$OUT/pf$cfg.log" 2>&1; echo "fetch cfg$cfg rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pw$cfg" -o pw -- python "$R/tools/step_loop.py" --config $cfg --iters 10 > "$OUT/pw$cfg.log" 2>&1; echo "write cfg$cfg rc=$?"
python "$R/tools/make_traffic_json.py" "$OUT/pf$cfg" "$OUT/pw$cfg" $cfg "$OUT/r02_dominant_cfg${cfg}_traffic.json" | cut -c1-400
rm -rf "$OUT/pf$cfg" "$OUT/pw$cfg"
done
