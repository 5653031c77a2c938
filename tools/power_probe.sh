#!/bin/bash
# Sample socket power and shader clock (rocm-smi) while a chain runs:  bash tools/power_probe.sh [f32|bf16x3]
PREC=${1:-f32}
python - "$PREC" <<'PY' &
import sys, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda', 0)
m = bench.build_model(dev); m.precision = sys.argv[1]
g = torch.Generator().manual_seed(0)
wav = (0.1 * torch.randn(16, 64000, generator=g)).to(dev); x = torch.randn(16, 1, 125, 88, generator=g).to(dev)
m.sample(x, wav, seed=0); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(12): m.sample(x, wav, seed=0)
torch.cuda.synchronize()
print("chain ms", (time.perf_counter() - t0) / 12 * 1e3, flush=True)
PY
PID=$!
sleep 6
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ' '; echo; sleep 0.7; done
wait $PID
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
