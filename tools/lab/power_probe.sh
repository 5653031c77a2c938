#!/bin/bash
# Sample socket power and shader clock (rocm-smi) while a chain runs:  bash tools/lab/power_probe.sh [f32|bf16x3] [config]
PREC=${1:-f32}
CFG=${2:-2}
python - "$PREC" "$CFG" <<'PY' &
import sys, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda', 0)
cfg = bench.CONFIGS[int(sys.argv[2])]
hp = dict(bench.HP); hp.update(kernel_size=cfg["k"], timesteps=cfg["S"])
m = bench.build_model(dev, hp=hp, sampler=cfg["sampler"]); m.precision = sys.argv[1]
g = torch.Generator().manual_seed(0)
B, T = cfg["B"], cfg["L"] // 512
wav = (0.1 * torch.randn(B, cfg["L"], generator=g)).to(dev); x = torch.randn(B, 1, T, 88, generator=g).to(dev)
m.sample(x, wav, seed=0); torch.cuda.synchronize()
t0 = time.perf_counter()
N = max(4, int(10e3 / max(cfg['B'] * T / 4.0, 1)))
N = 12 if int(sys.argv[2]) in (2, 4) else (24 if int(sys.argv[2]) == 3 else 8)
for _ in range(N): m.sample(x, wav, seed=0)
torch.cuda.synchronize()
print("chain ms", (time.perf_counter() - t0) / N * 1e3, flush=True)
PY
PID=$!
sleep 6
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ' '; echo; sleep 0.7; done
wait $PID
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
