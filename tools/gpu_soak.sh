#!/bin/bash
# the repeatability soaks on the final build of a round (profiles/rNN_soak.txt): bash tools/gpu_soak.sh [rNN]
set -u
O=gpurun_out/soak; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
RD=${1:-r06}; S=$O/${RD}_soak.txt; : > $S
{
echo "# repeatability soaks on the $RD build (one MI355X); every line is a tool's own summary"
echo "## tools/fused_soak.py --chains 30 --config 2   (30 x 200 fused launches + 30 x 200 tail launches, same seed: bitwise equal rolls, no time-out)"
timeout 600 python tools/fused_soak.py --chains 30 --config 2 2>&1 | tail -3
echo "## tools/fused_soak.py --chains 30 --config 3"
timeout 600 python tools/fused_soak.py --chains 30 --config 3 2>&1 | tail -3
echo "## tools/fused_soak.py --chains 20 --config 6   (160-frame blocks: 20 x 200 stack_kernel<5> launches + tail launches with 96-frame T4 items)"
timeout 600 python tools/fused_soak.py --chains 20 --config 6 2>&1 | tail -3
echo "## tools/fused_soak.py --chains 12 --config 5   (k = 15)"
timeout 600 python tools/fused_soak.py --chains 12 --config 5 2>&1 | tail -3
echo "## tools/determinism_soak.py 600   (split-K path: small launches, every chain twice)"
timeout 900 python tools/determinism_soak.py 600 2>&1 | tail -3
echo "## tools/xcd_stress.py: block mapping 0 (groups spread over the XCDs) must reproduce mapping 1 bit for bit"
for T in 500 640; do
  echo "# DR_TEST_TUNE=tune.stack_fl=2 --T $T --reps 40"
  DR_TEST_TUNE=tune.stack_fl=2 timeout 600 python tools/xcd_stress.py --T $T --B 4 --reps 40 2>&1 | tail -2
done
echo "# DR_TEST_TUNE=tune.stack_fl=5 --T 640 --reps 40 (160-frame flavour: 32-block groups)"; DR_TEST_TUNE=tune.stack_fl=5 timeout 600 python tools/xcd_stress.py --T 640 --B 4 --reps 40 2>&1 | tail -2
echo "# DR_TEST_TUNE=tune.stack_fl=5 --T 640 --k 15 --reps 20"; DR_TEST_TUNE=tune.stack_fl=5 timeout 600 python tools/xcd_stress.py --T 640 --B 4 --k 15 --reps 20 2>&1 | tail -2
echo "# DR_TEST_TUNE=tune.stack_fl=5 --T 640 --chain 12 --reps 8 (tail kernel on 160-frame blocks)"; DR_TEST_TUNE=tune.stack_fl=5 timeout 600 python tools/xcd_stress.py --T 640 --B 4 --chain 12 --reps 8 2>&1 | tail -2
echo "# --T 250 --B 8 --reps 40 (64-frame flavour)"; timeout 600 python tools/xcd_stress.py --T 250 --B 8 --reps 40 2>&1 | tail -2
echo "# --T 500 --chain 20 --reps 10 (whole chains, tail kernel)"; timeout 900 python tools/xcd_stress.py --T 500 --B 4 --chain 20 --reps 10 2>&1 | tail -2
echo "## part-filled launches (split-K beyond one resident round): whole captured chains twice, bitwise"
timeout 600 python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
hp = dict(bench.HP); hp.update(timesteps=20)
m = bench.build_model(dev, hp=hp, sampler="cfdg_ddpm_x0")
g = torch.Generator().manual_seed(21)
bad = 0; n = 0
for B, T in ((3, 125), (5, 125), (6, 125), (10, 125), (12, 125), (20, 125), (1, 640), (2, 640), (2, 320)):
    wav = (0.1 * torch.randn(B, T * 512, generator=g)).to(dev); x = torch.randn(B, 1, T, 88, generator=g).to(dev)
    ref = m.sample(x, wav, seed=4)[0].clone()
    for rep in range(4):
        m._fe_key = None
        n += 1; bad += int(not torch.equal(m.sample(x, wav, seed=4)[0], ref))
print(f"part-filled chains: {n} repeats of 9 geometries, {bad} differ; fallbacks {m.engine.fallbacks}")
PY
echo "# --T 125 --B 9 --reps 40 (padded launch, 9 guided clips)"; timeout 600 python tools/xcd_stress.py --T 125 --B 9 --reps 40 2>&1 | tail -2
echo "## LITMUS builds (WRONG on purpose): the same stress must FAIL.  -DDR_FAULT=1 = the hand-over barrier without its s_waitcnt vmcnt(0) (the round-3 race), -DDR_FAULT=2 = hand-offs with plain stores across XCDs"
for v in fault1 fault2; do
  L=$(python -m diffroll_amd.build --variant=$v | tail -1)
  for i in 1 2 3 4 5 6 7 8 9 10; do
    echo "# $v run $i: tune.stack_fl=2 --T 640 --reps 24"
    DR_LIB=$L DR_TEST_TUNE=blocked_accumulation=2,tune.stack_fl=2 timeout 600 python tools/xcd_stress.py --T 640 --B 4 --reps 24 2>&1 | grep -E "RESULT|mapping 1 repeatable"
    if [ $i -le 3 ]; then
      echo "# $v run $i: tune.stack_fl=5 --T 640 --reps 24 (160-frame flavour)"
      DR_LIB=$L DR_TEST_TUNE=tune.stack_fl=5 timeout 600 python tools/xcd_stress.py --T 640 --B 4 --reps 24 2>&1 | grep -E "RESULT|mapping 1 repeatable"
    fi
    # (fault1 = a bare s_barrier in the producers: tools/isa_audit.py -DDR_FAULT=1 finds the missing wait in every kernel
    # with an LDS-DMA hand-over; the whole-chain form runs the tail kernel too)
    if [ $v = fault1 ]; then
      echo "# $v run $i: --T 500 --chain 12 --reps 6 (tail kernel)"
      DR_LIB=$L timeout 600 python tools/xcd_stress.py --T 500 --B 4 --chain 12 --reps 6 2>&1 | grep -E "RESULT|mapping 1 repeatable"
    fi
  done
done
} >> $S 2>&1
cat $S
