#!/bin/bash
set -u
O=gpurun_out/r3e; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
b() { timeout 600 python bench.py --config $1 --no-split --no-cpu-baseline --steps ${2:-3} 2>$O/err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('cfg', j['config']['baseline_config'], j['value'], j['ms_per_step'], r['kernel'][:30], r['frac'], r['avg_launch_us'])"; }
echo "== config 7: XCD mapping model on / off"; DR_XCD_MODEL=1 b 7; DR_XCD_MODEL=0 b 7
echo "== config 6 (unchanged expected)"; DR_XCD_MODEL=1 b 6; DR_XCD_MODEL=0 b 6
echo "== config 2 per-phase sanity"; DR_STACK=0 DR_XCD_MODEL=1 b 2; DR_STACK=0 DR_XCD_MODEL=0 b 2
echo "== config 1: 32-frame direct 1x1 on / off"; DR_PW_SMALL=1 b 1 10; DR_PW_SMALL=0 b 1 10; DR_PW_SMALL=1 b 1 10; DR_PW_SMALL=0 b 1 10
echo "== ASan host library on the GPU box"
RT=$(python -c "from diffroll_amd.build import asan_runtime; print(asan_runtime())"); echo "runtime $RT"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0 timeout 120 python -c "import torch; print('torch ok', torch.cuda.is_available()); x=torch.zeros(4,device='cuda'); print('cuda ok', float(x.sum()))" > $O/asan_probe.txt 2>&1; echo "probe rc=$?"; tail -5 $O/asan_probe.txt
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0 DR_LIB=$PWD/diffroll_amd/lib/libdiffroll_amd_asan.so timeout 600 python -m pytest -q -m gpu -p no:cacheprovider "tests/test_gpu_parity.py::test_forward_golden" "tests/test_gpu_parity.py::test_frontend_golden" > $O/asan_gpu.txt 2>&1; echo "asan pytest rc=$?"; tail -5 $O/asan_gpu.txt
