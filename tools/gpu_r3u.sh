#!/bin/bash
# round 3: where the in-block K-split 1x1 (pwk_kernel) pays: small guided / generation batches, DR_PWK on / off
set -u
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r3u
for v in 1 0; do echo "== DR_PWK=$v guided"; DR_PWK=$v timeout 600 python tools/small_batch_ab.py --batches 1,2,3,4 2>&1 | grep "B="; done
for v in 1 0; do echo "== DR_PWK=$v generation"; DR_PWK=$v timeout 600 python tools/small_batch_ab.py --batches 1,2,4,8 --sampler generation_ddpm_x0 2>&1 | grep "B="; done
for v in 1 0; do echo "== DR_PWK=$v guided T=640"; DR_PWK=$v timeout 600 python tools/small_batch_ab.py --batches 1 --T 640 2>&1 | grep "B="; done
