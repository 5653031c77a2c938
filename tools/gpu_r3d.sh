#!/bin/bash
set -u
O=gpurun_out/r3d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/tail_ticks.py --config 2 2>&1 | tail -14
timeout 600 python tools/tail_ticks.py --config 3 2>&1 | tail -10
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -o sl -- python $GRAFT_REPO_ROOT/tools/step_loop.py --config 2 --iters 10 > $GRAFT_REPO_ROOT/$O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $O/kt sl "step_loop config 2 (single steps: tail without next in-proj)" | head -20
timeout 900 python -m pytest tests/test_gpu_fused.py -q -m gpu -x 2>&1 | tail -3
