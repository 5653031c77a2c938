#!/bin/bash
set -u
O=gpurun_out/r2d
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 500 python tools/stack_check.py > $O/stack.txt 2>&1; echo "stack rc=$?"
grep -v "rep " $O/stack.txt | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_sharding.py tests/test_gpu_fused.py -m gpu -q 2>&1 | tail -15 > $O/pytest.log; tail -8 $O/pytest.log
timeout 900 python tools/host_feed.py --config 2 --procs 8 --chains 3 > $O/host_feed_cfg2.txt 2>&1; cat $O/host_feed_cfg2.txt
timeout 600 python tools/host_feed.py --config 1 --procs 8 --chains 5 > $O/host_feed_cfg1.txt 2>&1; cat $O/host_feed_cfg1.txt
