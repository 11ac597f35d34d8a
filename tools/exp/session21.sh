#!/bin/bash
# k_mtf_scan_lists with phases A and B as one ranking each: tests, the single call's chain, the batch
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cudpp.py tests/test_gpu_bench_inputs.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
bash tools/exp/trace_single.sh 2>&1 | grep -E "scan_lists|chain"
python tools/exp/probe_single.py 2>/dev/null | tail -1 | cut -c1-220
for i in 1 2; do
timeout 600 python bench.py --steps 4 --warmup 1 --main-only --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j.get('kernel_ms_per_launch'))"
done
