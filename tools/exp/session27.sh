#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_r05.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_r05.log | tail -2
for i in 1 2; do
for per in 16 32 12; do
  GLC_FSP2_PER=$per timeout 600 python bench.py --steps 6 --warmup 1 --main-only --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('per $per', j['value'], j['value_no_stage_overlap_GBps'], j['kernel_ms_per_launch']['k_fs_part'])"
done; done
