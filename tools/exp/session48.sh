#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do for r in 1024 2048 4096; do
timeout 600 python bench.py --rows $r --steps 6 --warmup 1 --main-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('rows', j['config']['batch_rows'], j['value'], j['value_no_stage_overlap_GBps'], j['kernel_ms_per_launch'])"
done; done
