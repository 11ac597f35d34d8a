#!/bin/bash
# 8192-suffix tiles, 1024 threads: tiles per workgroup 8 / 16 against the default, alternating, three times
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/gpu-lossless-compression_amd/variants
for i in 1 2 3; do
for cfg in "main 0" "t8k1k 8" "t8k1k 16"; do
  set -- $cfg
  if [ "$1" = main ]; then unset GLC_LIB; else export GLC_LIB=$V/libglc_$1.so; fi
  GLC_FSP2_PER=$2 timeout 600 python bench.py --steps 6 --warmup 1 --main-only --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$1 per $2', j['value'], j['value_no_stage_overlap_GBps'], j['kernel_ms_per_launch']['k_fs_part'])"
done; done
