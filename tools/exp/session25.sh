#!/bin/bash
# k_fs_part2 with tiles of 8192 suffixes and 1024 threads: value with and without stage overlap, tiles per workgroup 2 / 4 / 8
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/gpu-lossless-compression_amd/variants
for cfg in "main 0" "t8k1k 2" "t8k1k 4" "t8k1k 8" "main 0" "t8k1k 2" "t8k1k 4"; do
  set -- $cfg
  if [ "$1" = main ]; then unset GLC_LIB; else export GLC_LIB=$V/libglc_$1.so; fi
  GLC_FSP2_PER=$2 timeout 600 python bench.py --steps 4 --warmup 1 --main-only --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$1 per $2', j['value'], j['value_no_stage_overlap_GBps'], j.get('kernel_ms_per_launch'))"
done
