#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/gpu-lossless-compression_amd/variants
GLC_LIB=$V/libglc_t16k.so timeout 600 python -m pytest tests/test_gpu_bucket_sorter.py tests/test_gpu_bench_inputs.py -m gpu -x -q 2>&1 | tail -1
for i in 1 2; do
for cfg in "main 16" "t16k 8" "t16k 4"; do
  set -- $cfg
  if [ "$1" = main ]; then unset GLC_LIB; else export GLC_LIB=$V/libglc_$1.so; fi
  GLC_FSP2_PER=$2 timeout 600 python bench.py --steps 6 --warmup 1 --main-only --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$1 per $2', j['value'], j['value_no_stage_overlap_GBps'], j['kernel_ms_per_launch']['k_fs_part'])"
done; done
