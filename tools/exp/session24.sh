#!/bin/bash
# k_fs_part2 with tiles of 8192 suffixes and 512 threads (16 per thread; runs of ~128 bytes): parity, the kernel, value
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/gpu-lossless-compression_amd/variants
GLC_LIB=$V/libglc_t8k.so timeout 600 python -m pytest tests/test_gpu_bucket_sorter.py tests/test_gpu_bench_inputs.py -m gpu -x -q 2>&1 | tail -2
for per in 4 8; do
  echo "== t8k per $per"; GLC_FSP2_PER=$per GLC_LIB=$V/libglc_t8k.so python tools/exp/part_probe.py 1024 4 2>/dev/null | grep -E "k_fs_part|k_fs_sort"
done
echo "== main"; python tools/exp/part_probe.py 1024 4 2>/dev/null | grep -E "k_fs_part|k_fs_sort"
for cfg in "main 0" "t8k 4" "t8k 8" "main 0" "t8k 4" "t8k 8"; do
  set -- $cfg
  if [ "$1" = main ]; then unset GLC_LIB; else export GLC_LIB=$V/libglc_$1.so; fi
  GLC_FSP2_PER=$2 timeout 600 python bench.py --steps 4 --warmup 1 --main-only --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$1 per $2', j['value'], j['value_no_stage_overlap_GBps'], j.get('kernel_ms_per_launch'))"
done
