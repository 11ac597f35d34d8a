#!/bin/bash
# single call: tiles per workgroup of k_fs_part2 for a call of one block (GLC_FSP2_PER1 1/2/4), k_fs_ties meeting 8 members at a time
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/gpu-lossless-compression_amd/variants
for cfg in "4 main" "1 main" "2 main" "1 w8"; do
  set -- $cfg
  echo "=== GLC_FSP2_PER1=$1 lib=$2"
  if [ "$2" = main ]; then unset GLC_LIB; else export GLC_LIB=$V/libglc_$2.so; fi
  GLC_FSP2_PER1=$1 bash tools/exp/trace_single.sh 2>&1 | grep -E "k_fs_part2|k_fs_ties|chain|ms_per_call"
  GLC_FSP2_PER1=$1 python tools/exp/probe_single.py 2>/dev/null | tail -1 | cut -c1-200
done
unset GLC_LIB
echo "=== batch value: ties 4 vs 8"
for lib in main w8 main w8; do
  if [ "$lib" = main ]; then unset GLC_LIB; else export GLC_LIB=$V/libglc_$lib.so; fi
  timeout 600 python bench.py --steps 4 --warmup 1 --main-only --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$lib', j['value'], j.get('kernel_ms_per_launch'))"
done
