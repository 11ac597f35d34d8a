#!/bin/bash
# One parameterised GPU-box session (replaces the 28 one-off session<N>.sh of rounds 4-5; those are in the history up to
# commit 1a2090e): for every variant library tag, optionally the tests that cover what the variant touches, then `bench.py`
# A/B against the in-tree library, alternating, twice.
#   bash tools/exp/session.sh [-t "tests/test_gpu_bucket_sorter.py tests/test_gpu_bench_inputs.py"] [-k k_fs_part] [-e "GLC_FSP2_PER=8"] main tagA tagB ...
# Variants: GLC_CXXFLAGS="-D..." GLC_LIB_OUT=$PWD/gpu-lossless-compression_amd/variants/libglc_<tag>.so python gpu-lossless-compression_amd/build.py
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
V=$PWD/gpu-lossless-compression_amd/variants
TESTS=""; KEY=k_fs_sort; EXTRA=""
while getopts "t:k:e:" o; do case $o in t) TESTS=$OPTARG;; k) KEY=$OPTARG;; e) EXTRA=$OPTARG;; *) exit 2;; esac; done
shift $((OPTIND - 1))
for tag in "$@"; do
  [ "$tag" = main ] && continue
  [ -n "$TESTS" ] && GLC_LIB=$V/libglc_$tag.so timeout 900 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -1
done
for i in 1 2; do for tag in "$@"; do
  if [ "$tag" = main ]; then unset GLC_LIB; else export GLC_LIB=$V/libglc_$tag.so; fi
  env $EXTRA timeout 600 python bench.py --steps 6 --warmup 1 --main-only --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$tag', j['value'], j.get('value_no_stage_overlap_GBps'), j.get('kernel_ms_per_launch', {}).get('$KEY'))"
done; done
