#!/bin/bash
# deep-data A/B: bash tools/exp/pd_all.sh main tag...   (two_regions halves / stretch, bench.py's partly_deep batch)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
V=$PWD/gpu-lossless-compression_amd/variants
for rep in 1 2; do for tag in "$@"; do
  if [ "$tag" = main ]; then unset GLC_LIB; else export GLC_LIB=$V/libglc_$tag.so; fi
  for k in halves stretch; do echo "$tag $(python tools/exp/two_regions.py $k 2 2>&1 | grep blocks | tail -1)"; done
  echo "$tag partly_deep $(python tools/exp/pd_batch.py 3 all 2>&1 | tail -1)"
done; done
