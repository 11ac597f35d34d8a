#!/bin/bash
# same-box A/B of the text-like path: bash tools/exp/ab_text.sh [-k] main tagA tagB ...   (variants/libglc_<tag>.so; -k: per-kernel table too)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
V=$PWD/gpu-lossless-compression_amd/variants
KS=0; [ "$1" = "-k" ] && { KS=1; shift; }
for rep in 1 2; do for tag in "$@"; do
  if [ "$tag" = main ]; then unset GLC_LIB; else export GLC_LIB=$V/libglc_$tag.so; fi
  for kind in text256 log256; do
    echo "$tag $kind $(python tools/exp/text_batch.py $kind 256 4 2>&1 | grep 'batch of' | awk '{print $5}' | sort -n | head -1) ms"
  done
done; done
if [ $KS = 1 ]; then for tag in "$@"; do
  if [ "$tag" = main ]; then unset GLC_LIB; else export GLC_LIB=$V/libglc_$tag.so; fi
  for kind in text256 log256; do echo "== $tag $kind"; bash tools/exp/kstats.sh python $PWD/tools/exp/text_batch.py $kind 256 3 | grep "glc::" | head -12; done
done; fi
