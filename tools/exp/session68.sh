#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/gpu-lossless-compression_amd/variants
for lib in main net; do
  if [ $lib = main ]; then unset GLC_LIB; else export GLC_LIB=$V/libglc_$lib.so; fi
  echo "== kstats $lib pd"; bash tools/exp/kstats.sh python $GRAFT_REPO_ROOT/tools/exp/pd_batch.py 3 all 2>&1 | grep "glc::" | head -8
done
