#!/bin/bash
# k_ss_sample: integer sort + ranks inside runs (main) against the network with text comparisons (variants/libglc_net.so)
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/gpu-lossless-compression_amd/variants
timeout 900 python -m pytest tests/test_gpu_sample_sorter.py tests/test_gpu_resume.py tests/test_gpu_bench_inputs.py tests/test_gpu_periodic.py -m gpu -x -q 2>&1 | tail -3
for round in 1 2; do
for lib in main net; do
  if [ $lib = main ]; then unset GLC_LIB; else export GLC_LIB=$V/libglc_$lib.so; fi
  echo "== $lib text256 $(timeout 300 python tools/exp/text_batch.py text256 256 4 2>&1 | tail -1)"
  echo "== $lib log256 $(timeout 300 python tools/exp/text_batch.py log256 256 4 2>&1 | tail -1)"
  echo "== $lib pd $(timeout 300 python tools/exp/pd_batch.py 4 all 2>&1 | tail -1)"
  echo "== $lib single text $(timeout 300 python tools/exp/single_timing.py text 2>&1 | tail -1)"
done
done
unset GLC_LIB
echo "== kstats main text256"; bash tools/exp/kstats.sh python $GRAFT_REPO_ROOT/tools/exp/text_batch.py text256 256 3 2>&1 | grep "glc::" | head -12
echo "== kstats main pd"; bash tools/exp/kstats.sh python $GRAFT_REPO_ROOT/tools/exp/pd_batch.py 3 all 2>&1 | grep "glc::" | head -16
