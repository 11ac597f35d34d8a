#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/gpu-lossless-compression_amd/variants
for k in text256 log256; do
echo "== clk $k"; GLC_LIB=$V/libglc_clk.so timeout 300 python tools/exp/ss_clocks.py $k 256 2>&1 | grep "k_ss_sample" | tail -1
done
for round in 1 2; do
  echo "== main text256 $(timeout 300 python tools/exp/text_batch.py text256 256 4 2>&1 | tail -1)"
  echo "== main log256 $(timeout 300 python tools/exp/text_batch.py log256 256 4 2>&1 | tail -1)"
  echo "== main pd $(timeout 300 python tools/exp/pd_batch.py 4 all 2>&1 | tail -1)"
  echo "== main single text $(timeout 300 python tools/exp/single_timing.py text 2>&1 | tail -1)"
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
