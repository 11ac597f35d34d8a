#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_resume.py tests/test_gpu_sample_sorter.py tests/test_gpu_bench_inputs.py -m gpu -x -q 2>&1 | tail -2
for round in 1 2; do
  echo "== text256 $(timeout 300 python tools/exp/text_batch.py text256 256 4 2>&1 | tail -1)"
  echo "== log256 $(timeout 300 python tools/exp/text_batch.py log256 256 4 2>&1 | tail -1)"
  echo "== pd $(timeout 300 python tools/exp/pd_batch.py 4 all 2>&1 | tail -1)"
done
echo "== kstats pd"; bash tools/exp/kstats.sh python $GRAFT_REPO_ROOT/tools/exp/pd_batch.py 3 all 2>&1 | grep "k_ss_sample"
timeout 900 python tools/exp/resume_stress.py 30 2718 2>&1 | tail -1
