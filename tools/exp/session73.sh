#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_resume.py tests/test_gpu_sample_sorter.py -m gpu -x -q 2>&1 | tail -2
for round in 1 2; do
  echo "== pd $(timeout 300 python tools/exp/pd_batch.py 4 all 2>&1 | tail -1)"
done
echo "== kstats pd"; bash tools/exp/kstats.sh python $GRAFT_REPO_ROOT/tools/exp/pd_batch.py 3 all 2>&1 | grep "glc::" | head -4
timeout 900 python tools/exp/resume_stress.py 30 31337 2>&1 | tail -1
