#!/bin/bash
# sampled symbol statistics with the one-count floor: the suite, then value with step 4 / step 1 alternating
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for i in 1 2; do
for s in 4 1; do
  echo "== GLC_FSH_STEP=$s"
  GLC_FSH_STEP=$s timeout 600 python bench.py --steps 4 --warmup 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'])"
done; done
