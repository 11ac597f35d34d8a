#!/bin/bash
# single call: workgroups per block of k_fs_ties
cd $GRAFT_REPO_ROOT
for g in 24 8 96 384; do
  echo "=== GLC_FST_GRID=$g"
  GLC_FSP2_PER1=1 GLC_FST_GRID=$g bash tools/exp/trace_single.sh 2>&1 | grep -E "k_fs_ties|chain"
done
python - <<'PY'
import importlib.util, os, sys, ctypes as C
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch, numpy as np
import oracle_lib as O
x = bench.zipf_blocks_on_device(torch, torch.device("cuda:0"), 1, 0, 1).cpu().numpy()
# runs of equal 5-symbol prefixes... what k_fs_ties sees: groups of suffixes with equal codes ~ equal first 5-6 symbols
sa = O.suffix_array(x) if hasattr(O, "suffix_array") else None
print("oracle has suffix_array:", sa is not None)
PY
