cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s14; mkdir -p $O
export GLC_FSP2_SHAPE=0
for k in 1 2 3; do for per in 4 8 12 16 24 32 64; do
  GLC_FSP2_PER=$per python bench.py --gib 4 --steps 6 --main-only --no-cpu-baseline --no-verify --details /tmp/d.json > /tmp/l.json 2>/dev/null
  python - <<PY
import json
j=json.load(open("/tmp/l.json")); k=j["kernel_ms_per_launch"]
print("per $per value", j["value"], "no-overlap", j.get("value_no_stage_overlap_GBps"), "part", k["k_fs_part"], "sort", k["k_fs_sort"])
PY
done; done > $O/value4.log 2>&1; sort -k2 -n -s $O/value4.log
