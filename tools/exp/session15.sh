cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s15; mkdir -p $O
for k in 1 2 3; do for pr in least same greatest; do
  GLC_SIDE_PRIO=$pr python bench.py --gib 4 --steps 6 --main-only --no-cpu-baseline --no-verify --details /tmp/d.json 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); print('$pr', j['value'], j.get('value_no_stage_overlap_GBps'))"
done; done > $O/prio.log 2>&1; sort -s -k1,1 $O/prio.log
