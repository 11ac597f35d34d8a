cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s17; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed|Error|assert" $O/pytest.log | tail -5
for k in 1 2; do for st in 1 4; do
  GLC_FSH_STEP=$st python bench.py --gib 4 --steps 6 --main-only --no-cpu-baseline --no-verify --details /tmp/d.json 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); print('step $st', j['value'], j.get('value_no_stage_overlap_GBps'), j['kernel_ms_per_launch'])"
done; done > $O/value.log 2>&1; cat $O/value.log
