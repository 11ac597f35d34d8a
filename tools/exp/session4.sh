cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s4; mkdir -p $O
timeout 300 python tools/exp/deep_kinds.py 16 2>&1 | grep -v amdgpu.ids > $O/deep_kinds.log; cat $O/deep_kinds.log
for k in page4k onebyte period2 phrase2000; do echo "== $k"; bash tools/exp/kstats.sh python $GRAFT_REPO_ROOT/tools/exp/deep_kinds.py 16 $k 2>&1 | head -22; done > $O/deep_kstats.log 2>&1; cat $O/deep_kstats.log
