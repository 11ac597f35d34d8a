cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_periodic.py -x -q -m gpu > $O/pytest_per.log 2>&1; tail -25 $O/pytest_per.log
timeout 300 python tools/exp/deep_kinds.py 16 2>&1 | grep -v amdgpu.ids > $O/deep_kinds.log; cat $O/deep_kinds.log
