cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_fuzz.py tests/test_gpu_compact.py -x -q -m gpu > $O/pytest_dec.log 2>&1; grep -E "passed|failed" $O/pytest_dec.log | tail -2
timeout 300 python tools/exp/dec_overlap_probe.py 1024 4 2>&1 | grep -v amdgpu.ids > $O/dec.log; cat $O/dec.log
