cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s11; mkdir -p $O
for v in main tol8 main tol8; do unset GLC_LIB; [ $v != main ] && export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_$v.so
  echo "== $v"; timeout 300 python tools/exp/deep_probe.py 64 "" 0 2>&1 | grep -v amdgpu.ids | cut -c1-120; done > $O/deep.log 2>&1; cat $O/deep.log
unset GLC_LIB
timeout 600 python -m pytest tests/test_gpu_resume.py tests/test_gpu_sample_sorter.py tests/test_gpu_periodic.py -x -q -m gpu 2>&1 | tail -2
