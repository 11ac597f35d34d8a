cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s12; mkdir -p $O
for v in main r4 r8 r16 r8w7; do
  unset GLC_LIB; [ $v != main ] && export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_$v.so
  echo "== $v"; timeout 120 python tools/exp/part_probe.py 1024 4 2>&1 | grep -v amdgpu.ids
  timeout 300 python -m pytest tests/test_gpu_bucket_sorter.py tests/test_gpu_refgold.py -x -q -m gpu 2>&1 | tail -1
done > $O/pipe.log 2>&1; cat $O/pipe.log
