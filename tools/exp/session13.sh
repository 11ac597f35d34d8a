cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s13; mkdir -p $O
for v in t4 t4c t4d t8b t8c t8d; do
  export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_$v.so
  echo "== $v"; GLC_FS_STOP_AFTER_PART=1 timeout 120 python tools/exp/part_probe.py 1024 4 2>&1 | grep k_fs_part
  unset GLC_FS_STOP_AFTER_PART; timeout 120 python tools/exp/part_probe.py 1024 4 2>&1 | grep k_fs_
  timeout 300 python -m pytest tests/test_gpu_bucket_sorter.py tests/test_gpu_refgold.py -x -q -m gpu 2>&1 | tail -1
done > $O/part4.log 2>&1; cat $O/part4.log
