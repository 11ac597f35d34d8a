cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s3; mkdir -p $O
for v in ep0 ep3; do
  export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_$v.so
  echo "== $v"; GLC_FS_STOP_AFTER_PART=1 timeout 120 python tools/exp/part_probe.py 1024 4 2>&1 | grep -v amdgpu.ids
done > $O/part.log 2>&1; cat $O/part.log
for v in main es1 es3 es4; do
  unset GLC_LIB; [ $v != main ] && export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_$v.so
  echo "== $v"; timeout 120 python tools/exp/part_probe.py 1024 4 2>&1 | grep -v amdgpu.ids
done > $O/sort.log 2>&1; cat $O/sort.log
for v in main em1 em2 em3 em4 em5; do
  unset GLC_LIB; [ $v != main ] && export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_$v.so
  echo "== $v"; timeout 120 python tools/probe_mtf.py 1024 4 2>&1 | grep -v amdgpu.ids
done > $O/mtf.log 2>&1; cat $O/mtf.log
