cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s2; mkdir -p $O
for v in ep0 ep1 ep2 ep4 ep4w; do
  export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_$v.so
  echo "== $v"; GLC_FS_STOP_AFTER_PART=1 timeout 120 python tools/exp/part_probe.py 1024 4 2>&1 | grep -v amdgpu.ids
done > $O/part.log 2>&1; cat $O/part.log
unset GLC_LIB
bash tools/exp/dec_trace.sh 1024 > $O/dec_trace.log 2>&1; cat $O/dec_trace.log
