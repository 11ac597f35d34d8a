cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/gpu-lossless-compression_amd/libglc_amd.so /tmp/good.so
for V in "$@"; do
  cp $R/gpurun_tmp_$V.so $R/gpu-lossless-compression_amd/libglc_amd.so
  touch $R/gpu-lossless-compression_amd/libglc_amd.so
  echo "== $V"
  timeout 60 python $R/tools/probe_bwt.py 256 1 2>&1 | grep -E "lb\]|bwt batch" | tail -6
done
cp /tmp/good.so $R/gpu-lossless-compression_amd/libglc_amd.so
