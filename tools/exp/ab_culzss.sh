cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/gpu-lossless-compression_amd/libglc_amd.so /tmp/good.so
for V in "$@"; do
  cp $R/gpurun_tmp_$V.so $R/gpu-lossless-compression_amd/libglc_amd.so; touch $R/gpu-lossless-compression_amd/libglc_amd.so
  rm -rf /tmp/pr
  timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pr -o x -- python $R/tools/bench_culzss.py --gib 0.5 > /tmp/log 2>&1
  echo "== $V: $(grep -o '"encode_GBps": [0-9.]*' /tmp/log) $(grep -o '"parity": "[^"]*"' /tmp/log)"
  python $R/profiles/summarize_rocpd.py /tmp/pr/x_results.db | grep -E "lzss_match" | awk -F'|' '{printf "   %-40s calls %s avg %s max %s\n", substr($2,1,40), $3, $5, $7}'
done
cp /tmp/good.so $R/gpu-lossless-compression_amd/libglc_amd.so
