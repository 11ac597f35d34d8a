cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/gpu-lossless-compression_amd/libglc_amd.so /tmp/good.so
for V in GOOD "$@"; do
  if [ $V = GOOD ]; then cp /tmp/good.so $R/gpu-lossless-compression_amd/libglc_amd.so; else cp $R/gpurun_tmp_$V.so $R/gpu-lossless-compression_amd/libglc_amd.so; fi
  touch $R/gpu-lossless-compression_amd/libglc_amd.so
  rm -rf /tmp/pr
  timeout 60 rocprofv3 --kernel-trace --stats -d /tmp/pr -o x -- python $R/tools/probe_bwt.py 256 2 > /tmp/log 2>&1
  echo "== $V: $(grep -E 'bwt batch' /tmp/log | tail -1)"
  python $R/profiles/summarize_rocpd.py /tmp/pr/x_results.db | grep -E "onesweep|rank1" | awk -F'|' '{printf "   %-40s calls %s avg %s max %s\n", substr($2,1,40), $3, $5, $7}'
done
cp /tmp/good.so $R/gpu-lossless-compression_amd/libglc_amd.so
