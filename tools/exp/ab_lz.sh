# usage: ab_lz.sh tag1 tag2 ... -- CULZSS encode timing + match/pack kernel stats for tools/exp/libglc_<tag>.so on one box
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in "$@"; do
  export GLC_LIB=$R/tools/exp/libglc_$V.so
  rm -rf /tmp/pr
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pr -o x -- python $R/tools/bench_culzss.py --gib 1 > /tmp/log 2>&1
  echo "== $V: $(grep -o '"encode_GBps": [0-9.]*' /tmp/log | head -1) $(grep -o '"parity": "[^"]*"' /tmp/log) $(grep -o '"roundtrip": "[^"]*"' /tmp/log)"
  python $R/profiles/summarize_rocpd.py /tmp/pr/x_results.db | grep -E "lzss_(match|pack|gather|layout)" | awk -F'|' '{printf "   %-40s calls %s avg %s max %s\n", substr($2,1,40), $3, $5, $7}'
done
