# usage: ab_dec.sh tag1 tag2 ... -- decode timing + per-kernel stats for tools/exp/libglc_<tag>.so on one box
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in "$@"; do
  export GLC_LIB=$R/tools/exp/libglc_$V.so
  rm -rf /tmp/pr
  timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pr -o x -- python $R/tools/probe_dec.py 1024 4 --check > /tmp/log 2>&1
  echo "== $V: $(grep -E 'decode batch' /tmp/log | tail -2 | tr '\n' ' ') $(grep 'round trip' /tmp/log)"
  python $R/profiles/summarize_rocpd.py /tmp/pr/x_results.db | grep -E "glc::k_(dec|imtf|ibwt)" | awk -F'|' '{printf "   %-50s %s %s %s\n", substr($2,1,50), $3, $4, $5}'
done
