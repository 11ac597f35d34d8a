#!/bin/bash
# decoder A/B on one box: bash tools/exp/ab_dec2.sh main tag...   (variants/libglc_<tag>.so; probe_dec.py 1024 rows, per-kernel stats)
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
for V in "$@"; do
  if [ $V = main ]; then unset GLC_LIB; else export GLC_LIB=$R/gpu-lossless-compression_amd/variants/libglc_$V.so; fi
  rm -rf /tmp/pr; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pr -o x -- python $R/tools/probe_dec.py 1024 4 --check > /tmp/log 2>&1
  echo "== $V: $(grep -E 'decode batch' /tmp/log | tail -2 | tr '\n' ' ') $(grep 'round trip' /tmp/log)"
  python $R/profiles/summarize_rocpd.py /tmp/pr/x_results.db | grep -E "glc::k_(dec|imtf|ibwt)" | awk -F'|' '{printf "   %-46s %s\n", substr($2,1,46), $5}'
done
