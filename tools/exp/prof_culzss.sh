cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pr
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr -o x -- python $GRAFT_REPO_ROOT/tools/bench_culzss.py --gib 0.5 > /tmp/log 2>&1
grep -o '"encode_GBps": [0-9.]*\|"decode_GBps": [0-9.]*\|"parity": "[^"]*"\|"roundtrip": "[^"]*"' /tmp/log
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py /tmp/pr/x_results.db | grep -E "glc::" | awk -F'|' '{printf "%-50s %s %s %s %s\n", substr($2,1,50), $3, $4, $5, $7}'
