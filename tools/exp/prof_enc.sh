cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pr
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr -o x -- python $GRAFT_REPO_ROOT/tools/probe_bwt.py 256 4 > /tmp/log 2>&1
grep -E "bwt batch" /tmp/log | tail -2
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py /tmp/pr/x_results.db | grep -E "glc::" | awk -F'|' '{printf "%-60s %s %s %s %s\n", substr($2,1,60), $3, $4, $5, $7}'
