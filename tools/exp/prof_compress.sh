cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pr
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr -o x -- python $GRAFT_REPO_ROOT/bench.py --gib 1 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/log 2>&1
grep -o '"value": [0-9.]*\|"decode_GBps": [0-9.]*' /tmp/log
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py /tmp/pr/x_results.db | grep -E "glc::" | awk -F'|' '{printf "%-58s %s %s %s %s\n", substr($2,1,58), $3, $4, $5, $7}'
