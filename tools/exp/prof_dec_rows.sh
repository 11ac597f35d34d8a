#!/bin/bash
# per-kernel decoder times of one plan for a batch of <rows> blocks (default 1024): bash tools/exp/prof_dec_rows.sh [rows]
ROWS=${1:-1024}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pr
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr -o x -- python $GRAFT_REPO_ROOT/tools/probe_dec.py $ROWS 4 > /tmp/log 2>&1
grep -E "decode batch|round trip" /tmp/log | tail -2
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py /tmp/pr/x_results.db | grep -E "glc::k_(dec|imtf|ibwt|rs_scan)" | awk -F'|' '{printf "%-60s %s %s %s %s\n", substr($2,1,60), $3, $4, $5, $6}'
