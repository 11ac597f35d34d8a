cd /tmp; export TMPDIR=/tmp
for R in 32 64 256; do
  rm -rf /tmp/pr_$R
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pr_$R -o x -- python $GRAFT_REPO_ROOT/bench.py --gib 1 --rows $R --steps 1 --warmup 1 --no-cpu-baseline --no-verify > /tmp/log_$R 2>&1
  grep -o '"decode_GBps": [0-9.]*' /tmp/log_$R
  python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py /tmp/pr_$R/x_results.db | grep -E "walk|imtf|dec_huff|ibwt" | awk -F'|' '{print $2, $3, $4}'
done
