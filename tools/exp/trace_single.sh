#!/bin/bash
# timeline of ONE cudppCompress call with a rows = 1 plan: kernel names, start offsets and durations (rocprofv3 --kernel-trace)
#   bash tools/exp/trace_single.sh [zipf|text]
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ts; timeout 300 rocprofv3 --kernel-trace -d /tmp/ts -o t -- python $GRAFT_REPO_ROOT/tools/exp/probe_single.py ${1:-zipf} > /tmp/ts.log 2>&1
tail -2 /tmp/ts.log
python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/ts/**/*.db", recursive=True)
if not db: print("no db", open("/tmp/ts.log").read()[-400:])
else:
    c = sqlite3.connect(db[0])
    rows = list(c.execute("select s.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
    # the last call = the last k_fs_hist ... k_huff_pack chain
    last = max(i for i, r in enumerate(rows) if "k_fs_hist" in r[0])
    # memsets ahead of k_fs_hist belong to the call
    first = last
    while first > 0 and ("fillBuffer" in rows[first - 1][0] or "memset" in rows[first - 1][0].lower()): first -= 1
    t0 = rows[first][1]
    prev_end = t0
    for name, s, e in rows[first:]:
        print("%-46s start %8.1f us  dur %7.1f us  gap %6.1f us" % (name[:46], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        prev_end = e
    print("chain: %.1f us from first start to last end" % ((rows[-1][2] - t0) / 1e3))
PY
