#!/bin/bash
# The encode's timed region kernel by kernel under rocprofv3 --kernel-trace: the last dispatches of bench.py's main leg as
# (name, queue, start, end) so that what runs beside what can be read off.   bash tools/exp/timeline.sh [steps] [extra bench flags]
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
STEPS=${1:-2}; shift
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 900 rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 1 --main-only --no-cpu-baseline --no-verify "$@" > $OUT/timeline.log 2>&1
python - <<PY
import sqlite3, glob, json
db = glob.glob("/tmp/tl/**/*.db", recursive=True)
c = sqlite3.connect(db[0])
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
print(cols)
qcol = "queue_id" if "queue_id" in cols else None
scol = "stream_id" if "stream_id" in cols else None
sel = "s.display_name, d.start, d.end" + (", d.%s" % qcol if qcol else ", 0") + (", d.%s" % scol if scol else ", 0")
rows = list(c.execute("select %s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start" % sel))
print(len(rows), "dispatches")
t0 = rows[0][1]
out = [[r[0][:40], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, r[3], r[4]] for r in rows]
json.dump(out, open("$OUT/timeline.json", "w"))
PY
tail -3 $OUT/timeline.log
ls -la $OUT/timeline.json
