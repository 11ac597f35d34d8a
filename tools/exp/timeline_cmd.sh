#!/bin/bash
# kernel timeline (name, start us, dur us, gap to the previous kernel's end) of the LAST <count> dispatches of any command:
#   bash tools/exp/timeline_cmd.sh <count> <command...>
N=$1; shift
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tlc; timeout 600 rocprofv3 --kernel-trace -d /tmp/tlc -o t -- "$@" > /tmp/tlc.log 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/tlc/**/*.db", recursive=True)
c = sqlite3.connect(db[0])
rows = list(c.execute("select s.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
rows = rows[-$N:]
t0 = rows[0][1]; prev = rows[0][1]
for nm, a, b in rows:
    nm = nm.replace("void ", "").replace("glc::", "")[:44]
    print("%-44s start %9.1f  dur %8.1f  gap %7.1f" % (nm, (a - t0) / 1e3, (b - a) / 1e3, (a - prev) / 1e3))
    prev = max(prev, b)
print("span %.1f us" % ((prev - t0) / 1e3))
PY
