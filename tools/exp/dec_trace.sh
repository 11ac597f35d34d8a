#!/bin/bash
# kernel timeline of the decoder's pipelined mode (rocprofv3 --kernel-trace): who runs beside whom, and how long each kernel
# takes there against its time alone.   bash tools/exp/dec_trace.sh [rows]
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/dt; DEC_PROBE_MODES=pipe timeout 300 rocprofv3 --kernel-trace -d /tmp/dt -o t -- python $GRAFT_REPO_ROOT/tools/exp/dec_overlap_probe.py ${1:-1024} 4 > /tmp/dt.log 2>&1
tail -2 /tmp/dt.log
python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/dt/**/*.db", recursive=True)
if not db: print("no db", open("/tmp/dt.log").read()[-400:])
else:
    c = sqlite3.connect(db[0])
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
    qcol = "d.queue_id" if "queue_id" in cols else "0"
    rows = list(c.execute("select s.display_name, d.start, d.end, %s from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start" % qcol))
    dec = [r for r in rows if any(k in r[0] for k in ("k_dec", "k_imtf", "k_ibwt", "k_rs_scan"))]
    # the last two calls: from the second-to-last k_dec_prepare on
    starts = [i for i, r in enumerate(dec) if "k_dec_prepare" in r[0]]
    first = starts[-2] if len(starts) >= 2 else 0
    t0 = dec[first][1]
    for name, s, e, q in dec[first:]:
        print("q%-3s %-40s start %9.1f us  end %9.1f us  dur %8.1f us" % (q, name[:40], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
