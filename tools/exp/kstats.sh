# per-kernel totals of a command under rocprofv3 --kernel-trace:  bash tools/exp/kstats.sh <command...>
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks; timeout 600 rocprofv3 --kernel-trace -d /tmp/ks -o k -- "$@" > /tmp/ks.log 2>&1
python - <<PY
import sqlite3,glob
db=glob.glob("/tmp/ks/**/*.db",recursive=True)
if not db: print("no db", open("/tmp/ks.log").read()[-400:])
else:
    c=sqlite3.connect(db[0])
    q="select s.display_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by 1 order by 3 desc"
    tot=0
    rows=list(c.execute(q))
    for r in rows: tot+=r[2]
    print("total kernel time %.2f ms" % (tot/1e6))
    for r in rows[:30]: print("%-60s x%-6d total %9.3f ms  avg %9.1f us" % (r[0][:60], r[1], r[2]/1e6, r[3]/1e3))
PY
