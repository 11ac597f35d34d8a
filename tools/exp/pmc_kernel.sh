# usage: pmc_kernel.sh <kernel-name-pattern> <counter>...   -- one rocprofv3 --pmc pass (--kernel-trace only) over tools/probe_bwt.py 256 2;
# prints each counter per launch and per 64 suffixes for the kernels whose name contains the pattern
pat=$1; shift
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/probe_bwt.py 256 2 > /tmp/l.txt 2>&1
python - "$pat" <<PY
import sqlite3,glob,sys
db=glob.glob("/tmp/pm/**/*.db",recursive=True)
if not db: print("no db", open("/tmp/l.txt").read()[-600:])
else:
    c=sqlite3.connect(db[0])
    q="select s.display_name, i.name, sum(e.value)/count(distinct d.id), count(distinct d.id), avg(d.end-d.start) from rocpd_pmc_event e join rocpd_kernel_dispatch d on e.event_id=d.event_id join rocpd_info_kernel_symbol s on d.kernel_id=s.id join rocpd_info_pmc i on e.pmc_id=i.id where s.display_name like '%"+sys.argv[1]+"%' group by 1,2"
    for r in c.execute(q): print(r[0][:28], "%-24s %.4e per launch = %8.2f per 64 suffixes if a launch is 256 blocks; %d launches, sum %.4e; kernel avg %.1f us, sum %.1f us (under the counters)" % (r[1], r[2], r[2]/ (268435456/64.0), r[3], r[2]*r[3], r[4]/1e3, r[4]*r[3]/1e3))
PY
