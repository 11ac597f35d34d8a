# LDS health of every kernel of the bench run: unaligned stalls, bank conflicts (one rocprofv3 --pmc pass, --kernel-trace only)
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm; timeout 400 rocprofv3 --pmc SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/bench.py --main-only --no-overlap-pass --steps 2 --warmup 1 > /tmp/l.txt 2>&1
python - <<PY
import sqlite3,glob,collections
db=glob.glob("/tmp/pm/**/*.db",recursive=True)
if not db: print("no db", open("/tmp/l.txt").read()[-400:])
else:
    c=sqlite3.connect(db[0])
    q="select s.display_name, i.name, sum(e.value)/count(distinct d.id), count(distinct d.id), avg(d.end-d.start) from rocpd_pmc_event e join rocpd_kernel_dispatch d on e.event_id=d.event_id join rocpd_info_kernel_symbol s on d.kernel_id=s.id join rocpd_info_pmc i on e.pmc_id=i.id group by 1,2"
    t=collections.defaultdict(dict)
    for r in c.execute(q): t[r[0]][r[1]]=r[2]; t[r[0]]["us"]=r[4]/1e3; t[r[0]]["n"]=r[3]
    for k,v in sorted(t.items(), key=lambda kv:-kv[1]["us"]*kv[1]["n"]):
        if v["us"]*v["n"]<50: continue
        print("%-44s %8.1f us x%-4d insts %.2e active %.2e idx %.2e bank %.2e addr %.2e unaligned %.2e" % (k[:44], v["us"], v["n"], v.get("SQ_INSTS_LDS",0), v.get("SQ_ACTIVE_INST_LDS",0), v.get("SQ_LDS_IDX_ACTIVE",0), v.get("SQ_LDS_BANK_CONFLICT",0), v.get("SQ_LDS_ADDR_CONFLICT",0), v.get("SQ_LDS_UNALIGNED_STALL",0)))
PY
