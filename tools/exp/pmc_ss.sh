# instruction mix of the sample sorter's kernels on text blocks (rocprofv3 --pmc passes, --kernel-trace only)
cd /tmp; export TMPDIR=/tmp
run() {
  rm -rf /tmp/pm; timeout 200 rocprofv3 --pmc $1 --kernel-trace -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/probe_data.py text > /tmp/l.txt 2>&1
  python - <<PY
import sqlite3,glob
db=glob.glob("/tmp/pm/**/*.db",recursive=True)
if not db: print("no db", open("/tmp/l.txt").read()[-400:])
else:
    c=sqlite3.connect(db[0])
    q="select s.display_name, i.name, sum(e.value)/count(distinct d.id), count(distinct d.id), avg(d.end-d.start) from rocpd_pmc_event e join rocpd_kernel_dispatch d on e.event_id=d.event_id join rocpd_info_kernel_symbol s on d.kernel_id=s.id join rocpd_info_pmc i on e.pmc_id=i.id where s.display_name like '%k_ss_%' or s.display_name like '%k_fs_part<true>%' group by 1,2"
    for r in c.execute(q): print("%-22s %-22s %.4e per launch = %8.1f per 64 suffixes   (kernel avg %.1f us under the counters)" % (r[0][5:27], r[1], r[2], r[2]/ (268435456/64.0), r[4]/1e3))
PY
}
run "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES"
run "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"
