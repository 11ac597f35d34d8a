# SQ counters of the bucket-sorter kernels (one rocprofv3 --pmc pass per counter group, --kernel-trace only)
cd /tmp; export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_ATOMIC_RETURN SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pm; timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/probe_bwt.py 256 2 > /tmp/l.txt 2>&1
  python - <<PY
import sqlite3,glob
db=glob.glob("/tmp/pm/**/*.db",recursive=True)
if not db: print("no db", open("/tmp/l.txt").read()[-300:])
else:
    c=sqlite3.connect(db[0])
    q="select s.display_name, i.name, sum(e.value)/count(distinct d.id), count(distinct d.id) from rocpd_pmc_event e join rocpd_kernel_dispatch d on e.event_id=d.event_id join rocpd_info_kernel_symbol s on d.kernel_id=s.id join rocpd_info_pmc i on e.pmc_id=i.id where s.display_name like '%k_fs_sort%' or s.display_name like '%k_fs_part%' group by 1,2"
    for r in c.execute(q): print(r[0][5:14], r[1], "%.4e per launch"%r[2], r[3])
PY
done
