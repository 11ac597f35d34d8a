# where the decoder kernels' cycles go: instruction counts, busy cycles, LDS activity (rocprofv3 --pmc passes, --kernel-trace only)
# usage: pmc_dec.sh [kernel-name-substring]   (default k_imtf_pos)
cd /tmp; export TMPDIR=/tmp
K=${1:-k_imtf_pos}
run() {
  rm -rf /tmp/pm; timeout 200 rocprofv3 --pmc $1 --kernel-trace -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/probe_dec.py 1024 1 > /tmp/l.txt 2>&1
  python - <<PY
import sqlite3,glob
db=glob.glob("/tmp/pm/**/*.db",recursive=True)
if not db: print("no db", open("/tmp/l.txt").read()[-400:])
else:
    c=sqlite3.connect(db[0])
    q="select s.display_name, i.name, sum(e.value)/count(distinct d.id), count(distinct d.id), avg(d.end-d.start) from rocpd_pmc_event e join rocpd_kernel_dispatch d on e.event_id=d.event_id join rocpd_info_kernel_symbol s on d.kernel_id=s.id join rocpd_info_pmc i on e.pmc_id=i.id where s.display_name like '%$K%' group by 1,2"
    for r in c.execute(q): print("%-26s %.4e per launch = %8.1f per 64 symbols   (kernel avg %.1f us under the counters)" % (r[1], r[2], r[2]/ (1073741824/64.0), r[4]/1e3))
PY
}
run "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
run "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"
run "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE"
