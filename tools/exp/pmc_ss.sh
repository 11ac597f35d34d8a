# instruction mix of the sample sorter's kernels on 256 distinct text (or log) blocks (rocprofv3 --pmc passes, --kernel-trace only)
# usage: pmc_ss.sh [text256|log256]
kind=${1:-text256}
cd /tmp; export TMPDIR=/tmp
run() {
  rm -rf /tmp/pm; timeout 300 rocprofv3 --pmc $1 --kernel-trace -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/tools/exp/text_batch.py $kind 256 1 > /tmp/l.txt 2>&1
  python - <<PY
import sqlite3,glob
db=glob.glob("/tmp/pm/**/*.db",recursive=True)
if not db: print("no db", open("/tmp/l.txt").read()[-400:])
else:
    c=sqlite3.connect(db[0])
    q="select s.display_name, i.name, sum(e.value), count(distinct d.id), sum(d.end-d.start) from rocpd_pmc_event e join rocpd_kernel_dispatch d on e.event_id=d.event_id join rocpd_info_kernel_symbol s on d.kernel_id=s.id join rocpd_info_pmc i on e.pmc_id=i.id where s.display_name like '%k_ss_%' or s.display_name like '%k_fs_part<true>%' group by 1,2"
    for r in c.execute(q): print("%-24s %-22s %.4e per batch (2 calls' launches / 2) = %8.1f per 64 suffixes   (%d launches, %.1f us per batch under the counters)" % (r[0].replace("void ","").replace("glc::","")[:24], r[1], r[2]/2, r[2]/2/(268435456/64.0), r[3], r[4]/2e3))
PY
}
run "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES"
run "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES"
run "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_WAVES"
