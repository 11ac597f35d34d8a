#!/bin/bash
# Per-kernel instruction counters of the encode / decode pipelines (runs ON the GPU box through gpurun):
#   bash tools/exp/pmc_insts.sh <tag>
# Three rocprofv3 --pmc passes with --kernel-trace only (counters never share a run with other trace domains), over
#   python bench.py --gib 0.25 --rows 256 --steps 1 --warmup 0 --no-cpu-baseline --no-verify --main-only --no-overlap-pass
# -> gpurun_out/<tag>_pmc_insts.json: for every kernel the counter sums per launch, the launch count and the average
# duration under the counters.  profiles/make_pmc_insts.py <tag> turns it into the tracked summary bench.py reads.
set -u
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $REPO/bench.py --gib 0.25 --rows 256 --steps 1 --warmup 0 --no-cpu-baseline --no-verify --main-only --no-overlap-pass"
i=0
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmi_$i
  timeout 300 rocprofv3 --pmc $SET --kernel-trace -d /tmp/pmi_$i -o p -- $CMD > /tmp/pmi_$i.log 2>&1
  i=$((i+1))
done
python - "$OUT/${TAG}_pmc_insts.json" "$CMD" <<'PY'
import glob, json, sqlite3, sys
res = {"command": "rocprofv3 --pmc <set> --kernel-trace -- " + sys.argv[2], "passes": [], "kernels": {}}
for i in range(3):
    db = glob.glob("/tmp/pmi_%d/**/*.db" % i, recursive=True)
    if not db:
        res["passes"].append({"pass": i, "error": open("/tmp/pmi_%d.log" % i).read()[-400:]})
        continue
    c = sqlite3.connect(db[0])
    q = ("select s.display_name, i.name, sum(e.value), count(distinct d.id), avg(d.end - d.start) "
         "from rocpd_pmc_event e join rocpd_kernel_dispatch d on e.event_id = d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id join rocpd_info_pmc i on e.pmc_id = i.id group by 1, 2")
    names = set()
    for name, ctr, total, launches, avg_ns in c.execute(q):
        k = res["kernels"].setdefault(name[:120], {"launches": launches, "avg_us_under_counters": {}, "per_launch": {}})
        k["per_launch"][ctr] = total / launches
        k["avg_us_under_counters"]["pass%d" % i] = avg_ns / 1e3
        names.add(ctr)
    res["passes"].append({"pass": i, "counters": sorted(names)})
json.dump(res, open(sys.argv[1], "w"), indent=1)
for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["per_launch"].get("SQ_INSTS_VALU", 0))[:14]:
    print("%-50s VALU %.3e SALU %.3e LDS %.3e" % (k[:50], v["per_launch"].get("SQ_INSTS_VALU", 0), v["per_launch"].get("SQ_INSTS_SALU", 0), v["per_launch"].get("SQ_INSTS_LDS", 0)))
PY
