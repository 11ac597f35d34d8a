#!/bin/bash
# Runs ON the GPU box (through gpurun): full bench line, rocprofv3 kernel stats of the same
# command shape, and the two HBM-traffic PMC passes (each counter in its own run, with
# --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).
#   usage: bash profiles/collect.sh <tag> [bench-gib]
# Outputs land in gpurun_out/<tag>_*; profiles/make_pmc_json.py turns them into the tracked
# summaries under profiles/.
set -u
TAG=${1:-r01}
GIB=${2:-4}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py --gib $GIB --details $OUT/${TAG}_bench_full.json > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp
rm -rf $OUT/${TAG}_prof $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o ${TAG} -- \
    python $REPO/bench.py --gib $GIB --steps 1 --warmup 1 --no-cpu-baseline --main-only --no-overlap-pass > $OUT/${TAG}_prof.log 2>&1
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $CTR --kernel-trace -d $OUT/${TAG}_pmc_$CTR -o pmc -- \
      python $REPO/bench.py --gib 0.25 --rows 256 --steps 1 --warmup 0 --no-cpu-baseline --no-verify --main-only --no-overlap-pass > $OUT/${TAG}_pmc_$CTR.log 2>&1
done
rm -rf $OUT/${TAG}_culzss_prof
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_culzss_prof -o ${TAG} -- \
    python $REPO/tools/bench_culzss.py --gib 1 > $OUT/${TAG}_culzss_bench.json 2> $OUT/${TAG}_culzss_prof.log
cd $REPO
true
ls -la $OUT | tail -20
cat $OUT/${TAG}_bench.json
