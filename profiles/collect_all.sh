#!/bin/bash
# Everything the round's evidence is made of, at ONE code state under ONE tag (runs ON the GPU box through gpurun):
#   bash profiles/collect_all.sh <tag>
#   1. profiles/collect.sh <tag> 4     bench line + details, rocprofv3 --kernel-trace --stats of the headline + CULZSS, the two HBM-traffic PMC passes
#   2. tools/exp/pmc_insts.sh <tag>    the three instruction-counter PMC passes (--kernel-trace only)
#   3. rocprofv3 --kernel-trace --stats of 256 DISTINCT text / log blocks (tools/exp/text_batch.py) and of the partly_deep batch
#   4. the C ring bench of the CULZSS host-pointer ABI, three times
#   5. profiles/finish_all.py <tag> ON the box (the databases are too big to travel); the summaries come back in
#      gpurun_out/<tag>_profiles/ -> cp gpurun_out/<tag>_profiles/* profiles/
set -u
TAG=${1:-r05z}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
bash profiles/collect.sh $TAG 4 > $OUT/${TAG}_collect.log 2>&1
bash tools/exp/pmc_insts.sh $TAG >> $OUT/${TAG}_collect.log 2>&1
cd /tmp
for W in text256 log256; do
  rm -rf $OUT/${TAG}_${W}_prof
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_${W}_prof -o $TAG -- python $REPO/tools/exp/text_batch.py $W 256 3 > $OUT/${TAG}_${W}.log 2>&1
done
rm -rf $OUT/${TAG}_pd_prof
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_pd_prof -o $TAG -- python $REPO/tools/exp/pd_batch.py 3 all > $OUT/${TAG}_pd.log 2>&1
cd $REPO
gcc -O2 -o /tmp/ringb tests/c_caller/culzss_ring_bench.c -Iinclude -L gpu-lossless-compression_amd -lglc_amd -lpthread -Wl,-rpath,$REPO/gpu-lossless-compression_amd 2> $OUT/${TAG}_ring.log
for i in 1 2 3; do timeout 120 /tmp/ringb 256 16 >> $OUT/${TAG}_ring.log 2>&1; done
# the databases stay on the box (gpurun brings back 64 MiB): summarised here, the summaries travel
python profiles/finish_all.py $TAG > $OUT/${TAG}_finish.log 2>&1
mkdir -p $OUT/${TAG}_profiles
cp profiles/${TAG}_* profiles/pmc_traffic.json profiles/pmc_insts.json profiles/isa_census.json $OUT/${TAG}_profiles/
rm -rf $OUT/${TAG}_prof $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_culzss_prof $OUT/${TAG}_text256_prof $OUT/${TAG}_log256_prof $OUT/${TAG}_pd_prof
du -sh $OUT; tail -5 $OUT/${TAG}_finish.log; ls $OUT/${TAG}_profiles
tail -3 $OUT/${TAG}_ring.log
cat $OUT/${TAG}_bench.json
