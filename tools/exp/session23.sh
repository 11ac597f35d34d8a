#!/bin/bash
# the dissection of k_fs_part2 and k_mtf_encode at the round's last code state -> gpurun_out/r05_dissect2.log
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/gpu-lossless-compression_amd/variants
LOG=gpurun_out/r05_dissect2.log
: > $LOG
for t in p2_0 p2_1 p2_2 p2_4 p2_16 p2_ns; do
  echo "== $t" >> $LOG
  GLC_FS_STOP_AFTER_PART=1 GLC_LIB=$V/libglc_$t.so python tools/exp/part_probe.py 1024 4 2>/dev/null | grep -E "k_fs_part|k_fs_hist" >> $LOG
done
for t in main m1 m2 m3 m4 m5; do
  echo "== $t" >> $LOG
  if [ $t = main ]; then unset GLC_LIB; else export GLC_LIB=$V/libglc_$t.so; fi
  python tools/probe_mtf.py 1024 5 2>/dev/null | grep -E "k_mtf" >> $LOG
done
cat $LOG
