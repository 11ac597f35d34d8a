# round 5, first GPU session: parity after the work-list / MTF / k_ss_long changes, then A/B of the occupancy variants
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
bash tools/exp/ab_value.sh main p1024 w6 > $O/ab_value.log 2>&1; cat $O/ab_value.log
for v in main pad86; do for pr in least same; do
  unset GLC_LIB; [ $v != main ] && export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_$v.so
  echo "== $v side priority $pr"; GLC_SIDE_PRIO=$pr timeout 300 python tools/exp/dec_overlap_probe.py 1024 4
done; done > $O/dec_overlap.log 2>&1; cat $O/dec_overlap.log
