# times the BWT batch with k_fs_sort cut short after phase GLC_FS_STOP (results are wrong on purpose)
for s in 0 1 2 3 4 5 99; do
  echo "stop=$s"; GLC_FS_STOP=$s timeout 120 python tools/probe_bwt.py 256 3 2>&1 | tail -1
done
