# usage: ab_bwt.sh tag...   -- glcBwtBatch time (256 Zipf blocks) with gpu-lossless-compression_amd/variants/libglc_<tag>.so ("main" = the in-tree library)
for v in "$@"; do
  unset GLC_LIB
  if [ $v != main ]; then export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_$v.so; fi
  echo "$v: $(python tools/probe_bwt.py 256 4 2>&1 | tail -2 | tr '\n' ' ')"
done
