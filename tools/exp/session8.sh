cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s8; mkdir -p $O
for v in main fold0; do unset GLC_LIB; [ $v != main ] && export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_$v.so
  echo "== $v"; python tools/exp/single_timing.py zipf 2>&1 | grep -v amdgpu.ids; python tools/exp/single_timing.py text 2>&1 | grep -v amdgpu.ids; done > $O/single_timing.log 2>&1; cat $O/single_timing.log
