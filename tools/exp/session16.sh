cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s16; mkdir -p $O
for v in main k32 main k32; do
  unset GLC_LIB; [ $v != main ] && export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_$v.so
  echo "== $v $(timeout 120 python tools/probe_mtf.py 1024 4 2>&1 | grep k_mtf_encode)"
done > $O/mtf3.log 2>&1; cat $O/mtf3.log
export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_k32.so
timeout 600 python -m pytest tests/test_gpu_cudpp.py tests/test_gpu_refgold.py -x -q -m gpu 2>&1 | tail -1
