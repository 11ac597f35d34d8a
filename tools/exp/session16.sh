cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cudpp.py tests/test_gpu_refgold.py tests/test_gpu_fuzz.py tests/test_gpu_bench_inputs.py tests/test_gpu_huffman_ties.py tests/test_c_caller.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in mtfold main mtfold main; do unset GLC_LIB; [ $v != main ] && export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_$v.so
  echo "== $v"; python tools/probe_mtf.py 1024 4 2>&1 | grep k_mtf_encode
  python bench.py --gib 4 --steps 6 --main-only --no-cpu-baseline --no-verify --details /tmp/d.json 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); print(j['value'], j.get('value_no_stage_overlap_GBps'), j['kernel_ms_per_launch'])"
done > $O/mtf.log 2>&1; cat $O/mtf.log
