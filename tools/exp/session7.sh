cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s7; mkdir -p $O
gcc -O2 -std=gnu99 -Wall -pthread -I include -I /opt/rocm/include tests/c_caller/culzss_ring_bench.c -o /tmp/ring_bench -L gpu-lossless-compression_amd -lglc_amd -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/gpu-lossless-compression_amd -Wl,-rpath,/opt/rocm/lib
for n in 16 32 64 128 256 16 256; do /tmp/ring_bench $n 16; done > $O/ring_len.log 2>&1; cat $O/ring_len.log
timeout 600 python -m pytest tests/test_c_caller.py tests/test_gpu_culzss.py tests/test_gpu_lzss_refgold.py -x -q -m gpu > $O/pytest_lz.log 2>&1; tail -3 $O/pytest_lz.log
