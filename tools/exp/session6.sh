cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
gcc -O2 -std=gnu99 -Wall -pthread -I include -I /opt/rocm/include tests/c_caller/culzss_ring_bench.c -o /tmp/ring_bench -L gpu-lossless-compression_amd -lglc_amd -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/gpu-lossless-compression_amd -Wl,-rpath,/opt/rocm/lib
bash tools/exp/lz_ring_trace.sh > $O/lz_trace.log 2>&1; cat $O/lz_trace.log
