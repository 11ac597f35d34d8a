#!/usr/bin/env python3
"""How much do concurrent plans buy?  T host threads, each with its own plan (rows blocks) and stream,
encode disjoint batches of the same 4 GiB Zipf workload.  usage: two_plans.py [threads] [rows] [gib]"""
import importlib.util, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 128
gib = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
pipe = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
dev = torch.device("cuda:0")
n = 1 << 20
nblocks = int(gib * 1024)
d_in = bench.zipf_blocks_on_device(torch, dev, nblocks, 0, 1)
L = glc.lib()
nsub = n // 4096
stride = glc.compressed_stride_words(n)
out = dict(bwt_index=torch.empty(nblocks, dtype=torch.int32, device=dev), hist=torch.empty(nblocks * 256, dtype=torch.int32, device=dev),
           offsets=torch.empty(nblocks * nsub, dtype=torch.int32, device=dev), size=torch.empty(nblocks, dtype=torch.int32, device=dev),
           words=torch.empty(nblocks * stride, dtype=torch.int32, device=dev))
ctx = glc.Cudpp()
plans, streams = [], []
for t in range(T):
    p = glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=rows); s = torch.cuda.Stream(dev)
    p.set_stream(s.cuda_stream)
    if pipe: p.set_pipelining(True)
    plans.append(p); streams.append(s)
batches = list(range(0, nblocks, rows))

def worker(t):
    for b0 in batches[t::T]:
        nb = min(rows, nblocks - b0)
        rc = L.glcCompressBatch(plans[t].handle, d_in.data_ptr() + b0 * n, out["bwt_index"].data_ptr() + 4 * b0,
                                out["hist"].data_ptr() + 1024 * b0, out["offsets"].data_ptr() + 4 * nsub * b0, nsub,
                                out["size"].data_ptr() + 4 * b0, out["words"].data_ptr() + 4 * stride * b0, stride, n, nb)
        assert rc == 0
    plans[t].synchronize()

for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if it: print("threads=%d rows=%d pipelined=%s : %.2f GB/s" % (T, rows, pipe, nblocks * n / (t1 - t0) / 1e9))
d_back = torch.empty_like(d_in)

def dworker(t):
    for b0 in batches[t::T]:
        nb = min(rows, nblocks - b0)
        rc = L.glcDecompressBatch(plans[t].handle, out["bwt_index"].data_ptr() + 4 * b0, out["hist"].data_ptr() + 1024 * b0,
                                  out["offsets"].data_ptr() + 4 * nsub * b0, nsub, out["words"].data_ptr() + 4 * stride * b0,
                                  stride, d_back.data_ptr() + b0 * n, n, nb)
        assert rc == 0
    plans[t].synchronize()

for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=dworker, args=(t,)) for t in range(T)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if it: print("decode threads=%d rows=%d : %.2f GB/s  ok=%s" % (T, rows, nblocks * n / (t1 - t0) / 1e9, bool(torch.equal(d_back, d_in))))
sizes = out["size"].cpu().numpy().astype("int64")
print("ratio %.4f" % (nblocks * n / (sizes.sum() * 4.0)))
