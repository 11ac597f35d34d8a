#!/usr/bin/env python3
"""The four kinds of block of bench.py's `deep_repeats` leg, one kind per batch: time per block of glcBwtBatch, which tier
finished them, BWT against the oracle (first block).  usage: deep_kinds.py [copies] [kind substring]   (under kstats.sh:
per-kernel totals)"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import numpy as np, torch, datagen
import oracle_lib as O
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 16
only = sys.argv[2] if len(sys.argv) > 2 else None
N = 1 << 20
rng = np.random.default_rng(7)
onebyte = np.full(N, 65, dtype=np.uint8); onebyte[-1] = 66
t = datagen.text_bytes(N, seed=12).copy()
for o in range(0, N - 2000, 16384):
    t[o:o + 2000] = t[:2000]
KINDS = {"page4k": np.tile(rng.integers(0, 256, 4096, dtype=np.uint8), N // 4096), "onebyte": onebyte,
         "period2": np.tile(np.frombuffer(b"ab", dtype=np.uint8), N // 2), "phrase2000": t}
dev = torch.device("cuda:0")
L = glc.lib()
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_BWT, N, rows=copies) as plan:
    for name, x in KINDS.items():
        if only and only not in name: continue
        d_in = torch.from_numpy(np.tile(x, copies)).to(dev)
        d_out = torch.zeros_like(d_in); d_idx = torch.zeros(copies, dtype=torch.int32, device=dev)
        ts = []
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            assert L.glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), N, copies) == 0
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        want, widx = O.bwt(x)
        got = d_out[:N].cpu().numpy()
        ok = np.array_equal(got, want) and int(d_idx[0].item()) == widx and bool(torch.equal(d_out.view(copies, N)[-1], d_out.view(copies, N)[0]))
        print("%-12s %8.2f ms per batch of %d = %.3f ms per block (%.2f GB/s)  %s  (flagged, given up, resumed) = %r" % (
            name, min(ts), copies, min(ts) / copies, copies * N / min(ts) / 1e6, "ok" if ok else "WRONG",
            plan.last_sort_stats() + (plan.last_sort_resumed(),)), flush=True)
