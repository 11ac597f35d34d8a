#!/usr/bin/env python3
"""Blocks with repeats deeper than the sample sorter's cap: BWT against the oracle, which tier finished them, and the time of
a batch with the doubling rounds resumed from the sample sorter's order (sorter 0) and from scratch (sorter 5).
usage: deep_probe.py [copies] [case substring] [modes, e.g. 0,5]"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import numpy as np, torch, datagen
import oracle_lib as O
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 8
only = sys.argv[2] if len(sys.argv) > 2 else None        # substring of the case name
modes = tuple(int(m) for m in sys.argv[3].split(",")) if len(sys.argv) > 3 else (0, 5)
N = 1 << 20
rng = np.random.default_rng(7)

def zipf_with_zero_pages():
    x = datagen.zipf_bytes(N, seed=11).copy()
    for off in (40000, 300000 + 123, 900001):
        x[off:off + 4096] = 0
    return x
def text_with_duplicate():
    x = datagen.text_bytes(N, seed=12).copy()
    x[600000:620000] = x[100000:120000]
    return x
def text_long_phrase():
    x = datagen.text_bytes(N, seed=13).copy()
    ph = rng.integers(97, 123, 2000, dtype=np.uint8)
    for off in range(5000, N - 2000, 16384):
        x[off:off + 2000] = ph
    return x
def log_with_runs():
    x = datagen.log_bytes(N, seed=14).copy()
    x[200000:200000 + 1500] = 32
    x[700000:700000 + 9000] = 0
    return x
CASES = {"zipf + 3 zero pages": zipf_with_zero_pages, "text + a duplicated 20 KB": text_with_duplicate,
         "text + a 2000-byte phrase every 16 KiB": text_long_phrase, "log + runs of 1500 / 9000": log_with_runs,
         "a 4 KiB page repeated": lambda: np.tile(rng.integers(0, 256, 4096, dtype=np.uint8), N // 4096),
         "two-byte period": lambda: np.tile(np.frombuffer(b"ab", dtype=np.uint8), N // 2),
         "one byte, then another at the end": lambda: np.concatenate([np.zeros(N - 1, dtype=np.uint8), np.ones(1, dtype=np.uint8)])}
dev = torch.device("cuda:0")
L = glc.lib()
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_BWT, N, rows=copies) as plan:
    for name, gen in CASES.items():
        if only and only not in name: continue
        x = gen()
        want, widx = O.bwt(x)
        d_in = torch.from_numpy(np.tile(x, copies)).to(dev)
        d_out = torch.zeros_like(d_in); d_idx = torch.zeros(copies, dtype=torch.int32, device=dev)
        line = "%-40s" % name
        for mode in modes:
            plan.set_sorter(mode)
            ts = []
            for it in range(3):
                d_out.zero_()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                assert L.glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), N, copies) == 0
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            got = d_out.cpu().numpy().reshape(copies, N); gi = d_idx.cpu().numpy()
            ok = all(np.array_equal(got[c], want) and int(gi[c]) == widx for c in range(copies))
            line += "  sorter %d: %s %8.2f ms per batch of %d, (flagged, given up, resumed) = %r" % (
                mode, "ok   " if ok else "WRONG", min(ts), copies, plan.last_sort_stats() + (plan.last_sort_resumed(),))
        print(line)
    plan.set_sorter(0)
