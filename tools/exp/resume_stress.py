#!/usr/bin/env python3
"""Random blocks with deep parts inside (pasted duplicates, runs, periodic stretches of random length and place, in text, log
lines, Zipf or float bytes), batches of 4-8, random block sizes: BWT + index against the oracle with the resumed doubling on
(glcPlanSetSorter 6).  usage: resume_stress.py [batches] [seed]"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import numpy as np, torch, datagen
import oracle_lib as O
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
nbatch = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
L = glc.lib()
base = [lambda n, s: datagen.text_bytes(n, seed=s), lambda n, s: datagen.log_bytes(n, seed=s),
        lambda n, s: datagen.zipf_bytes(n, seed=s), lambda n, s: datagen.float_bytes(n, seed=s)]

def deepen(x):
    n = x.size
    for _ in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(0, 4))
        ln = int(min(n // 3, rng.integers(140, 60000)))
        a = int(rng.integers(0, n - ln))
        if kind == 0:                                           # a region pasted in a second time
            b = int(rng.integers(0, n - ln)); x[b:b + ln] = x[a:a + ln].copy()
        elif kind == 1:                                         # a run of one byte
            x[a:a + ln] = int(rng.integers(0, 256))
        elif kind == 2:                                         # a periodic stretch
            p = int(rng.integers(2, 40)); x[a:a + ln] = np.resize(x[a:a + p].copy(), ln)
        else:                                                   # the same phrase many times
            ph = x[a:a + min(ln, 3000)].copy()
            for o in range(int(rng.integers(0, 4096)), n - ph.size, int(rng.integers(ph.size + 1, 8 * ph.size + 2))):
                x[o:o + ph.size] = ph
    return x

bad = 0
t0 = time.time()
with glc.Cudpp() as ctx:
    for it in range(nbatch):
        n = int(rng.choice([1 << 20, 1 << 20, 300007, 65536 + 17, 999999]))
        rows = int(rng.integers(4, 9))
        blocks = [deepen(base[int(rng.integers(0, 4))](n, int(rng.integers(1, 1 << 30))).copy()) if rng.random() < 0.85
                  else base[int(rng.integers(0, 4))](n, int(rng.integers(1, 1 << 30))) for _ in range(rows)]
        x = np.concatenate(blocks)
        with glc.Plan(ctx, glc.CUDPP_BWT, n, rows=rows) as plan:
            plan.set_sorter(6 if it % 3 else 0)
            d_in = torch.from_numpy(x).to(dev)
            d_out = torch.zeros_like(d_in); d_idx = torch.zeros(rows, dtype=torch.int32, device=dev)
            assert L.glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), n, rows) == 0
            torch.cuda.synchronize()
            got = d_out.cpu().numpy(); gi = d_idx.cpu().numpy()
            stats = plan.last_sort_stats() + (plan.last_sort_resumed(),)
        ok = True
        for i, blk in enumerate(blocks):
            want, widx = O.bwt(blk)
            if int(gi[i]) != widx or not np.array_equal(got[i * n:(i + 1) * n], want):
                ok = False; bad += 1
                np.save("/tmp/resume_bad_%d_%d.npy" % (it, i), blk)
                print("MISMATCH batch %d block %d (n = %d)" % (it, i, n))
        print("batch %2d: n = %7d, %d blocks, (flagged, given up, resumed) = %r  %s   [%.0f s]" % (it, n, rows, stats, "ok" if ok else "WRONG", time.time() - t0), flush=True)
print("done: %d mismatching blocks" % bad)
sys.exit(1 if bad else 0)
