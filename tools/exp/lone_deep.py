import importlib.util, os, sys, time, statistics
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py")); bench._GLC = glc
dev = torch.device("cuda:0"); n = 1 << 20
tb = bench.text_blocks_on_device(torch, dev, 2, seed=0x5EED0011).view(2, n).clone()
lb = bench.log_buffers_on_device(torch, dev, 2, seed=0x5EED0013).view(2, n).clone()
tb[0, 600000:620000] = tb[0, 100000:120000]
for o in range(5000, n - 2000, 16384): tb[1, o:o + 2000] = tb[1, :2000]
lb[0, 200000:201500] = 32; lb[0, 700000:709000] = 0
lb[1, 500000:520000] = lb[1, 40000:60000]
kinds = {"textdup": tb[0], "phrase": tb[1], "logruns": lb[0], "logdup": lb[1]}
for nb in (1, 2, 3):
  for mode in (0, 6):
    res = []
    for name, x in kinds.items():
        d = x.repeat(nb).contiguous()
        with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=nb) as plan:
            plan.set_sorter(mode)
            out = glc.compress_batch(plan, d, n, nb); plan.synchronize()
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                glc.compress_batch_into(plan, d, n, nb, out); plan.synchronize()
                ts.append(time.perf_counter() - t0)
            res.append("%s %.2f" % (name, statistics.median(ts) * 1e3))
    print("blocks per call %d, mode %d (6 = resume whatever the count): ms per call  " % (nb, mode) + "  ".join(res))
