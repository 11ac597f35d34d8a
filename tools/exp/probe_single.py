import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "zipf"
d_in = bench.zipf_blocks_on_device(torch, dev, 1, 0, 1) if kind == "zipf" else bench.text_blocks_on_device(torch, dev, 1).view(-1)
print(bench.leg_single_call(torch, glc, dev, d_in[:1 << 20], iters=50))
