# phase stamps of k_huff_build for ONE block (library built with GLC_CXXFLAGS=-DGLC_HB_TIMING, picked with GLC_LIB)
import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
dev = torch.device("cuda:0")
d_in = bench.zipf_blocks_on_device(torch, dev, 1, 0, 1)
print(bench.leg_single_call(torch, glc, dev, d_in[:1 << 20], iters=10))
out = (C.c_ulonglong * 8)()
L = glc.lib()
L.glcDebugHuffStamps.argtypes = [C.c_void_p]
print("rc", L.glcDebugHuffStamps(out))
v = list(out)
names = ["hist", "tree", "codes", "words", "scan"]
for i, nm in enumerate(names): print("%-6s %8d cycles" % (nm, v[i + 1] - v[i]))
