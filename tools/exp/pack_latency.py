#!/usr/bin/env python3
"""k_lzss_pack_wave latency on candidate streams whose walks never fall into step (run under tools/exp/kstats.sh or
rocprofv3 --kernel-trace; 256 packets = one wave each, so the kernel time is the per-packet latency)."""
import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
spec = importlib.util.spec_from_file_location("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
glc = importlib.util.module_from_spec(spec); sys.modules["glc_binding"] = glc; spec.loader.exec_module(glc)
import datagen, oracle_lib as O
L = glc.lib()
n = 1 << 20
kind = sys.argv[1] if len(sys.argv) > 1 else "all_3"
if kind == "log":
    c = O.lzss_candidates(datagen.log_bytes(n))
else:
    k = int(kind.split("_")[1])
    room = 4096 - (np.arange(n) % 4096)
    ln = np.minimum(np.full(n, k, dtype=np.int64), room); ln[ln <= 2] = 1
    c = np.empty(2 * n, dtype=np.uint8); c[0::2] = ln.astype(np.uint8); c[1::2] = 7
L.initGPU()
buf, cand = L.initCPUmem(n), L.initCPUmem(2 * n)
C.memmove(cand, c.ctypes.data, 2 * n)
m = C.c_int(0)
for _ in range(3):
    rc = L.aftercompression_wrapper(buf, n, cand, C.byref(m))
print(kind, "rc", rc, "packed", m.value)
