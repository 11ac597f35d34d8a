#!/usr/bin/env python3
"""Hunt for rows that share the cap with their predecessor and go unmarked, and for rows marked as continuations up front (members
of a run left at the cap) that do not share it (library built with -DGLC_DEBUG_CAND)."""
import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
spec = importlib.util.spec_from_file_location("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py")); glc = importlib.util.module_from_spec(spec); spec.loader.exec_module(glc)
L = glc.lib()
L.glcDebugCand.argtypes = [C.c_void_p, C.c_int]
N = 1 << 20
spec2 = importlib.util.spec_from_file_location("tres", os.path.join(ROOT, "tests", "test_gpu_resume.py")); tres = importlib.util.module_from_spec(spec2); spec2.loader.exec_module(tres)
kind = sys.argv[2] if len(sys.argv) > 2 else "tail_run"
blocks = [tres.GENS[kind](N, 100 + 7 * i) for i in range(4)]
x = np.concatenate(blocks)
d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros_like(d_in); d_idx = torch.zeros(4, dtype=torch.int32, device="cuda")
buf = (C.c_uint * 64)()
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_BWT, N, rows=4) as plan:
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
        L.glcDebugCand(None, 1)
        assert L.glcBwtBatch(plan.handle, d_in.data_ptr(), d_out.data_ptr(), d_idx.data_ptr(), N, 4) == 0
        torch.cuda.synchronize()
        L.glcDebugCand(buf, 0)
        if buf[0]:
            print("iteration %d: %d unmarked rows that share the cap" % (it, buf[0]))
            for k in range(min(buf[0], 15)):
                b, r, a, c = buf[1 + 4 * k: 5 + 4 * k]
                pre, b = b >> 31, b & 0x7FFFFFFF                # (bit 31: a row marked GRP_SAME up front that does NOT share the cap)
                if pre: print("   the next one: marked as a continuation up front, and is not")
                xa, xc = blocks[b][a:a + 160], blocks[b][c:c + 160]
                lcp = int(np.argmax(xa[:min(len(xa), len(xc))] != xc[:min(len(xa), len(xc))])) if not np.array_equal(xa[:min(len(xa), len(xc))], xc[:min(len(xa), len(xc))]) else min(len(xa), len(xc))
                print("   block %d row %d: suffixes %d, %d (from the end: %d, %d), common prefix >= %d" % (b, r, a, c, N - a, N - c, lcp))
print("done", kind)
