#!/usr/bin/env python3
"""Which of k_ss_sample's paths the inputs of tests/test_gpu_sample_sorter.py::test_sample_step_paths take: needs a library built
with -DGLC_SS_CLOCKS (GLC_LIB points at it).  Prints, per input and sorter mode: windows ranked pair by pair, long runs ordered
on their own, blocks sent to the network with text comparisons."""
import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_gpu_sample_sorter.py")); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
spec = importlib.util.spec_from_file_location("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py")); glc = importlib.util.module_from_spec(spec); spec.loader.exec_module(glc)
L = glc.lib()
L.glcSsClocks.argtypes = [C.c_void_p, C.c_int]
N = 1 << 20
inputs = {"text": datagen.text_bytes(N), "log": datagen.log_bytes(N), "word_every_90": t._text_with_word(N, 41, 90), "word_every_200": t._text_with_word(N, 42, 200),
          "word_every_40": t._text_with_word(N, 43, 40, tail=20), "tail_copy_40": t._text_with_tail_copy(N, 44, 40), "tail_copy_300": t._text_with_tail_copy(N, 45, 300),
          "tail_phrase_x40": np.concatenate([datagen.text_bytes(N - 4000, seed=47), np.tile(datagen.text_bytes(100, seed=48), 40)]),
          "tail_copy_3000_log": np.concatenate([datagen.log_bytes(N - 3000, seed=46), datagen.log_bytes(N, seed=46)[5000:8000]])}
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_BWT, N, rows=1) as plan:
    for name, x in inputs.items():
        for mode in (0, 6):
            plan.set_sorter(mode)
            buf = (C.c_ulonglong * 48)()
            L.glcSsClocks(buf, 1)
            t._bwt(glc, plan, torch, x)
            L.glcSsClocks(buf, 0)
            print("%-20s mode %d: k_ss_sample runs %d, windows pair by pair %d, long runs %d, blocks to the network %d; tiers gave up %r"
                  % (name, mode, buf[40], buf[41], buf[42], buf[43], plan.last_sort_stats()))
