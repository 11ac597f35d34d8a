#!/usr/bin/env python3
"""Where k_ss_cut workgroups (thread 0) and k_ss_windows waves spend their life: needs a library built with
-DGLC_SS_CLOCKS (GLC_LIB points at it).
usage: GLC_LIB=... ss_clocks.py [text256|log256] [rows]   -- s_memrealtime ticks (100 MHz) per phase"""
import ctypes as C, importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
kind = sys.argv[1] if len(sys.argv) > 1 else "text256"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
n = 1 << 20
bench._GLC = glc
d_in = (bench.text_blocks_on_device(torch, dev, rows) if kind == "text256" else bench.log_buffers_on_device(torch, dev, rows)).view(-1)
if kind == "logruns":                                          # pd_batch.py's kind: log lines with 1500 spaces and 9000 zero bytes inside
    v = d_in.view(rows, n)
    v[:, 200000:201500] = 32
    v[:, 700000:709000] = 0
L = glc.lib()
L.glcSsClocks.argtypes = [C.c_void_p, C.c_int]
cut = ["load words", "gather + sort samples", "merge pivots", "bin", "scan + scatter", "list long bins", "-", "write back"]
smp = ["draw samples", "integer sort", "-", "runs ordered", "-", "text network", "tol check + splitters + l0", "cells"]
win = ["prologue", "window words", "round setup", "gather", "count", "move + re-read", "rows", "-"]
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=rows) as plan:
    if os.environ.get("GLC_SORTER"): plan.set_sorter(int(os.environ["GLC_SORTER"]))   # (5: the exact form alone on deep blocks)
    out = glc.compress_batch(plan, d_in, n, rows)
    plan.synchronize()
    for it in range(2):
        buf = (C.c_ulonglong * 48)()
        L.glcSsClocks(buf, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        glc.compress_batch_into(plan, d_in, n, rows, out)
        plan.synchronize()
        t1 = time.perf_counter()
        L.glcSsClocks(buf, 0)
        for base, names, what in ((0, cut, "k_ss_cut workgroup"), (16, win, "k_ss_windows wave"), (32, smp, "k_ss_sample workgroup")):
            tot = sum(buf[base:base + 8]); cnt = max(1, buf[base + 8])
            print("%s batch %.2f ms; %s: %d of them, %.2f us each: " % (kind, (t1 - t0) * 1e3, what, cnt, tot / cnt / 100.0) +
                  ", ".join("%s %.2f us" % (names[i], buf[base + i] / cnt / 100.0) for i in range(8)))
        r = max(1, buf[25])
        print("   rounds per wave %.2f; per round: summed wave-max run length of the 4 item loops %.1f, undecided positions %.1f, window %.1f positions"
              % (r / max(1, buf[24]), buf[26] / r, buf[27] / r, buf[28] / r))
        print("   members per round by run size: 2: %.1f, 3-4: %.1f, 5-16: %.1f, 17-64: %.1f, 65+: %.1f; rounds with <= 8 undecided %.3f, longest run <= 2: %.3f, <= 4: %.3f, <= 16: %.3f"
              % (buf[9] / r, buf[10] / r, buf[11] / r, buf[12] / r, buf[13] / r, buf[14] / r, buf[29] / r, buf[15] / r, buf[30] / r))
        print("   k_ss_sample: windows pair by pair %d, long runs %d, blocks to the network %d; slowest 'runs ordered' exact %.1f us, tolerant %.1f us"
              % (buf[41], buf[42], buf[43], buf[44] / 100.0, buf[45] / 100.0))
