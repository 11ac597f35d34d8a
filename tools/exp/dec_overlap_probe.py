#!/usr/bin/env python3
"""Does the decoder's stage pipelining overlap anything?  ONE plan, K calls of glcDecompressBatch back to back on a batch of
`rows` Zipf blocks: stages back to back (pipelining off) against stage B of call k (inverse BWT: LF + walk + emit, side
stream) under stage A of call k + 1 (Huffman + inverse MTF, the plan's stream).  The per-kernel profile of the serial pass
says what stage A and stage B cost on their own, i.e. what perfect overlap would give.
usage: dec_overlap_probe.py [rows] [calls]     env: GLC_LIB (variant build, e.g. -DGLC_WALK_PAD=86), GLC_SIDE_PRIO=least|same|greatest"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
import torch
glc = bench._load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
n = 1 << 20
bench._GLC = glc
d_in = bench.zipf_blocks_on_device(torch, dev, rows, 0, 1)
with glc.Cudpp() as ctx, glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=rows) as plan:
    comp = glc.compress_batch(plan, d_in, n, rows)
    outs = [torch.empty_like(d_in) for _ in range(2)]

    def run(k):
        for i in range(k):
            glc.decompress_batch(plan, comp, n, rows, outs[i & 1])
        plan.synchronize(); torch.cuda.synchronize()

    modes = {'pipe': (True,), 'serial': (False,)}.get(os.environ.get('DEC_PROBE_MODES', ''), (False, True, False, True))
    for mode in modes:
        plan.set_pipelining(mode)
        run(2)
        t0 = time.perf_counter(); run(calls); t = (time.perf_counter() - t0) * 1e3 / calls
        print("pipelining %-5s  %.3f ms per call of %d blocks  (%.2f GB/s)" % (mode, t, rows, rows * n / t / 1e6), flush=True)
    if os.environ.get('DEC_PROBE_MODES'):
        sys.exit(0)
    plan.set_pipelining(False)
    plan.enable_timing(3)
    run(2)
    prof = plan.kernel_profiles()
    plan.enable_timing(0)
    a = b = 0.0
    for k, v in sorted(prof.items()):
        if not k.startswith(("k_dec", "k_imtf", "k_ibwt")) or not v["launches"]:
            continue
        ms = v["ms"] / v["launches"]
        print("   %-36s %.3f ms per launch" % (k, ms))
        if k.startswith("k_ibwt"): b += ms
        else: a += ms
    print("stage A (Huffman + iMTF) %.3f ms, stage B (iBWT) %.3f ms, sum %.3f, max %.3f" % (a, b, a + b, max(a, b)))
    back = glc.decompress_batch(plan, comp, n, rows)
    print("round trip:", bool(torch.equal(back, d_in)))
