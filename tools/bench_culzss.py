#!/usr/bin/env python3
"""CULZSS throughput (BASELINE.json configs[2]): log-style ASCII, 1 MiB buffers of 4096-byte
packets with a 128-byte window.  Reports, as one JSON line:
  * device-resident encode (match search + token selection + packing) and decode GB/s
    (HIP events on the launch stream, inputs resident in HBM),
  * the same through the reference's host-pointer wrapper ABI (PCIe inclusive; never the
    headline value),
  * compression ratio, parity of sampled buffers against the lock-step oracle,
  * the oracle (CPU port) timed on a bounded sample.
usage: bench_culzss.py [--gib 1.0] [--unique-mib 64]
"""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
MiB = 1 << 20


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=1.0)
    ap.add_argument("--unique-mib", type=int, default=64)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import torch
    import datagen
    import oracle_lib as O
    glc = _load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
    L = glc.lib()
    dev = torch.device("cuda:0")
    nbuf = max(1, int(args.gib * 1024))
    uniq = min(args.unique_mib, nbuf)
    host = np.concatenate([datagen.log_bytes(MiB, seed=0x5EED0003 + i) for i in range(uniq)])
    d_u = torch.from_numpy(host).to(dev)
    d_in = d_u.repeat((nbuf + uniq - 1) // uniq)[: nbuf * MiB].contiguous()
    stride = L.glcLzssPackStride(MiB)
    d_packed = torch.empty(nbuf * stride, dtype=torch.uint8, device=dev)
    d_sizes = torch.empty(nbuf, dtype=torch.int32, device=dev)
    d_work = torch.empty(L.glcLzssWorkBytes(MiB, nbuf), dtype=torch.uint8, device=dev)
    d_out = torch.empty(nbuf * MiB, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)
    sp = st.cuda_stream

    def enc():
        assert L.glcLzssEncodeDevice(d_in.data_ptr(), MiB, nbuf, None, d_packed.data_ptr(), d_sizes.data_ptr(),
                                     d_work.data_ptr(), sp) == 1

    def dec():
        assert L.glcLzssDecodeDevice(d_packed.data_ptr(), d_sizes.data_ptr(), MiB, nbuf, d_out.data_ptr(), sp) == 1

    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(args.iters):
            fn()
        e1.record(st); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters

    ms_enc = timed(enc)
    ms_dec = timed(dec)
    assert torch.equal(d_out, d_in), "CULZSS round trip failed"
    sizes = d_sizes.cpu().numpy().astype(np.int64)
    raw = int((sizes == 0).sum())
    comp_bytes = int(sizes.sum()) + raw * MiB
    # parity of a sample vs the oracle
    ok = 0
    pick = [0, uniq // 2, uniq - 1]
    t0 = time.perf_counter()
    for b in pick:
        blk = host[b * MiB:(b + 1) * MiB]
        want = O.lzss_pack(O.lzss_candidates(blk), MiB)
        got = d_packed[b * stride: b * stride + int(sizes[b])].cpu().numpy()
        ok += int(want is not None and np.array_equal(got, want))
    cpu_s = (time.perf_counter() - t0) / len(pick)
    # the reference's wrapper ABI (host pointers: H2D + kernels + D2H of 2x candidates and packed bytes)
    L.initGPU()
    buf, bufout = L.initCPUmem(MiB), L.initCPUmem(2 * MiB)
    in_d, out_d = L.initGPUmem(MiB), L.initGPUmem(2 * MiB)
    nwrap = 32
    n = C.c_int(0)
    t0 = time.perf_counter()
    for i in range(nwrap):
        C.memmove(buf, host[(i % uniq) * MiB:].ctypes.data, MiB)
        L.compression_kernel_wrapper(buf, MiB, bufout, 0, 0, 128, 0, i % 4, in_d, out_d)
        L.onestream_finish_GPU(i % 4)
        L.aftercompression_wrapper(buf, MiB, bufout, C.byref(n))
    wrap_s = (time.perf_counter() - t0) / nwrap
    L.deleteCPUmem(buf); L.deleteCPUmem(bufout); L.deleteGPUmem(in_d); L.deleteGPUmem(out_d)
    total = nbuf * MiB
    res = {
        "metric": "CULZSS encode/decode GB/s (input bytes), 1 MiB buffers, 4096-B packets, 128-B window",
        "config": {"workload": "configs[2]: %g GiB log-style ASCII (%d MiB unique, tiled), device resident" % (args.gib, uniq)},
        "encode_GBps": round(total / ms_enc / 1e6, 3), "decode_GBps": round(total / ms_dec / 1e6, 3),
        "encode_ms": round(ms_enc, 3), "decode_ms": round(ms_dec, 3),
        "compression_ratio": round(total / comp_bytes, 4), "raw_stored_buffers": raw,
        "roundtrip": "decode(encode(x)) == x on all %d buffers" % nbuf,
        "parity": "%d/%d sampled buffers byte-exact vs oracle" % (ok, len(pick)),
        "wrapper_abi_pcie_inclusive_GBps": round(MiB / wrap_s / 1e9, 4),
        "cpu_baseline": {"value": round(MiB / cpu_s / 1e9, 5), "unit": "GB/s", "cores": 1, "kind": "port",
                         "sample": "3 x 1 MiB buffers through oracle lock-step EncodeKernel emulation + aftercomp"},
    }
    print(json.dumps(res))


if __name__ == "__main__":
    main()
