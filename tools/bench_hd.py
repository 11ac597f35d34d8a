#!/usr/bin/env python3
"""CUHD-shaped Huffman-only decode throughput (BASELINE.json configs[4], one GPU's share).
Symbols ~ Binomial(255, 0.5) as in the reference demo (demo.cc:93-105).  Encodes `--unique-mib`
MiB on the host, decodes `--mib` MiB on the device (the unique stream concatenated at unit
granularity is NOT a valid stream, so the device stream is one real stream of --mib symbols
built from a host encode of the whole input), reports decoded GB/s (output bytes / time, HIP
events), per-kernel split, and the oracle's bit-serial decoder on a bounded sample.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    import numpy as np
    import torch
    import oracle_lib as O
    glc = _load("glc_binding", os.path.join(ROOT, "gpu-lossless-compression_amd", "glc_binding.py"))
    L = glc.lib()
    dev = torch.device("cuda:0")
    n = args.mib << 20
    data = np.random.default_rng(5).binomial(255, 0.5, size=n).astype(np.uint8)
    hist = np.bincount(data, minlength=256).astype(np.uint64)
    lens, codes = glc.hd_build_table(hist)
    t0 = time.perf_counter()
    units = glc.hd_encode_host(data, lens, codes)
    t_enc = time.perf_counter() - t0
    d_units = torch.from_numpy(units.view(np.int32)).to(dev)
    work = torch.empty(L.glcHdWorkBytes(units.size), dtype=torch.uint8, device=dev)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev)

    def dec():
        assert L.glcHdDecodeDevice(d_units.data_ptr(), units.size, lens.ctypes.data, codes.ctypes.data,
                                   out.data_ptr(), n, work.data_ptr(), st.cuda_stream)

    dec()
    torch.cuda.synchronize()
    ok = bool(torch.equal(out.cpu(), torch.from_numpy(data)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(args.iters):
        dec()
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    # oracle on a bounded sample
    ns = min(n, 32 << 20)
    su = glc.hd_encode_host(data[:ns], lens, codes)
    t0 = time.perf_counter()
    O.hd_decode(su, lens, codes, ns)
    t_cpu = time.perf_counter() - t0
    print(json.dumps({
        "metric": "cuhd_decode_throughput", "unit": "GB/s", "value": n / ms / 1e6, "ms": ms,
        "symbols": n, "units": int(units.size), "ratio": n / (units.size * 4.0), "round_trip_ok": ok,
        "host_encode_GBps": n / t_enc / 1e9,
        "cpu_baseline": {"value": ns / t_cpu / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                         "sample": "%d MiB bit-serial oracle decode" % (ns >> 20)},
    }))


if __name__ == "__main__":
    main()
