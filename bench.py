#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic Zipf(1.0)-distributed bytes in
independent 1 MiB blocks, cudppCompress (BWT -> MTF -> Huffman) encode, inputs
resident in HBM before the timed region.  A "step" is one pass of the hot path
over the whole per-GPU input (--gib, default 4 GiB) in plan-sized batches.
For N > 1 the blocks are dealt round-robin (global block g -> rank g % N), each
rank encodes its own blocks with no data-path collective (weak scaling: the
per-GPU input is fixed).  The one exchange step of SURVEY.md 8(e) -- gathering the
compacted bitstreams on rank 0 over RCCL -- runs once after the timed region and is
reported as `gather_to_rank0` (--with-gather moves it into every timed step).

The timed encode leg uses one plan with its stages back to back, so that the launch time of the
dominant kernel reported under `roofline` is the kernel's own (--enc-pipeline and --enc-threads 2
--rows 128 overlap stages / plans for +5-10 % throughput).  The decode leg, reported beside it, uses
--plans host threads with stage pipelining.

Prints ONE JSON line on rank 0; `value` = whole-job input GB/s of the encode.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "gpu-lossless-compression_amd")
sys.path.insert(0, os.path.join(ROOT, "tests"))

MiB = 1 << 20
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def zipf_blocks_on_device(torch, dev, nblocks, first_global_block, stride_blocks, seed=0x5EED0002):
    """Zipf(1.0) bytes over 256 symbols, identity symbol permutation.  Block g of the
    global stream is generated from seed+g so any rank / any N produces the same bytes."""
    out = torch.empty(nblocks * MiB, dtype=torch.uint8, device=dev)
    p = 1.0 / torch.arange(1, 257, dtype=torch.float64)
    cdf = torch.cumsum(p / p.sum(), 0).to(torch.float32).to(dev)
    gen = torch.Generator(device=dev)
    for i in range(nblocks):
        g = first_global_block + i * stride_blocks
        gen.manual_seed(seed + g)
        u = torch.rand(MiB, generator=gen, device=dev)
        out[i * MiB:(i + 1) * MiB] = torch.searchsorted(cdf, u).clamp_(max=255).to(torch.uint8)
    return out


def cpu_baseline(sample_blocks):
    """The oracle (a port of the reference algorithm, same bitstream) timed on the
    host cores of this box over a bounded sample of the same workload."""
    import bz2
    from concurrent.futures import ThreadPoolExecutor

    import oracle_lib as O
    O.lib()
    t0 = time.perf_counter()
    for blk in sample_blocks[:8]:
        O.compress(blk)
    t1 = time.perf_counter()
    single = 8 * MiB / (t1 - t0) / 1e9
    ncores = os.cpu_count() or 1
    import numpy as np
    flat = np.concatenate(sample_blocks)
    t0 = time.perf_counter()
    O.compress_many(flat, MiB, ncores)                      # pthreads inside the oracle, one block per thread
    t1 = time.perf_counter()
    allcore = len(sample_blocks) * MiB / (t1 - t0) / 1e9
    t0 = time.perf_counter()
    for blk in sample_blocks[:4]:
        bz2.compress(blk.tobytes(), 9)
    t1 = time.perf_counter()
    bz = 4 * MiB / (t1 - t0) / 1e9
    return {"value": round(single, 5), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "8 x 1 MiB Zipf blocks of this workload through oracle/glc_oracle.c orc_compress (single thread)",
            "all_cores_value": round(allcore, 5), "all_cores": ncores,
            "all_cores_sample": "%d blocks, one block per thread" % len(sample_blocks),
            "libbz2_9_single_core_GBps": round(bz, 5)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gib", type=float, default=4.0, help="input per GPU (GiB)")
    ap.add_argument("--rows", type=int, default=256, help="blocks per batched call (plan rows)")
    ap.add_argument("--plans", type=int, default=3, help="plans (each with its own stream) per GPU")
    ap.add_argument("--enc-threads", type=int, default=1,
                    help="host threads (= plans) used by the timed encode leg; 2 x 128-block plans give +7-10 %% "
                         "throughput, but kernels of concurrent sorts share the machine, so the per-launch duration "
                         "that `roofline` reports no longer describes the kernel")
    ap.add_argument("--dec-threads", type=int, default=3, help="host threads (= plans) used by the decode leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--enc-pipeline", action="store_true",
                    help="stage pipelining in the timed encode leg too (+4-8 %% throughput; MTF + Huffman of batch i then "
                         "share the machine with the sort of batch i+1, so the per-launch time of the roofline kernel is "
                         "no longer its own)")
    ap.add_argument("--no-dec-pipeline", action="store_true", help="decode leg: no stage pipelining")
    ap.add_argument("--with-gather", action="store_true",
                    help="N>1: include the RCCL gather of the bitstreams to rank 0 in the timed region")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d ...`" % (args.gpus, args.gpus))
    # GLC_BENCH_ONE_DEVICE=1: dry run of the N > 1 code path on a one-GPU box (all ranks share device 0 and
    # talk over gloo; RCCL refuses two ranks on one device).  Not a measurement.
    one_device = os.environ.get("GLC_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    glc = _load("glc_binding", os.path.join(PKG, "glc_binding.py"))
    glc.lib()                                             # fails loudly if the HIP library is missing
    gather_mod = _load("glc_dist", os.path.join(PKG, "dist_gather.py"))

    nblocks = max(1, int(args.gib * 1024))
    rows = min(args.rows, nblocks)
    n = MiB
    d_in = zipf_blocks_on_device(torch, dev, nblocks, rank, world)
    nsub = n // 4096
    stride = glc.compressed_stride_words(n)
    out = dict(bwt_index=torch.empty(nblocks, dtype=torch.int32, device=dev),
               hist=torch.empty(nblocks * 256, dtype=torch.int32, device=dev),
               offsets=torch.empty(nblocks * nsub, dtype=torch.int32, device=dev),
               size=torch.empty(nblocks, dtype=torch.int32, device=dev),
               words=torch.empty(nblocks * stride, dtype=torch.int32, device=dev))
    compact = torch.empty(nblocks * stride, dtype=torch.int32, device=dev)
    compact_off = torch.empty(nblocks + 1, dtype=torch.int64, device=dev)
    L = glc.lib()
    ctx = glc.Cudpp()
    # CUDPP's convention is one plan per host thread.  Several plans driven by their own host threads and
    # streams keep more of the machine busy than one (the sort of a batch is a chain of latency-bound
    # kernels with one host round trip per round): measured on MI355X, 4 GiB: 1 plan x 256 blocks 21.2 GB/s
    # encode / 27.2 decode, 2 x 128 23.6 / 27.8, 3 x 128 22.3 / 31.1, 3 x 256 23.0 / 30.4.  The timed encode
    # leg defaults to ONE plan so that the dominant kernel's launch time (roofline) is its own; the decode
    # leg uses all of them.
    nplans = max(1, args.plans)
    plans, streams = [], []
    for _ in range(nplans):
        pl = glc.Plan(ctx, glc.CUDPP_COMPRESS, n, rows=rows)
        st_ = torch.cuda.current_stream(dev) if nplans == 1 else torch.cuda.Stream(dev)
        pl.set_stream(st_.cuda_stream)
        pl.set_pipelining(bool(args.enc_pipeline))
        plans.append(pl)
        streams.append(st_)
    plan = plans[0]
    batches = list(range(0, nblocks, rows))
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=nplans)

    def run_threads(fn, nthreads):
        nthreads = max(1, min(nthreads, nplans))
        if nthreads == 1:
            fn(0, 1)
            return
        for f in [pool.submit(fn, t, nthreads) for t in range(nthreads)]:
            f.result()

    def enc_worker(t, nt):
        torch.cuda.set_device(dev)                            # the HIP device is per host thread
        pl = plans[t]
        for b0 in batches[t::nt]:
            nb = min(rows, nblocks - b0)
            rc = L.glcCompressBatch(pl.handle, d_in.data_ptr() + b0 * n, out["bwt_index"].data_ptr() + 4 * b0,
                                    out["hist"].data_ptr() + 1024 * b0, out["offsets"].data_ptr() + 4 * nsub * b0, nsub,
                                    out["size"].data_ptr() + 4 * b0, out["words"].data_ptr() + 4 * stride * b0, stride, n, nb)
            if rc != 0:
                raise RuntimeError("glcCompressBatch -> %d" % rc)
        pl.synchronize()

    def encode_all():
        run_threads(enc_worker, args.enc_threads)
        rc = L.glcCompactStreams(plan.handle, out["words"].data_ptr(), stride, out["size"].data_ptr(), nblocks,
                                 compact.data_ptr(), compact_off.data_ptr())
        if rc != 0:
            raise RuntimeError("glcCompactStreams -> %d" % rc)
        plan.synchronize()                                    # the compacted streams are complete for any stream

    def step():
        # the hot path: every rank encodes its own blocks; nothing crosses GPUs.
        # --with-gather puts the result collection on rank 0 inside the timed region as well.
        encode_all()
        if world > 1 and args.with_gather:
            return gather_mod.gather_streams(dist, torch, compact, compact_off, dst=0)
        return None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    for pl in plans:
        pl.synchronize()
        pl.enable_timing(3)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gathered = step()
    barrier()
    t1 = time.perf_counter()
    kp = {"ms": 0.0, "launches": 0, "bytes": 0.0}
    for pl in plans:
        pl.synchronize()
        k1 = pl.kernel_profile()
        for key in kp:
            kp[key] += k1[key]
    stage_ms = plan.last_timing()
    for pl in plans:
        pl.enable_timing(0)

    # result collection (the one exchange step of the multi-GPU path), timed on its own
    gather_ms = None
    if world > 1:
        barrier()
        tg0 = time.perf_counter()
        gathered = gather_mod.gather_streams(dist, torch, compact, compact_off, dst=0)
        barrier()
        gather_ms = (time.perf_counter() - tg0) * 1e3
        del gathered

    # decode leg (not part of `value`): every block back through the HIP
    # decoder, then the full-size property check decode(encode(x)) == x on all bytes
    d_back = torch.empty_like(d_in)

    def dec_worker(t, nt):
        torch.cuda.set_device(dev)
        pl = plans[t]
        for b0 in batches[t::nt]:
            nb = min(rows, nblocks - b0)
            rc = L.glcDecompressBatch(pl.handle, out["bwt_index"].data_ptr() + 4 * b0, out["hist"].data_ptr() + 1024 * b0,
                                      out["offsets"].data_ptr() + 4 * nsub * b0, nsub, out["words"].data_ptr() + 4 * stride * b0,
                                      stride, d_back.data_ptr() + b0 * n, n, nb)
            if rc != 0:
                raise RuntimeError("glcDecompressBatch -> %d" % rc)
        pl.synchronize()

    def decode_all():
        run_threads(dec_worker, args.dec_threads)

    for pl in plans:                                          # second half of a call overlaps the first half of the next
        pl.set_pipelining(not args.no_dec_pipeline)

    decode_all()
    barrier()
    td0 = time.perf_counter()
    decode_all()
    barrier()
    td1 = time.perf_counter()
    roundtrip_ok = bool(torch.equal(d_back, d_in))
    if not roundtrip_ok:
        raise RuntimeError("round trip failed: decode(encode(x)) != x")
    del d_back
    dec_elapsed = torch.tensor([td1 - td0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dec_elapsed, op=dist.ReduceOp.MAX)
    decode_gbps = float(nblocks) * n * world / float(dec_elapsed.item()) / 1e9

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    total_bytes = float(nblocks) * n * world * args.steps
    value = total_bytes / elapsed / 1e9

    # compression ratio + parity of a sample against the oracle (after the timed region)
    sizes = out["size"].cpu().numpy().astype(np.int64)
    ratio = float(nblocks * n) / float(sizes.sum() * 4)
    verify = None
    sample_host = []
    if rank == 0:
        pick = sorted(set(int(x) for x in np.linspace(0, nblocks - 1, min(nblocks, max(16, os.cpu_count() or 16))).astype(int)))
        sample_host = [d_in[b * n:(b + 1) * n].cpu().numpy() for b in pick]
        if not args.no_verify:
            import oracle_lib as O
            okc = 0
            for b, blk in zip(pick[:4], sample_host[:4]):
                want = O.compress(blk)
                got = out["words"][b * stride: b * stride + int(sizes[b])].cpu().numpy().view(np.uint32)
                okc += int(int(out["bwt_index"][b].item()) == want["bwt_index"] and int(sizes[b]) == want["size"]
                           and np.array_equal(got, want["words"]))
            verify = "%d/4 sampled blocks bit-exact vs oracle" % okc
            if okc != 4:
                raise RuntimeError("parity failure in bench sample: " + verify)

    if rank == 0:
        avg_ms = kp["ms"] / max(1, kp["launches"])
        per_launch_bytes = kp["bytes"] / max(1, kp["launches"])
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("k_rs_onesweep8_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "encode+decode GB/s (input bytes) per GPU and whole-node; compression ratio parity",
            "value": round(value, 4), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[1]: %g GiB/GPU Zipf(1.0) bytes, 1 MiB blocks, cudppCompress BWT+MTF+Huffman encode"
                                   % args.gib,
                       "value_is": "encode input bytes of all ranks / wall time (inputs resident in HBM; no data-path collective"
                                   + ("; RCCL gather of the bitstreams to rank 0 included)" if args.with_gather else ")"),
                       "block_bytes": n, "blocks_per_gpu": nblocks, "batch_rows": rows,
                       "plans_per_gpu": nplans, "encode_host_threads": min(args.enc_threads, nplans),
                       "decode_host_threads": min(args.dec_threads, nplans),
                       "stage_pipelining": {"encode": bool(args.enc_pipeline), "decode": not args.no_dec_pipeline},
                       "parallelism": "blocks round-robin over %d GPU(s), no data-path collective" % world},
            "compression_ratio": round(ratio, 4),
            "decode_GBps": round(decode_gbps, 4),
            "roundtrip": "decode(encode(x)) == x on all %d blocks per GPU" % nblocks,
            "frac_of_hbm_read_roofline": round(value / world / HBM_PEAK_GBPS, 6),
            "stage_ms_last_batch": {"bwt": round(stage_ms[0], 3), "mtf": round(stage_ms[1], 3),
                                    "huffman": round(stage_ms[2], 3), "total": round(stage_ms[3], 3)},
            "roofline": {"kernel": "glc::k_rs_onesweep<8, false> (stable LSD radix scatter with decoupled look-back, suffix sorter)",
                         "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "avg_launch_ms": round(avg_ms, 4), "launches": kp["launches"],
                         "algorithmic_bytes_per_launch": round(per_launch_bytes, 1),
                         "timing": "hipEvent pairs on the launch stream around every launch inside the timed region"},
            "parity": verify,
        }
        if gather_ms is not None:
            res["gather_to_rank0"] = {"ms": round(gather_ms, 2),
                                      "what": "all_gather of totals + padded gather of the compacted streams (RCCL), outside the timed region"}
        if not args.no_cpu_baseline and world == 1:          # host-side baseline: rank 0 at N=1 only
            res["cpu_baseline"] = cpu_baseline(sample_host)
        print(json.dumps(res))
    pool.shutdown()
    for pl in plans:
        pl.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
