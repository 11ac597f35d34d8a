"""Result collection for the multi-GPU path.

Blocks are encoded independently per rank (no data-path collective).  The one
exchange step is collecting the variable-length bitstreams on a root:
  1. all_gather of each rank's total word count (8 bytes per rank),
  2. one gather of the compacted streams, padded to the largest rank (the ranks
     hold equal numbers of equally sized blocks, so the padding is a few %).
`backend="nccl"` is RCCL over xGMI on ROCm: a gather to one root is 7 direct
links into the root, no ring.  Backend-agnostic so the logic is covered by a
world_size-2 gloo test on CPU (tests/test_dist_gather.py).
"""


def gather_streams(dist, torch, compact, compact_off, dst=0):
    """compact: int32 tensor holding this rank's streams back to back;
    compact_off: int64 [nblocks+1] word offsets (last = total).
    Returns on dst: dict(total_words, per_rank_words, buffers=[tensor per rank]); None elsewhere."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    if dist.get_backend() == "gloo" and compact.is_cuda:
        # gloo has no device-side gather: stage through the host (single-box dry runs of the N > 1 path)
        res = gather_streams(dist, torch, compact.cpu(), compact_off.cpu(), dst)
        if res is not None:
            res["buffers"] = [b.to(compact.device) for b in res["buffers"]]
        return res
    my_total = compact_off[-1:].clone()
    totals = [torch.empty_like(my_total) for _ in range(world)]
    dist.all_gather(totals, my_total)
    per_rank = [int(t.item()) for t in totals]
    maxw = max(per_rank)
    if maxw > compact.numel():
        raise RuntimeError("stream buffer too small for the padded gather")
    send = compact[:maxw]
    if rank == dst:
        bufs = [torch.empty(maxw, dtype=compact.dtype, device=compact.device) for _ in range(world)]
        dist.gather(send, gather_list=bufs, dst=dst)
        return {"total_words": sum(per_rank), "per_rank_words": per_rank,
                "buffers": [b[:w] for b, w in zip(bufs, per_rank)]}
    dist.gather(send, gather_list=None, dst=dst)
    return None
