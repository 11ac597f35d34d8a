"""The one exchange step of the multi-GPU path and its mirror.

Blocks are encoded independently: global block g lives on rank g % world as that rank's block g // world, with
no data-path collective.  What crosses GPUs is result collection on a root and, for decoding, its inverse:

  gather_blocks   1. all_gather of {blocks, total words} per rank (16 bytes each),
                  2. gather of the per-block RECORDS -- everything a decoder needs besides the words:
                     {compressedSize, bwtIndex, hist[256], encodeOffset[nsub]} -- fixed size per block,
                  3. gather-v of the word streams with grouped point-to-point operations (exact lengths, no
                     padding to the largest rank): on RCCL that is one ncclGroup of ncclRecv on the root and one
                     ncclSend per sender, i.e. 7 direct xGMI links into the root, no ring.
  scatter_blocks  the mirror: the root sends every rank its records and its words; each rank then expands them
                  (glcExpandStreams) and decodes its own blocks independently (glcDecompressBatch).

`backend="nccl"` is RCCL on ROCm; with gloo everything is staged through host tensors, which is how the logic is
covered on CPU (tests/test_dist_gather.py, world_size 2) and on a one-GPU box (tests/test_gpu_dist.py).
The device-side preparation is in the C ABI: glcCompactStreams / glcExpandStreams (include/cudpp.h).
"""

RECORD_FIXED = 2 + 256          # compressedSize, bwtIndex, hist[256]; followed by encodeOffset[nsub]


def pack_records(torch, out, nblk, nsub):
    """int32 [nblk, 258 + nsub] from the output arrays of glcCompressBatch (dict of tensors as glc_binding makes them)."""
    return torch.cat([out["size"][:nblk].view(nblk, 1), out["bwt_index"][:nblk].view(nblk, 1),
                      out["hist"][:nblk * 256].view(nblk, 256), out["offsets"][:nblk * nsub].view(nblk, nsub)], dim=1).contiguous()


def unpack_records(torch, rec, nsub):
    """inverse of pack_records: dict(size, bwt_index, hist, offsets) of contiguous int32 tensors"""
    nblk = rec.shape[0]
    return dict(size=rec[:, 0].contiguous(), bwt_index=rec[:, 1].contiguous(),
                hist=rec[:, 2:258].contiguous().view(nblk * 256),
                offsets=rec[:, 258:258 + nsub].contiguous().view(nblk * nsub), nsub=nsub)


def _host_staged(dist, t):
    return dist.get_backend() == "gloo" and t.is_cuda


def gather_blocks(dist, torch, compact, compact_off, records, dst=0):
    """compact: int32 tensor, this rank's streams back to back; compact_off: int64 [nblk+1] word offsets (last =
    total); records: int32 [nblk, R] (pack_records).  Returns on dst a dict
        nblk[r], words[r]         per rank
        buffers[r]                int32 tensor of exactly words[r] words (rank r's streams, back to back)
        records[r]                int32 [nblk[r], R]
        offsets[r]                int64 [nblk[r] + 1] word offsets into buffers[r] (from the record sizes)
    and None elsewhere.  Global block g is block g // world of rank g % world (see block_of)."""
    if _host_staged(dist, compact):
        res = gather_blocks(dist, torch, compact.cpu(), compact_off.cpu(), records.cpu(), dst)
        if res is not None:
            dev = compact.device
            for k in ("buffers", "records", "offsets"):
                res[k] = [t.to(dev) for t in res[k]]
        return res
    world, rank = dist.get_world_size(), dist.get_rank()
    nblk = records.shape[0]
    total = int(compact_off[nblk].item())
    head = torch.tensor([nblk, total], dtype=torch.int64, device=compact.device)
    heads = [torch.empty_like(head) for _ in range(world)]
    dist.all_gather(heads, head)
    nblks = [int(h[0].item()) for h in heads]
    words = [int(h[1].item()) for h in heads]
    R = records.shape[1]
    ops, bufs, recs = [], None, None
    if rank == dst:
        bufs = [compact[:total] if r == rank else torch.empty(words[r], dtype=compact.dtype, device=compact.device)
                for r in range(world)]
        recs = [records if r == rank else torch.empty((nblks[r], R), dtype=records.dtype, device=records.device)
                for r in range(world)]
        for r in range(world):
            if r != rank:
                if nblks[r]:
                    ops.append(dist.P2POp(dist.irecv, recs[r], r))
                if words[r]:
                    ops.append(dist.P2POp(dist.irecv, bufs[r], r))
    else:
        if nblk:
            ops.append(dist.P2POp(dist.isend, records, dst))
        if total:
            ops.append(dist.P2POp(dist.isend, compact[:total], dst))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if rank != dst:
        return None
    offs = []
    for r in range(world):
        o = torch.zeros(nblks[r] + 1, dtype=torch.int64, device=compact.device)
        if nblks[r]:
            o[1:] = torch.cumsum(recs[r][:, 0].to(torch.int64), 0)
        if int(o[-1].item()) != words[r]:
            raise RuntimeError("rank %d: record sizes (%d words) do not add up to the gathered stream (%d words)"
                               % (r, int(o[-1].item()), words[r]))
        offs.append(o)
    return {"nblk": nblks, "words": words, "buffers": bufs, "records": recs, "offsets": offs}


def scatter_blocks(dist, torch, gathered, src=0, device=None):
    """The mirror of gather_blocks.  `gathered` is its result on src (None elsewhere).  Every rank gets back
    (compact, compact_off, records) for ITS blocks: int32 words back to back, int64 [nblk+1] offsets, int32 records."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = gathered["buffers"][0].device if gathered is not None else torch.device("cpu")
    staged = dist.get_backend() == "gloo" and device.type == "cuda"
    wdev = torch.device("cpu") if staged else device
    if rank == src:
        head = torch.tensor([[gathered["nblk"][r], gathered["words"][r], gathered["records"][r].shape[1]]
                             for r in range(world)], dtype=torch.int64, device=wdev)
    else:
        head = torch.empty((world, 3), dtype=torch.int64, device=wdev)
    dist.broadcast(head, src)
    nblk, total, R = (int(x) for x in head[rank].tolist())
    ops = []
    if rank == src:
        keep = []
        for r in range(world):
            if r == rank:
                continue
            rec, buf = gathered["records"][r].to(wdev), gathered["buffers"][r].to(wdev)
            keep += [rec, buf]
            if rec.numel():
                ops.append(dist.P2POp(dist.isend, rec, r))
            if buf.numel():
                ops.append(dist.P2POp(dist.isend, buf, r))
        rec, buf = gathered["records"][rank].to(wdev), gathered["buffers"][rank].to(wdev)
    else:
        import torch as _t
        rec = _t.empty((nblk, R), dtype=_t.int32, device=wdev)
        buf = _t.empty(total, dtype=_t.int32, device=wdev)
        if rec.numel():
            ops.append(dist.P2POp(dist.irecv, rec, src))
        if buf.numel():
            ops.append(dist.P2POp(dist.irecv, buf, src))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    rec, buf = rec.to(device), buf.to(device)
    off = torch.zeros(nblk + 1, dtype=torch.int64, device=device)
    if nblk:
        off[1:] = torch.cumsum(rec[:, 0].to(torch.int64), 0)
    return buf, off, rec


def block_of(gathered, g):
    """words (int32 tensor) and record (int32 row) of GLOBAL block g in a gather_blocks result"""
    world = len(gathered["buffers"])
    r, i = g % world, g // world
    o = gathered["offsets"][r]
    return gathered["buffers"][r][int(o[i].item()):int(o[i + 1].item())], gathered["records"][r][i]


# kept for callers of the first round's interface: totals + streams only
def gather_streams(dist, torch, compact, compact_off, dst=0):
    nblk = compact_off.numel() - 1
    sizes = (compact_off[1:] - compact_off[:-1]).to(torch.int32).view(nblk, 1)
    res = gather_blocks(dist, torch, compact, compact_off, sizes, dst)
    if res is None:
        return None
    return {"total_words": sum(res["words"]), "per_rank_words": res["words"], "buffers": res["buffers"]}
