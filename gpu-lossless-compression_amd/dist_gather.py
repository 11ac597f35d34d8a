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

The exchange itself is in the C ABI (include/glc_exchange.h, csrc/exchange.cpp: glcGatherCounts / glcGatherStreams /
glcScatterStreams over RCCL, device-side preparation glcCompactStreams / glcExpandStreams / glcPackRecords); `RcclExchange`
below is its caller and what bench.py uses when every rank has its own GPU.  RCCL refuses two ranks on one device and
does not exist on a CPU box, so the SAME protocol is also written over torch.distributed point-to-point operations
(gather_blocks / scatter_blocks: gloo, host-staged) -- that is how the logic is covered by tests/test_dist_gather.py
(world_size 2, CPU) and tests/test_gpu_dist.py (two processes on one MI355X).
"""
import ctypes as C

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


class RcclExchange:
    """Caller of the C-ABI exchange (include/glc_exchange.h).  One communicator per process; the unique id travels over
    torch.distributed (any backend) when the process group has more than one rank.  Results have the shape of
    gather_blocks / scatter_blocks so that callers can use either."""

    def __init__(self, glc, torch, dist=None):
        self.L, self.torch = glc.lib(), torch
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        uid = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            self._chk(self.L.glcCommGetUniqueId(uid.numpy().ctypes.data), "glcCommGetUniqueId")
        if self.world > 1:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            t = uid.to(dev)
            dist.broadcast(t, 0)
            uid = t.cpu()
        self.comm = C.c_void_p(0)
        self._chk(self.L.glcCommInitRank(C.byref(self.comm), self.world, uid.numpy().ctypes.data, self.rank), "glcCommInitRank")
        self.counts = (C.c_ulonglong * (2 * self.world))()

    @staticmethod
    def _chk(rc, what):
        if rc != 0:
            raise RuntimeError("%s -> %d" % (what, rc))

    def close(self):
        if self.comm:
            self.L.glcCommDestroy(self.comm)
            self.comm = C.c_void_p(0)

    def pack_records(self, out, nblk, nsub, first_block=0, stream=None):
        """device records [nblk, 258 + nsub] of blocks first_block .. first_block + nblk of glcCompressBatch's outputs"""
        torch = self.torch
        rec = torch.empty((nblk, RECORD_FIXED + nsub), dtype=torch.int32, device=out["size"].device)
        b = first_block
        self._chk(self.L.glcPackRecords(out["bwt_index"].data_ptr() + 4 * b, out["hist"].data_ptr() + 1024 * b,
                                        out["offsets"].data_ptr() + 4 * nsub * b, nsub, out["size"].data_ptr() + 4 * b, nsub, nblk,
                                        rec.data_ptr(), stream), "glcPackRecords")
        return rec

    def gather_begin(self, nwords_dev, records, stream=None):
        """first half of gather(): ENQUEUES the count exchange of a batch (glcGatherCountsBegin: all-gather + copy to pinned
        memory + event) and returns a ticket; the host does not wait."""
        t = C.c_int(-1)
        self._chk(self.L.glcGatherCountsBegin(self.comm, int(records.shape[0]), 0, nwords_dev, C.byref(t), stream), "glcGatherCountsBegin")
        return t.value

    def gather(self, compact, nwords_dev, records, dst=0, stream=None, out_words=None, out_records=None, ticket=None):
        """compact: int32 tensor holding this rank's streams back to back from its start; nwords_dev: device pointer to
        their word count (the last offset glcCompactStreams wrote); records: int32 [nblk, R].  On dst returns the
        gather_blocks dict (buffers / records are views of two rank-ordered arrays), None elsewhere.  Waits once (the
        counts -- for the exchange a gather_begin ticket names, if one is given); the transfers are only enqueued on `stream`."""
        torch = self.torch
        nblk, R = int(records.shape[0]), int(records.shape[1])
        if ticket is None:
            ticket = self.gather_begin(nwords_dev, records, stream)
        self._chk(self.L.glcGatherCountsEnd(self.comm, ticket, self.counts), "glcGatherCountsEnd")
        nblks = [int(self.counts[2 * r]) for r in range(self.world)]
        words = [int(self.counts[2 * r + 1]) for r in range(self.world)]
        allw = allr = None
        if self.rank == dst:
            allw = out_words if out_words is not None else torch.empty(max(1, sum(words)), dtype=torch.int32, device=compact.device)
            allr = out_records if out_records is not None else torch.empty((max(1, sum(nblks)), R), dtype=torch.int32, device=compact.device)
            if allw.numel() < sum(words) or allr.numel() < sum(nblks) * R:
                raise RuntimeError("gather: receive buffers too small")
        self._chk(self.L.glcGatherStreams(self.comm, dst, compact.data_ptr(), records.data_ptr(), R, self.counts,
                                          allw.data_ptr() if allw is not None else None,
                                          allr.data_ptr() if allr is not None else None, stream), "glcGatherStreams")
        if self.rank != dst:
            return None
        bufs, recs, offs, wo, bo = [], [], [], 0, 0
        flat = allr.view(-1)
        for r in range(self.world):
            bufs.append(allw[wo:wo + words[r]])
            recs.append(flat[bo * R:(bo + nblks[r]) * R].view(nblks[r], R))
            wo += words[r]; bo += nblks[r]
        return {"nblk": nblks, "words": words, "buffers": bufs, "records": recs, "offsets": None,
                "all_words": allw, "all_records": allr, "counts": list(self.counts)}

    def finish(self, gathered, stream_sync=True):
        """after the stream has run: per-rank word offsets from the record sizes (checked against the counts)"""
        torch = self.torch
        if gathered is None or gathered["offsets"] is not None:
            return gathered
        offs = []
        for r in range(self.world):
            o = torch.zeros(gathered["nblk"][r] + 1, dtype=torch.int64, device=gathered["all_words"].device)
            if gathered["nblk"][r]:
                o[1:] = torch.cumsum(gathered["records"][r][:, 0].to(torch.int64), 0)
            if int(o[-1].item()) != gathered["words"][r]:
                raise RuntimeError("rank %d: record sizes (%d words) do not add up to the gathered stream (%d words)"
                                   % (r, int(o[-1].item()), gathered["words"][r]))
            offs.append(o)
        gathered["offsets"] = offs
        return gathered

    def scatter(self, gathered, counts, R, src=0, stream=None, device=None):
        """the mirror: every rank gets (compact words, int64 offsets, records) of ITS blocks.  `counts`: the list the
        gather produced (known on every rank: glcGatherCounts is an all-gather)."""
        torch = self.torch
        cnt = (C.c_ulonglong * (2 * self.world))(*counts)
        nblk, total = int(cnt[2 * self.rank]), int(cnt[2 * self.rank + 1])
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        buf = torch.empty(max(1, total), dtype=torch.int32, device=device)
        rec = torch.empty((max(1, nblk), R), dtype=torch.int32, device=device)
        aw = gathered["all_words"].data_ptr() if gathered is not None else None
        ar = gathered["all_records"].data_ptr() if gathered is not None else None
        self._chk(self.L.glcScatterStreams(self.comm, src, aw, ar, R, cnt, buf.data_ptr(), rec.data_ptr(), stream), "glcScatterStreams")
        if stream is None:
            torch.cuda.synchronize(device)
        rec, buf = rec[:nblk], buf[:total]
        off = torch.zeros(nblk + 1, dtype=torch.int64, device=device)
        if nblk:
            off[1:] = torch.cumsum(rec[:, 0].to(torch.int64), 0)
        return buf, off, rec
