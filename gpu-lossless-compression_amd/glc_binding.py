"""ctypes binding of libglc_amd.so -- the host-side mirror, in Python, of what a
C caller of include/cudpp.h / include/culzss.h does.  Device memory comes from
torch (plumbing only): every call passes raw device pointers and sizes.

There is no CPU fallback: if the library is missing or a call fails, this
raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GLC_LIB") or os.path.join(HERE, "libglc_amd.so")    # GLC_LIB: A/B builds of the same library

# enum values of include/cudpp.h (identical to the reference header)
CUDPP_SUCCESS = 0
CUDPP_ERROR_INVALID_HANDLE = 1
CUDPP_ERROR_ILLEGAL_CONFIGURATION = 2
CUDPP_ERROR_INVALID_PLAN = 3
CUDPP_ERROR_INSUFFICIENT_RESOURCES = 4
CUDPP_ERROR_UNKNOWN = 9999
CUDPP_UCHAR = 1
CUDPP_UINT = 5
CUDPP_ADD = 0
CUDPP_SCAN = 0
CUDPP_COMPRESS = 10
CUDPP_BWT = 12
CUDPP_MTF = 13
CUDPP_SA = 14
CUDPP_INVALID_HANDLE = 0xC0DABAD1
CUDPP_OPTION_FORWARD = 0x1
CUDPP_OPTION_BACKWARD = 0x2

HUFF_BLOCK = 4096
HUFF_MAX_WORDS = 1536
GLC_HD_MAX_LEN = 11

CUDPP_SYMBOLS = [
    "cudppCreate", "cudppDestroy", "cudppPlan", "cudppDestroyPlan", "cudppCompress",
    "cudppBurrowsWheelerTransform", "cudppMoveToFrontTransform", "cudppSuffixArray",
    "glcCompressBatch", "glcBwtBatch", "glcMtfBatch", "glcDecompressBatch", "glcPlanSetStream",
    "glcPlanSynchronize", "glcPlanEnableTiming", "glcPlanLastTiming", "glcPlanKernelProfile",
    "glcCompactStreams", "glcPlanSetPipelining", "glcPlanSetSorter", "glcPlanLastSortStats", "glcPlanLastSortStatsEx", "glcPlanLastSortRetries", "glcPlanLastSortResumed", "glcPlanLastSortPeriodic", "glcPlanLastSortSkipped", "glcPlanDebugSortFlags", "glcPlanDebugBucketFill", "glcHuffmanEncodeBatch", "glcExpandStreams", "glcPlanKernelProfileEx", "glcProbeStreamRead", "glcGenZipfPhilox", "glcGenFloatPhilox", "glcPlanKernelProfileLost",
    "glcCompressBatchCompact", "glcDecompressBatchCompact",
]
CULZSS_SYMBOLS = [
    "compression_kernel_wrapper", "aftercompression_wrapper", "decompression_kernel_wrapper",
    "onestream_finish_GPU", "initGPUmem", "initCPUmem", "deleteGPUmem", "deleteCPUmem", "initGPU",
    "resetGPU", "streams_in_GPU", "deleteGPUStreams", "signalExitThreads", "deinitGPUmem",
    "dedeleteGPUmem", "deinitGPU", "culzss_compress", "culzss_decompress",
    "glcLzssEncodeDevice", "glcLzssDecodeDevice", "glcLzssLastKernelMs", "glcLzssPackStride",
    "glcLzssWorkBytes", "culzss_container_bound", "culzss_container_compress", "culzss_container_decompress",
    "culzss_compress_file", "culzss_decompress_file", "glcLzssEnableProfile", "glcLzssKernelProfile",
]
EXCHANGE_SYMBOLS = ["glcCommGetUniqueId", "glcCommInitRank", "glcCommAdopt", "glcCommDestroy", "glcCommInfo", "glcPackRecords",
                    "glcUnpackRecords", "glcGatherCounts", "glcGatherCountsBegin", "glcGatherCountsReady", "glcGatherCountsEnd",
                    "glcGatherStreams", "glcScatterStreams"]
HD_SYMBOLS = ["glcHdBuildTable", "glcHdEncodeHost", "glcHdWorkBytes", "glcHdDecodeDevice", "glcHdDecodeDeviceTable", "glcHdDecodeDeviceTableOnDevice", "glcHdEnableProfile",
              "glcHdKernelProfile"]


class CUDPPConfiguration(C.Structure):
    _fields_ = [("algorithm", C.c_int), ("op", C.c_int), ("datatype", C.c_int),
                ("options", C.c_uint), ("bucket_mapper", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libglc_amd.so is not built: run `python __graft_entry__.py` "
                           "(there is no CPU fallback)")
    # torch bundles its own libamdhip64.so.7 (ROCm 7.0) while the library links the
    # system one (ROCm 7.2) under the same soname: whichever is loaded first serves
    # both.  A process that uses torch for device memory must load torch's first,
    # or the two HIP runtimes disagree (hipGetDevice fails).  Pure C callers have
    # no torch and are unaffected.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, sz, u = C.c_void_p, C.c_size_t, C.c_uint
    L.cudppCreate.argtypes = [C.POINTER(sz)]
    L.cudppDestroy.argtypes = [sz]
    L.cudppPlan.argtypes = [sz, C.POINTER(sz), CUDPPConfiguration, sz, sz, sz]
    L.cudppDestroyPlan.argtypes = [sz]
    L.cudppCompress.argtypes = [sz, vp, vp, vp, vp, vp, vp, vp, sz]
    L.cudppBurrowsWheelerTransform.argtypes = [sz, vp, vp, vp, sz]
    L.cudppMoveToFrontTransform.argtypes = [sz, vp, vp, sz]
    L.cudppSuffixArray.argtypes = [sz, vp, vp, sz]
    L.glcCompressBatch.argtypes = [sz, vp, vp, vp, vp, sz, vp, vp, sz, sz, sz]
    L.glcBwtBatch.argtypes = [sz, vp, vp, vp, sz, sz]
    L.glcMtfBatch.argtypes = [sz, vp, vp, sz, sz]
    L.glcDecompressBatch.argtypes = [sz, vp, vp, vp, sz, vp, sz, vp, sz, sz]
    L.glcCompressBatchCompact.argtypes = [sz, vp, vp, vp, vp, sz, vp, vp, sz, vp, vp, sz, sz]
    L.glcDecompressBatchCompact.argtypes = [sz, vp, vp, vp, sz, vp, sz, vp, vp, sz, sz]
    L.glcPlanSetStream.argtypes = [sz, vp]
    L.glcPlanSynchronize.argtypes = [sz]
    L.glcPlanSetPipelining.argtypes = [sz, C.c_int]
    L.glcHuffmanEncodeBatch.argtypes = [sz, vp, vp, vp, sz, vp, vp, sz, sz, sz]
    L.glcPlanSetSorter.argtypes = [sz, C.c_int]
    L.glcPlanLastSortStats.argtypes = [sz, C.POINTER(C.c_uint)]
    L.glcPlanLastSortStatsEx.argtypes = [sz, C.POINTER(C.c_uint)]
    L.glcPlanLastSortRetries.argtypes = [sz, C.POINTER(C.c_uint)]
    L.glcPlanLastSortResumed.argtypes = [sz, C.POINTER(C.c_uint)]
    L.glcPlanLastSortPeriodic.argtypes = [sz, C.POINTER(C.c_uint)]
    L.glcPlanLastSortSkipped.argtypes = [sz, C.POINTER(C.c_uint)]
    L.glcPlanDebugSortFlags.argtypes = [sz, C.POINTER(C.c_uint), C.POINTER(C.c_uint), sz]
    L.glcPlanDebugBucketFill.argtypes = [sz, sz, C.POINTER(C.c_uint)]
    L.glcPlanEnableTiming.argtypes = [sz, C.c_int]
    L.glcPlanLastTiming.argtypes = [sz, C.POINTER(C.c_float)]
    L.glcPlanKernelProfile.argtypes = [sz, C.POINTER(C.c_double)]
    L.glcPlanKernelProfileEx.argtypes = [sz, C.c_int, C.c_char_p, sz, C.POINTER(C.c_double)]
    L.glcPlanKernelProfileLost.argtypes = [sz, C.POINTER(C.c_ulonglong)]
    L.glcCompactStreams.argtypes = [sz, vp, sz, vp, sz, vp, vp]
    L.glcExpandStreams.argtypes = [sz, vp, vp, sz, vp, sz, vp]
    for name in CUDPP_SYMBOLS:
        getattr(L, name).restype = C.c_int
    L.glcProbeStreamRead.argtypes = [vp, sz, C.c_int, C.POINTER(C.c_float), vp]
    L.glcProbeStreamRead.restype = C.c_int
    L.glcGenZipfPhilox.argtypes = [vp, sz, C.c_ulonglong, C.c_uint, vp, vp]
    L.glcGenZipfPhilox.restype = C.c_int
    L.glcGenFloatPhilox.argtypes = [vp, sz, C.c_ulonglong, C.c_uint, vp]
    L.glcGenFloatPhilox.restype = C.c_int
    # CULZSS
    if hasattr(L, "compression_kernel_wrapper"):
        L.compression_kernel_wrapper.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_int, vp, vp]
        L.compression_kernel_wrapper.restype = C.c_int
        L.aftercompression_wrapper.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_int)]
        L.aftercompression_wrapper.restype = C.c_int
        L.decompression_kernel_wrapper.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
        L.decompression_kernel_wrapper.restype = C.c_int
        L.onestream_finish_GPU.argtypes = [C.c_int]
        L.onestream_finish_GPU.restype = C.c_int
        for nm in ("initGPUmem", "initCPUmem", "deinitGPUmem"):
            getattr(L, nm).argtypes = [C.c_int]
            getattr(L, nm).restype = vp
        for nm in ("deleteGPUmem", "deleteCPUmem", "dedeleteGPUmem"):
            getattr(L, nm).argtypes = [vp]
            getattr(L, nm).restype = None
        for nm in ("initGPU", "resetGPU", "deleteGPUStreams", "signalExitThreads", "deinitGPU"):
            getattr(L, nm).argtypes = []
            getattr(L, nm).restype = None
        L.streams_in_GPU.restype = C.c_int
        L.culzss_compress.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_int)]
        L.culzss_compress.restype = C.c_int
        L.culzss_decompress.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_int)]
        L.culzss_decompress.restype = C.c_int
        L.glcLzssEncodeDevice.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
        L.glcLzssEncodeDevice.restype = C.c_int
        L.glcLzssDecodeDevice.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
        L.glcLzssDecodeDevice.restype = C.c_int
        L.glcLzssPackStride.argtypes = [C.c_int]
        L.glcLzssPackStride.restype = C.c_ulonglong
        L.glcLzssWorkBytes.argtypes = [C.c_int, C.c_int]
        L.glcLzssWorkBytes.restype = C.c_ulonglong
        ull = C.c_ulonglong
        L.culzss_container_bound.argtypes = [ull]
        L.culzss_container_bound.restype = ull
        L.culzss_container_compress.argtypes = [vp, ull, vp, ull, C.POINTER(ull)]
        L.culzss_container_compress.restype = C.c_int
        L.culzss_container_decompress.argtypes = [vp, ull, vp, ull, C.POINTER(ull)]
        L.culzss_container_decompress.restype = C.c_int
        L.culzss_compress_file.argtypes = [C.c_char_p, C.c_char_p]
        L.culzss_compress_file.restype = C.c_int
        L.culzss_decompress_file.argtypes = [C.c_char_p, C.c_char_p]
        L.culzss_decompress_file.restype = C.c_int
        L.glcLzssEnableProfile.argtypes = [C.c_int]
        L.glcLzssEnableProfile.restype = C.c_int
        L.glcLzssKernelProfile.argtypes = [C.c_int, C.c_char_p, sz, C.POINTER(C.c_double)]
        L.glcLzssKernelProfile.restype = C.c_int
        L.glcLzssLastKernelMs.argtypes = []
        L.glcLzssLastKernelMs.restype = C.c_float
    if hasattr(L, "glcHdDecodeDevice"):
        L.glcHdBuildTable.argtypes = [vp, vp, vp]
        L.glcHdBuildTable.restype = C.c_int
        L.glcHdEncodeHost.argtypes = [vp, sz, vp, vp, vp, sz]
        L.glcHdEncodeHost.restype = sz
        L.glcHdWorkBytes.argtypes = [sz]
        L.glcHdWorkBytes.restype = sz
        L.glcHdDecodeDevice.argtypes = [vp, sz, vp, vp, vp, sz, vp, vp]
        L.glcHdDecodeDevice.restype = C.c_int
        L.glcHdDecodeDeviceTable.argtypes = [vp, sz, vp, vp, sz, vp, vp]
        L.glcHdDecodeDeviceTable.restype = C.c_int
        L.glcHdDecodeDeviceTableOnDevice.argtypes = [vp, sz, vp, vp, sz, vp, vp]
        L.glcHdDecodeDeviceTableOnDevice.restype = C.c_int
        L.glcHdEnableProfile.argtypes = [C.c_int]
        L.glcHdEnableProfile.restype = C.c_int
        L.glcHdKernelProfile.argtypes = [C.c_int, C.c_char_p, sz, C.POINTER(C.c_double)]
        L.glcHdKernelProfile.restype = C.c_int
    if hasattr(L, "glcGatherStreams"):                             # include/glc_exchange.h
        ullp = C.POINTER(C.c_ulonglong)
        L.glcCommGetUniqueId.argtypes = [vp]
        L.glcCommInitRank.argtypes = [C.POINTER(vp), C.c_int, vp, C.c_int]
        L.glcCommAdopt.argtypes = [C.POINTER(vp), vp]
        L.glcCommDestroy.argtypes = [vp]
        L.glcCommInfo.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.glcPackRecords.argtypes = [vp, vp, vp, sz, vp, sz, sz, vp, vp]
        L.glcUnpackRecords.argtypes = [vp, sz, sz, vp, vp, vp, sz, vp, vp]
        L.glcGatherCounts.argtypes = [vp, C.c_ulonglong, C.c_ulonglong, vp, ullp, vp]
        L.glcGatherCountsBegin.argtypes = [vp, C.c_ulonglong, C.c_ulonglong, vp, C.POINTER(C.c_int), vp]
        L.glcGatherCountsReady.argtypes = [vp, C.c_int, C.POINTER(C.c_int)]
        L.glcGatherCountsEnd.argtypes = [vp, C.c_int, ullp]
        L.glcGatherStreams.argtypes = [vp, C.c_int, vp, vp, sz, ullp, vp, vp, vp]
        L.glcScatterStreams.argtypes = [vp, C.c_int, vp, vp, sz, ullp, vp, vp, vp]
        for name in EXCHANGE_SYMBOLS:
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


class HdError(RuntimeError):
    pass


class CudppError(RuntimeError):
    def __init__(self, fn, code):
        super().__init__("%s returned CUDPPResult %d" % (fn, code))
        self.code = code


def _chk(fn, code):
    if code != CUDPP_SUCCESS:
        raise CudppError(fn, code)


def config(algorithm, datatype=CUDPP_UCHAR, options=0, op=CUDPP_ADD):
    return CUDPPConfiguration(algorithm, op, datatype, options, 0)


class Cudpp:
    """cudppCreate/cudppDestroy pair."""

    def __init__(self):
        h = C.c_size_t(0)
        _chk("cudppCreate", lib().cudppCreate(C.byref(h)))
        self.handle = h.value

    def close(self):
        if self.handle:
            lib().cudppDestroy(self.handle)
            self.handle = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Plan:
    """cudppPlan/cudppDestroyPlan pair (rows = blocks per batched call)."""

    def __init__(self, cudpp, algorithm, n, rows=1, datatype=CUDPP_UCHAR, options=0):
        self.algorithm, self.n, self.rows = algorithm, n, rows
        h = C.c_size_t(0)
        rc = lib().cudppPlan(cudpp.handle, C.byref(h), config(algorithm, datatype, options), n, rows, 0)
        _chk("cudppPlan", rc)
        self.handle = h.value

    def close(self):
        if self.handle:
            lib().cudppDestroyPlan(self.handle)
            self.handle = 0

    def set_stream(self, stream_ptr):
        _chk("glcPlanSetStream", lib().glcPlanSetStream(self.handle, stream_ptr))

    def synchronize(self):
        _chk("glcPlanSynchronize", lib().glcPlanSynchronize(self.handle))

    def set_pipelining(self, on=True):
        """overlap the suffix sort of a batch with the MTF + Huffman stages of the previous one"""
        _chk("glcPlanSetPipelining", lib().glcPlanSetPipelining(self.handle, 1 if on else 0))

    def set_sorter(self, mode):
        """0 bucket sorter + general sorter for flagged blocks (default), 1 general only, 2 general, prefix doubling only"""
        _chk("glcPlanSetSorter", lib().glcPlanSetSorter(self.handle, int(mode)))

    def last_flagged_blocks(self):
        a = C.c_uint(0)
        _chk("glcPlanLastSortStats", lib().glcPlanLastSortStats(self.handle, C.byref(a)))
        return a.value

    def last_sort_stats(self):
        """(blocks the bucket sorter gave up on, blocks the sample sorter gave up on too) of the last call"""
        a = (C.c_uint * 2)()
        _chk("glcPlanLastSortStatsEx", lib().glcPlanLastSortStatsEx(self.handle, a))
        return a[0], a[1]

    def last_sort_retries(self):
        """blocks of the last call the sample sorter finished in its second attempt (other samples)"""
        a = (C.c_uint * 1)()
        _chk("glcPlanLastSortRetries", lib().glcPlanLastSortRetries(self.handle, a))
        return a[0]

    def last_sort_resumed(self):
        """blocks of the last call whose prefix doubling resumed from the sample sorter's order (deep repeats inside them)"""
        a = (C.c_uint * 1)()
        _chk("glcPlanLastSortResumed", lib().glcPlanLastSortResumed(self.handle, a))
        return a[0]

    def last_sort_periodic(self):
        """blocks of the last call finished by the periodic tier (one periodic stretch: closed form over the sorted rotations)"""
        a = (C.c_uint * 1)()
        _chk("glcPlanLastSortPeriodic", lib().glcPlanLastSortPeriodic(self.handle, a))
        return a[0]

    def last_sort_skipped(self):
        """(skipped, streak): did the plan's last call go straight to the sample sorter, and the streak of all-text-like calls"""
        a = (C.c_uint * 2)()
        _chk("glcPlanLastSortSkipped", lib().glcPlanLastSortSkipped(self.handle, a))
        return bool(a[0]), int(a[1])

    def enable_timing(self, mode=1):
        """0 off, 1 stage events, 3 stage events + dominant-kernel events"""
        _chk("glcPlanEnableTiming", lib().glcPlanEnableTiming(self.handle, int(mode)))

    def kernel_profile(self):
        """dominant kernel {ms, launches, units}; resets the accumulators"""
        a = (C.c_double * 3)()
        _chk("glcPlanKernelProfile", lib().glcPlanKernelProfile(self.handle, a))
        return dict(ms=a[0], launches=int(a[1]), units=a[2])

    def kernel_profiles(self):
        """{kernel name: dict(ms, launches, units)} of every profiled kernel (call after synchronize())"""
        out, i = {}, 0
        name = C.create_string_buffer(96)
        a = (C.c_double * 3)()
        while lib().glcPlanKernelProfileEx(self.handle, i, name, 96, a) == CUDPP_SUCCESS:
            if a[1] > 0:
                out[name.value.decode()] = dict(ms=a[0], launches=int(a[1]), units=a[2])
            i += 1
        return out

    def last_timing(self):
        a = (C.c_float * 4)()
        _chk("glcPlanLastTiming", lib().glcPlanLastTiming(self.handle, a))
        return list(a)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def compressed_stride_words(n):
    return (HUFF_MAX_WORDS + 1) * ((n + HUFF_BLOCK - 1) // HUFF_BLOCK)


# ---------------------------------------------------------------------------
# torch-tensor conveniences (device pointers in, device tensors out)
# ---------------------------------------------------------------------------
def compress_batch(plan, d_in, n, nblk):
    """d_in: uint8 cuda tensor of nblk*n bytes.  Returns dict of cuda tensors."""
    import torch
    dev = d_in.device
    nsub = (n + HUFF_BLOCK - 1) // HUFF_BLOCK
    stride = compressed_stride_words(n)
    out = dict(
        bwt_index=torch.empty(nblk, dtype=torch.int32, device=dev),
        hist=torch.empty(nblk * 256, dtype=torch.int32, device=dev),
        offsets=torch.empty(nblk * nsub, dtype=torch.int32, device=dev),
        size=torch.empty(nblk, dtype=torch.int32, device=dev),
        words=torch.empty(nblk * stride, dtype=torch.int32, device=dev),
        stride=stride, nsub=nsub,
    )
    rc = lib().glcCompressBatch(plan.handle, d_in.data_ptr(), out["bwt_index"].data_ptr(),
                                out["hist"].data_ptr(), out["offsets"].data_ptr(), nsub,
                                out["size"].data_ptr(), out["words"].data_ptr(), stride, n, nblk)
    _chk("glcCompressBatch", rc)
    return out


def huffman_encode_batch(plan, d_sym, n, nblk):
    """stand-alone Huffman stage on symbols (uint8 cuda tensor of nblk*n bytes); same outputs as compress_batch"""
    import torch
    dev = d_sym.device
    nsub = (n + HUFF_BLOCK - 1) // HUFF_BLOCK
    stride = compressed_stride_words(n)
    out = dict(hist=torch.empty(nblk * 256, dtype=torch.int32, device=dev),
               offsets=torch.empty(nblk * nsub, dtype=torch.int32, device=dev),
               size=torch.empty(nblk, dtype=torch.int32, device=dev),
               words=torch.empty(nblk * stride, dtype=torch.int32, device=dev), stride=stride, nsub=nsub)
    rc = lib().glcHuffmanEncodeBatch(plan.handle, d_sym.data_ptr(), out["hist"].data_ptr(), out["offsets"].data_ptr(),
                                     nsub, out["size"].data_ptr(), out["words"].data_ptr(), stride, n, nblk)
    _chk("glcHuffmanEncodeBatch", rc)
    return out


def compress_batch_into(plan, d_in, n, nblk, out):
    rc = lib().glcCompressBatch(plan.handle, d_in.data_ptr(), out["bwt_index"].data_ptr(),
                                out["hist"].data_ptr(), out["offsets"].data_ptr(), out["nsub"],
                                out["size"].data_ptr(), out["words"].data_ptr(), out["stride"], n, nblk)
    _chk("glcCompressBatch", rc)


def compress_batch_compact(plan, d_in, n, nblk, words=None, block_off=None, start=None, meta=None, first=0):
    """glcCompressBatchCompact.  words: int32 cuda tensor that receives the streams back to back (default: room for the
    worst case); block_off: int64 tensor of >= first + nblk + 1 entries; start: data_ptr of a device u64 the first block
    begins at (None = 0); meta: dict with bwt_index / hist / offsets / size tensors for >= first + nblk blocks (default:
    new ones); first: index of this batch's first block in those arrays.  Returns the dict (+ words, block_off)."""
    import torch
    dev = d_in.device
    nsub = (n + HUFF_BLOCK - 1) // HUFF_BLOCK
    if meta is None:
        meta = dict(bwt_index=torch.empty(first + nblk, dtype=torch.int32, device=dev),
                    hist=torch.empty((first + nblk) * 256, dtype=torch.int32, device=dev),
                    offsets=torch.empty((first + nblk) * nsub, dtype=torch.int32, device=dev),
                    size=torch.empty(first + nblk, dtype=torch.int32, device=dev), nsub=nsub)
    if words is None:
        words = torch.empty(nblk * compressed_stride_words(n), dtype=torch.int32, device=dev)
    if block_off is None:
        block_off = torch.empty(first + nblk + 1, dtype=torch.int64, device=dev)
    rc = lib().glcCompressBatchCompact(plan.handle, d_in.data_ptr(), meta["bwt_index"].data_ptr() + 4 * first,
                                       meta["hist"].data_ptr() + 1024 * first, meta["offsets"].data_ptr() + 4 * nsub * first, nsub,
                                       meta["size"].data_ptr() + 4 * first, words.data_ptr(), words.numel(),
                                       block_off.data_ptr() + 8 * first, start, n, nblk)
    _chk("glcCompressBatchCompact", rc)
    out = dict(meta)
    out.update(words=words, block_off=block_off, nsub=nsub)
    return out


def decompress_batch_compact(plan, comp, n, nblk, first=0, d_out=None):
    """glcDecompressBatchCompact on the dict compress_batch_compact returns (blocks first .. first + nblk)"""
    import torch
    nsub = comp["nsub"]
    if d_out is None:
        d_out = torch.empty(nblk * n, dtype=torch.uint8, device=comp["words"].device)
    rc = lib().glcDecompressBatchCompact(plan.handle, comp["bwt_index"].data_ptr() + 4 * first, comp["hist"].data_ptr() + 1024 * first,
                                         comp["offsets"].data_ptr() + 4 * nsub * first, nsub, comp["words"].data_ptr(),
                                         comp["words"].numel(), comp["block_off"].data_ptr() + 8 * first, d_out.data_ptr(), n, nblk)
    _chk("glcDecompressBatchCompact", rc)
    return d_out


def decompress_batch(plan, comp, n, nblk, d_out=None):
    import torch
    if d_out is None:
        d_out = torch.empty(nblk * n, dtype=torch.uint8, device=comp["words"].device)
    rc = lib().glcDecompressBatch(plan.handle, comp["bwt_index"].data_ptr(), comp["hist"].data_ptr(),
                                  comp["offsets"].data_ptr(), comp["nsub"], comp["words"].data_ptr(),
                                  comp["stride"], d_out.data_ptr(), n, nblk)
    _chk("glcDecompressBatch", rc)
    return d_out


# ---------------------------------------------------------------------------
# CUHD-shaped stream (include/glc_hd.h)
# ---------------------------------------------------------------------------
def hd_build_table(hist256):
    """hist256: 256 counts.  Returns (lens u8[256], codes u16[256])."""
    import numpy as np
    h = np.ascontiguousarray(hist256, dtype=np.uint64)
    lens = np.zeros(256, dtype=np.uint8)
    codes = np.zeros(256, dtype=np.uint16)
    if lib().glcHdBuildTable(h.ctypes.data, lens.ctypes.data, codes.ctypes.data) == 0:
        raise HdError("glcHdBuildTable failed (empty histogram?)")
    return lens, codes


def hd_encode_host(data_u8, lens, codes):
    """Host encoder: returns the stream as uint32 units (incl. the zero pad unit)."""
    import numpy as np
    a = np.ascontiguousarray(data_u8, dtype=np.uint8)
    cap = (a.size * GLC_HD_MAX_LEN + 31) // 32 + 2
    out = np.zeros(cap, dtype=np.uint32)
    n = lib().glcHdEncodeHost(a.ctypes.data, a.size, lens.ctypes.data, codes.ctypes.data, out.ctypes.data, cap)
    if n == 0:
        raise HdError("glcHdEncodeHost failed (symbol without a code?)")
    return out[:n].copy()


def hd_decode_device(d_units, lens, codes, nsym, stream=None):
    """d_units: int32/uint32 cuda tensor.  Returns a uint8 cuda tensor of nsym bytes."""
    import torch
    nunits = d_units.numel()
    work = torch.empty(lib().glcHdWorkBytes(nunits), dtype=torch.uint8, device=d_units.device)
    out = torch.empty(max(1, nsym), dtype=torch.uint8, device=d_units.device)
    ok = lib().glcHdDecodeDevice(d_units.data_ptr(), nunits, lens.ctypes.data, codes.ctypes.data,
                                 out.data_ptr(), nsym, work.data_ptr(), stream)
    if not ok:
        raise HdError("glcHdDecodeDevice failed")
    return out[:nsym]
