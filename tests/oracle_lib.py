"""ctypes access to the CPU oracle (oracle/libglc_oracle.so) and, when it was
built, the reference's own gold suffix-array routine (oracle/_ref/libsagold.so,
compiled unmodified from cudpp-inpar/apps/cudpp_testrig/sa_gold.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product path never touches it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libglc_oracle.so")
SAGOLD_SO = os.path.join(ORACLE_DIR, "_ref", "libsagold.so")

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)


def build_oracle():
    """Compile the oracle (and _ref when /root/reference exists)."""
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


def _load():
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    lib = C.CDLL(ORACLE_SO)
    lib.orc_suffix_array.argtypes = [_u8p, C.c_uint32, _u32p]
    lib.orc_bwt.argtypes = [_u8p, C.c_uint32, _u8p, _i32p]
    lib.orc_mtf.argtypes = [_u8p, C.c_uint32, _u8p]
    lib.orc_imtf.argtypes = [_u8p, C.c_uint32, _u8p]
    lib.orc_ibwt.argtypes = [_u8p, C.c_uint32, C.c_int32, _u8p]
    lib.orc_huff_codes.argtypes = [_u32p, _u32p, _u8p]
    lib.orc_huff_codes.restype = C.c_int
    lib.orc_huff_encode.argtypes = [_u8p, C.c_uint32, _u32p, _u32p, _u32p, _u32p, C.c_uint32]
    lib.orc_huff_encode.restype = C.c_int
    lib.orc_compress.argtypes = [_u8p, C.c_uint32, _i32p, _u32p, _u32p, _u32p, _u32p, C.c_uint32]
    lib.orc_compress.restype = C.c_int
    lib.orc_decompress.argtypes = [C.c_int32, _u32p, _u32p, _u32p, C.c_uint32, _u8p]
    lib.orc_decompress.restype = C.c_int
    lib.orc_lzss_candidates.argtypes = [_u8p, C.c_int, _u8p]
    lib.orc_lzss_pack.argtypes = [_u8p, C.c_int, _u8p, C.POINTER(C.c_int)]
    lib.orc_lzss_pack.restype = C.c_int
    lib.orc_lzss_decode.argtypes = [_u8p, C.c_int, _u8p, C.POINTER(C.c_int)]
    lib.orc_lzss_decode.restype = C.c_int
    lib.orc_lzss_container_compress.argtypes = [_u8p, C.c_uint64, _u8p, C.POINTER(C.c_uint64)]
    lib.orc_lzss_container_compress.restype = C.c_int
    lib.orc_lzss_container_decompress.argtypes = [_u8p, C.c_uint64, _u8p, C.POINTER(C.c_uint64)]
    lib.orc_lzss_container_decompress.restype = C.c_int
    lib.orc_compress_many.argtypes = [_u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    lib.orc_compress_many.restype = C.c_int
    lib.orc_hd_decode.argtypes = [_u32p, C.c_uint64, _u8p, C.POINTER(C.c_uint16), _u8p, C.c_uint64]
    lib.orc_hd_decode.restype = C.c_int
    lib.orc_huffman_cost.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    lib.orc_huffman_cost.restype = C.c_uint64
    lib.orc_crc32.argtypes = [_u8p, C.c_size_t]
    lib.orc_crc32.restype = C.c_uint32
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p8(a):
    return a.ctypes.data_as(_u8p)


def _p32(a):
    return a.ctypes.data_as(_u32p)


def _as_u8(data):
    a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else data,
                             dtype=np.uint8)
    return a


HUFF_BLOCK = 4096
MAX_BLOCK_WORDS = 1536


def compressed_capacity_words(n):
    nblk = (n + HUFF_BLOCK - 1) // HUFF_BLOCK
    return max(1, nblk) * (MAX_BLOCK_WORDS + 1)


def suffix_array(data):
    a = _as_u8(data)
    sa = np.zeros(max(1, a.size), dtype=np.uint32)
    lib().orc_suffix_array(_p8(a), a.size, _p32(sa))
    return sa[: a.size]


def bwt(data):
    a = _as_u8(data)
    out = np.zeros(max(1, a.size), dtype=np.uint8)
    idx = C.c_int32(-1)
    lib().orc_bwt(_p8(a), a.size, _p8(out), C.byref(idx))
    return out[: a.size], idx.value


def ibwt(L, index):
    a = _as_u8(L)
    out = np.zeros(max(1, a.size), dtype=np.uint8)
    lib().orc_ibwt(_p8(a), a.size, index, _p8(out))
    return out[: a.size]


def mtf(data):
    a = _as_u8(data)
    out = np.zeros(max(1, a.size), dtype=np.uint8)
    lib().orc_mtf(_p8(a), a.size, _p8(out))
    return out[: a.size]


def imtf(data):
    a = _as_u8(data)
    out = np.zeros(max(1, a.size), dtype=np.uint8)
    lib().orc_imtf(_p8(a), a.size, _p8(out))
    return out[: a.size]


def huff_codes(hist256):
    h = np.ascontiguousarray(hist256, dtype=np.uint32)
    codes = np.zeros(257, dtype=np.uint32)
    lens = np.zeros(257, dtype=np.uint8)
    n = lib().orc_huff_codes(_p32(h), _p32(codes), _p8(lens))
    return codes, lens, n


def huff_encode(mtf_bytes):
    a = _as_u8(mtf_bytes)
    cap = compressed_capacity_words(a.size)
    nblk = (a.size + HUFF_BLOCK - 1) // HUFF_BLOCK
    hist = np.zeros(256, dtype=np.uint32)
    off = np.zeros(max(1, nblk), dtype=np.uint32)
    size = C.c_uint32(0)
    comp = np.zeros(cap, dtype=np.uint32)
    rc = lib().orc_huff_encode(_p8(a), a.size, _p32(hist), _p32(off), C.byref(size), _p32(comp), cap)
    return dict(rc=rc, hist=hist, offsets=off[:nblk], size=size.value, words=comp[: size.value])


def compress(data):
    """Oracle cudppCompress.  Returns dict(bwt_index, hist, offsets, size, words, rc)."""
    a = _as_u8(data)
    cap = compressed_capacity_words(a.size)
    nblk = (a.size + HUFF_BLOCK - 1) // HUFF_BLOCK
    hist = np.zeros(256, dtype=np.uint32)
    off = np.zeros(max(1, nblk), dtype=np.uint32)
    size = C.c_uint32(0)
    comp = np.zeros(cap, dtype=np.uint32)
    idx = C.c_int32(-1)
    rc = lib().orc_compress(_p8(a), a.size, C.byref(idx), _p32(hist), _p32(off), C.byref(size), _p32(comp), cap)
    return dict(rc=rc, bwt_index=idx.value, hist=hist, offsets=off[:nblk], size=size.value,
                words=comp[: size.value])


def compress_many(blocks, n, nthreads):
    """bench.py cpu_baseline: len(blocks)//n blocks through orc_compress on `nthreads` pthreads."""
    a = _as_u8(blocks)
    tot = C.c_uint64(0)
    lib().orc_compress_many(_p8(a), n, a.size // n, nthreads, C.byref(tot))
    return tot.value


def decompress(bwt_index, hist, offsets, words, n):
    hist = np.ascontiguousarray(hist, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    words = np.ascontiguousarray(words, dtype=np.uint32)
    out = np.zeros(max(1, n), dtype=np.uint8)
    rc = lib().orc_decompress(bwt_index, _p32(hist), _p32(offsets), _p32(words), n, _p8(out))
    if rc != 0:
        raise ValueError("oracle decoder: corrupt stream")
    return out[:n]


def lzss_candidates(data):
    a = _as_u8(data)
    assert a.size % 4096 == 0
    out = np.zeros(2 * a.size, dtype=np.uint8)
    lib().orc_lzss_candidates(_p8(a), a.size, _p8(out))
    return out


def lzss_pack(cand, buf_length):
    c = _as_u8(cand)
    out = np.zeros(buf_length + 64 + 2 * (buf_length // 4096), dtype=np.uint8)
    n = C.c_int(0)
    ok = lib().orc_lzss_pack(_p8(c), buf_length, _p8(out), C.byref(n))
    if not ok:
        return None
    return out[: n.value].copy()


def lzss_decode(packed):
    p = _as_u8(packed)
    orig = int.from_bytes(bytes(p[-6:-2]), "big")
    out = np.zeros(max(1, orig) + 4096, dtype=np.uint8)
    n = C.c_int(0)
    lib().orc_lzss_decode(_p8(p), p.size, _p8(out), C.byref(n))
    return out[: n.value].copy()


def lzss_container_compress(data):
    a = _as_u8(data)
    nb = (a.size + (1 << 20) - 1) >> 20
    out = np.zeros(8 + 4 * nb + nb * ((1 << 20) + 4096), dtype=np.uint8)
    n = C.c_uint64(0)
    if not lib().orc_lzss_container_compress(_p8(a), a.size, _p8(out), C.byref(n)):
        return None
    return out[: n.value].copy()


def lzss_container_decompress(blob):
    a = _as_u8(blob)
    nb = int(a[:4].view(np.uint32)[0])
    out = np.zeros(nb << 20, dtype=np.uint8)
    n = C.c_uint64(0)
    if not lib().orc_lzss_container_decompress(_p8(a), a.size, _p8(out), C.byref(n)):
        return None
    return out[: n.value].copy()


def hd_decode(units, lens, codes, nsym):
    """bit-serial decode of a CUHD-shaped stream (row f3 checker)"""
    u = np.ascontiguousarray(units, dtype=np.uint32)
    l = np.ascontiguousarray(lens, dtype=np.uint8)
    c = np.ascontiguousarray(codes, dtype=np.uint16)
    out = np.zeros(max(1, nsym), dtype=np.uint8)
    rc = lib().orc_hd_decode(_p32(u), u.size, _p8(l), c.ctypes.data_as(C.POINTER(C.c_uint16)), _p8(out), nsym)
    if rc != 0:
        raise ValueError("oracle hd decoder: corrupt stream (%d)" % rc)
    return out[:nsym]


def huffman_cost(hist256):
    h = np.ascontiguousarray(hist256, dtype=np.uint64)
    d = C.c_int(0)
    cost = lib().orc_huffman_cost(h.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(d))
    return cost, d.value


def crc32(data):
    a = _as_u8(data)
    return lib().orc_crc32(_p8(a), a.size)


# --------------------------------------------------------------------------
# the reference's own gold (only where oracle/_ref was built)
# --------------------------------------------------------------------------
def have_ref_gold():
    return os.path.exists(SAGOLD_SO)


_gold = None


def ref_sa_gold(data):
    """computeSaGold (sa_gold.cpp:110-119), the reference's CPU skew SA."""
    global _gold
    if _gold is None:
        _gold = C.CDLL(SAGOLD_SO)
        _gold._Z13computeSaGoldPhPjm.argtypes = [_u8p, _u32p, C.c_size_t]
    a = _as_u8(data).copy()
    ref = np.zeros(a.size + 3, dtype=np.uint32)
    _gold._Z13computeSaGoldPhPjm(_p8(a), _p32(ref), a.size)
    return ref[: a.size]


COMPRESSGOLD_SO = os.path.join(os.path.dirname(SAGOLD_SO), "libcompressgold.so")
_cgold = None


def have_ref_compress_gold():
    return os.path.exists(COMPRESSGOLD_SO)


def _cg():
    """oracle/_ref/libcompressgold.so: the CPU gold of test_compress.cpp (FindMinimumCountTest, huffman_build_tree_cpu,
    computeMtfGold, computeBwtGold, Huffman + inverse-MTF half of computeCompressGold), built by
    oracle/mk_ref_compress_gold.sh from the reference's own lines."""
    global _cgold
    if _cgold is None:
        L = C.CDLL(COMPRESSGOLD_SO)
        ip = C.POINTER(C.c_int)
        L.ref_huffman_tree.argtypes = [_u32p, ip, _u32p, ip, ip, ip, ip, _u32p]
        L.ref_huffman_tree.restype = C.c_int
        L.ref_compress_gold_decode.argtypes = [_u32p, _u32p, _u32p, C.c_size_t, _u8p, _u8p]
        L.ref_compress_gold_decode.restype = C.c_int
        L.ref_mtf_gold.argtypes = [_u8p, _u8p, C.c_uint]
        L.ref_bwt_gold.argtypes = [_u8p, _u8p, C.c_uint]
        L.ref_bwt_gold.restype = C.c_int
        _cgold = L
    return _cgold


def ref_huffman_tree(hist256):
    """huffman_build_tree_cpu on hist (+EOF count 1).  Returns dict(head, nnodes, value, count, level, left, right)."""
    h = np.zeros(257, dtype=np.uint32)
    h[:256] = np.asarray(hist256, dtype=np.uint32)
    arr = {k: np.zeros(513, dtype=np.int32) for k in ("value", "level", "left", "right", "parent")}
    cnt = np.zeros(513, dtype=np.uint32)
    nn = C.c_uint32(0)
    ip = C.POINTER(C.c_int)
    head = _cg().ref_huffman_tree(_p32(h), arr["value"].ctypes.data_as(ip), _p32(cnt), arr["level"].ctypes.data_as(ip),
                                  arr["left"].ctypes.data_as(ip), arr["right"].ctypes.data_as(ip),
                                  arr["parent"].ctypes.data_as(ip), C.byref(nn))
    return dict(head=head, nnodes=nn.value, count=cnt, **arr)


def ref_codes_from_tree(tree):
    """root-to-leaf paths of the reference's tree, left = 0 / right = 1 (the walk of the reference's own decoder,
    test_compress.cpp:258-266).  Returns (codes as python ints [257], lens [257])."""
    codes, lens = [0] * 257, np.zeros(257, dtype=np.uint16)
    stack = [(tree["head"], 0, 0)]
    while stack:
        node, code, ln = stack.pop()
        if tree["value"][node] != -1:                      # COMPOSITE_NODE
            codes[int(tree["value"][node])] = code; lens[int(tree["value"][node])] = ln
            continue
        stack.append((int(tree["left"][node]), code << 1, ln + 1))
        stack.append((int(tree["right"][node]), (code << 1) | 1, ln + 1))
    return codes, lens


def ref_compress_gold_decode(hist256, offsets256, words, n):
    """Huffman decode + inverse MTF of a 1 MiB stream by computeCompressGold's own lines.  Returns (symbols, bytes)."""
    h = np.zeros(257, dtype=np.uint32); h[:256] = np.asarray(hist256, dtype=np.uint32)
    off = np.ascontiguousarray(offsets256, dtype=np.uint32)
    w = np.zeros(len(words) + 4, dtype=np.uint32); w[: len(words)] = words     # the gold reads one word ahead
    sym = np.zeros(n, dtype=np.uint8); out = np.zeros(n, dtype=np.uint8)
    _cg().ref_compress_gold_decode(_p32(h), _p32(off), _p32(w), n, _p8(sym), _p8(out))
    return sym, out


def ref_mtf_gold(data):
    a = _as_u8(data).copy(); out = np.zeros(a.size, dtype=np.uint8)
    _cg().ref_mtf_gold(_p8(out), _p8(a), a.size)
    return out


def ref_bwt_gold(data):
    a = _as_u8(data).copy(); out = np.zeros(a.size, dtype=np.uint8)
    idx = _cg().ref_bwt_gold(_p8(a), _p8(out), a.size)
    return out, idx


# --------------------------------------------------------------------------
# the reference test-input generators (glibc rand, srand(95835))
# test_compress.cpp:439-441,552-556,687-692 ; test_sa.cpp:124-126
# --------------------------------------------------------------------------
def glibc_rand_bytes(n, mod, seed=95835):
    libc = C.CDLL("libc.so.6")
    libc.srand(seed)
    rand = libc.rand
    return np.fromiter(((rand() % mod) + 1 for _ in range(n)), dtype=np.uint8, count=n)


AFTERCOMP_SO = os.path.join(os.path.dirname(SAGOLD_SO), "libaftercomp.so")
_aftercomp = None


def have_ref_aftercomp():
    return os.path.exists(AFTERCOMP_SO)


def ref_aftercomp_lib():
    """oracle/_ref/libaftercomp.so: the reference's own aftercomp + aftercompression_wrapper
    (cuda-lzss-cluster/gpu_compress.cu:462-672), built by oracle/mk_ref_aftercomp.sh from the reference's lines."""
    global _aftercomp
    if _aftercomp is None:
        L = C.CDLL(AFTERCOMP_SO)
        L.aftercompression_wrapper.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        L.aftercompression_wrapper.restype = C.c_int
        _aftercomp = L
    return _aftercomp


FINDMATCH_SO = os.path.join(os.path.dirname(SAGOLD_SO), "libfindmatch.so")
_findmatch = None


class _EncodedString(C.Structure):          # gpu_compress.h:75-79
    _fields_ = [("offset", C.c_int), ("length", C.c_int)]


def have_ref_findmatch():
    return os.path.exists(FINDMATCH_SO)


def ref_findmatch_lib():
    """oracle/_ref/libfindmatch.so: the reference's own FindMatch (cuda-lzss-cluster/gpu_compress.cu:104-168),
    built by oracle/mk_ref_findmatch.sh from the reference's lines."""
    global _findmatch
    if _findmatch is None:
        L = C.CDLL(FINDMATCH_SO)
        L.ref_FindMatch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_FindMatch.restype = _EncodedString
        _findmatch = L
    return _findmatch
