"""Seeded synthetic inputs shared by tests, smoke() and bench.py."""
import numpy as np


def zipf_bytes(n, seed=0x5eed0002, s=1.0):
    """i.i.d. bytes, byte value = rank-th symbol of Zipf(s) over 256 symbols (config 2)."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    p = 1.0 / np.arange(1, 257) ** s
    p /= p.sum()
    cdf = np.cumsum(p)
    u = rng.random(n)
    return np.minimum(np.searchsorted(cdf, u), 255).astype(np.uint8)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11) on arrays of 32-bit counters held in uint64; returns the four output words.
    Checked against the Random123 known-answer vectors in tests/test_cpu_oracle.py."""
    c = [np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3)]
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = c[0] * np.uint64(0xD2511F53)
        p1 = c[2] * np.uint64(0xCD9E8D57)
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & m32, (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & m32]
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c


def zipf_thresholds(s=1.0):
    """thr[k] = floor(2^32 * P(symbol <= k)), k = 0..254, for Zipf(s) over 256 symbols (identity permutation)"""
    p = 1.0 / np.arange(1, 257, dtype=np.float64) ** s
    cdf = np.cumsum(p / p.sum())
    return np.minimum(np.floor(cdf[:255] * 4294967296.0), 4294967295.0).astype(np.uint64).astype(np.uint32)


def zipf_philox_bytes(first_byte, n, seed=0x5eed0002, s=1.0):
    """bytes [first_byte, first_byte + n) of the Zipf(s) stream defined by Philox4x32-10 with
    key = seed and counter = byte index / 4 -- byte i = number of thresholds <= word (i & 3) of that Philox output.  The
    device twin is glcGenZipfPhilox (csrc/probe.hip): any block of the 4 GiB workload is reproducible on either side."""
    assert first_byte % 4 == 0 and n % 4 == 0
    ctr = np.arange(first_byte // 4, (first_byte + n) // 4, dtype=np.uint64)
    w = philox4x32_10(ctr & np.uint64(0xFFFFFFFF), ctr >> np.uint64(32), np.zeros_like(ctr), np.zeros_like(ctr), seed, 0)
    u = np.stack(w, axis=1).reshape(-1).astype(np.uint32)
    return np.searchsorted(zipf_thresholds(s), u, side="right").astype(np.uint8)


def float_philox_bytes(first_byte, n, seed=0x5eed0004):
    """bytes [first_byte, first_byte + n) of a float32 ~ N(0, 1) stream as raw little-endian bytes.
    Value i = ((sum of the four Philox4x32-10 words of counter i, each >> 10) * 2^-22 - 2) * sqrt(3) -- Irwin-Hall(4) scaled
    to unit variance; integer sum, one exact conversion, exact scaling and subtraction, one correctly rounded multiply, so
    glcGenFloatPhilox (csrc/probe.hip) produces the same bits on the device."""
    assert first_byte % 4 == 0 and n % 4 == 0
    ctr = np.arange(first_byte // 4, (first_byte + n) // 4, dtype=np.uint64)
    w = philox4x32_10(ctr & np.uint64(0xFFFFFFFF), ctr >> np.uint64(32), np.zeros_like(ctr), np.zeros_like(ctr), seed, 0)
    s = sum((x >> np.uint64(10)) for x in w).astype(np.float32)                # < 2^24: exact
    z = (s * np.float32(2.384185791015625e-07) - np.float32(2.0)) * np.float32(1.7320508075688772)
    return z.astype(np.float32).view(np.uint8).copy()


def float_bytes(n, seed=0x5eed0004):
    """float32 ~ N(0,1) as little-endian bytes (config 4)."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    return rng.standard_normal((n + 3) // 4, dtype=np.float32).view(np.uint8)[:n].copy()


_WORDS = None


def text_bytes(n, seed=0x5eed0001):
    """'enwik-style' text: order-1 word model over a fixed vocabulary with XML-ish tags (config 1)."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    global _WORDS
    if _WORDS is None:
        vr = np.random.Generator(np.random.Philox(key=12345))
        letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
        lp = 1.0 / np.arange(1, 27)
        lp /= lp.sum()
        _WORDS = []
        for _ in range(4096):
            ln = int(vr.integers(2, 10))
            _WORDS.append(bytes(vr.choice(letters, size=ln, p=lp)))
    out = bytearray()
    nw = len(_WORDS)
    zp = 1.0 / np.arange(1, nw + 1)
    zp /= zp.sum()
    prev = 0
    while len(out) < n:
        picks = rng.choice(nw, size=4096, p=zp)
        mix = rng.random(4096)
        for k in range(4096):
            w = (prev * 31 + 7) % nw if mix[k] < 0.35 else int(picks[k])   # order-1 dependence
            prev = w
            r = mix[k]
            if r > 0.985:
                out += b"<" + _WORDS[w] + b">"
            elif r > 0.97:
                out += b"</" + _WORDS[w] + b">\n"
            elif r > 0.93:
                out += _WORDS[w] + b". "
            else:
                out += _WORDS[w] + b" "
        if len(out) >= n:
            break
    return np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()


def text_bytes_fast(n, seed=0x5eed0001):
    """the model of text_bytes (same vocabulary, same order-1 dependence w -> (31 w + 7) mod 4096 with probability 0.35, same
    tag / full-stop / space frequencies), vectorised: ~100 MB/s instead of ~1.  Not the same bytes as text_bytes (the
    random numbers are drawn in a different order); bench.py uses it for batches of DISTINCT text blocks."""
    text_bytes(16)                                             # builds _WORDS
    rng = np.random.Generator(np.random.Philox(key=seed))
    nw = len(_WORDS)
    wl = np.array([len(w) for w in _WORDS], dtype=np.int64)
    wo = np.concatenate([[0], np.cumsum(wl)])
    wb = np.frombuffer(b"".join(_WORDS), dtype=np.uint8)
    zp = 1.0 / np.arange(1, nw + 1)
    cdf = np.cumsum(zp / zp.sum())
    m = int(n / 5.5) + 4096                                     # tokens: a little more than enough (mean token ~6.8 bytes)
    picks = np.minimum(np.searchsorted(cdf, rng.random(m)), nw - 1).astype(np.int64)
    mix = rng.random(m)
    dep = mix < 0.35
    dep[0] = False
    # w[k] = f^d(picks[k - d]) where d = number of dependent tokens in a row ending at k
    idx = np.arange(m)
    last = np.maximum.accumulate(np.where(~dep, idx, 0))
    d = idx - last
    w = picks[last]
    for step in range(1, int(d.max()) + 1):
        w = np.where(d >= step, (w * 31 + 7) % nw, w)
    kind = np.where(mix > 0.985, 3, np.where(mix > 0.97, 2, np.where(mix > 0.93, 1, 0)))      # <w> | </w>\n | "w. " | "w "
    pre = np.array([0, 0, 2, 1])[kind]
    post = np.array([1, 2, 2, 1])[kind]
    tl = pre + wl[w] + post
    off = np.concatenate([[0], np.cumsum(tl)])
    total = int(off[-1])
    out = np.zeros(total + 8, dtype=np.uint8)
    # word bytes
    rep = np.repeat(np.arange(m), wl[w])
    within = np.arange(rep.size) - np.repeat(np.concatenate([[0], np.cumsum(wl[w])])[:-1], wl[w])
    out[off[:-1][rep] + pre[rep] + within] = wb[wo[w][rep] + within]
    # punctuation
    s0 = off[:-1]; e0 = off[:-1] + pre + wl[w]
    k1, k2, k3, k0 = kind == 1, kind == 2, kind == 3, kind == 0
    out[e0[k0]] = 32
    out[e0[k1]] = 46; out[e0[k1] + 1] = 32
    out[s0[k2]] = 60; out[s0[k2] + 1] = 47; out[e0[k2]] = 62; out[e0[k2] + 1] = 10
    out[s0[k3]] = 60; out[e0[k3]] = 62
    assert total >= n
    return out[:n].copy()


def log_bytes(n, seed=0x5eed0003):
    """log-style ASCII lines (config 3)."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    levels = [b"INFO", b"WARN", b"DEBUG", b"ERROR"]
    svcs = [b"auth", b"db", b"cache", b"api", b"queue", b"sched"]
    msgs = [b"request completed", b"connection reset by peer", b"cache miss for key", b"retrying operation",
            b"user login ok", b"slow query detected", b"heartbeat", b"flushed buffers"]
    out = bytearray()
    t = 0
    while len(out) < n:
        r = rng.integers(0, 1 << 30, size=8)
        t += int(r[0] % 997)
        ms = t % 1000; s = (t // 1000) % 60; mi = (t // 60000) % 60; h = (t // 3600000) % 24
        out += b"2026-09-%02dT%02d:%02d:%02d.%03dZ host-%02d svc-%s[%d]: %s %s k=%d v=%d\n" % (
            1 + (t // 86400000) % 28, h, mi, s, ms, r[1] % 16, svcs[r[2] % 6], r[3] % 32768,
            levels[r[4] % 4], msgs[r[5] % 8], r[6] % 1000, r[7] % 100000)
    return np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()


def symbols_from_hist(hist, a=0x9E3779B1, c=12345):
    """n = sum(hist) symbols (n must be a power of two) with exactly the given counts, spread by the affine
    permutation i -> (a * i + c) mod n (a odd): version-independent, so a committed histogram names its bytes."""
    hist = np.asarray(hist, dtype=np.int64)
    n = int(hist.sum())
    assert n & (n - 1) == 0 and a & 1
    ordered = np.repeat(np.arange(hist.size, dtype=np.uint8), hist)
    idx = (np.arange(n, dtype=np.uint64) * np.uint64(a) + np.uint64(c)) & np.uint64(n - 1)
    out = np.empty(n, dtype=np.uint8)
    out[idx.astype(np.int64)] = ordered
    return out


def glibc_rand_bytes(n, mod, seed=95835):
    """the reference's test inputs: srand(95835); (rand() % mod) + 1  (test_compress.cpp:439-441,552-556,687-692;
    test_sa.cpp:124-126) -- glibc's rand(), which is what the reference's testrig links."""
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(seed)
    rand = libc.rand
    return np.fromiter(((rand() % mod) + 1 for _ in range(n)), dtype=np.uint8, count=n)


def lzss_synthetic_candidates():
    """candidate streams (c0 = 1 literal | match length, c1) chosen for the token walk, not produced by a match search:
    walks from different starts that never fall into step (one length everywhere), jumps of exactly / just under /
    just over the 64 positions a lane owns, the longest jumps, and random mixtures"""
    rng = np.random.default_rng(77)
    n = 16 * 4096
    out = {}

    def stream(lengths):
        # as EncodeKernel leaves them: no match crosses the end of its packet (gpu_compress.cu:313-317), and a length
        # below 3 is a literal
        room = 4096 - (np.arange(n) % 4096)
        lengths = np.minimum(lengths.astype(np.int64), room)
        lengths[lengths <= 2] = 1
        c = np.empty(2 * n, dtype=np.uint8)
        c[0::2] = lengths.astype(np.uint8)
        c[1::2] = rng.integers(0, 256, n, dtype=np.uint8)
        return c

    for k in (3, 4, 5, 7, 63, 64, 65, 127):
        out["all_%d" % k] = stream(np.full(n, k, dtype=np.uint8))
    out["alternate_3_4"] = stream(np.where(np.arange(n) % 2 == 0, 3, 4).astype(np.uint8))
    out["by_residue"] = stream((3 + (np.arange(n) % 5)).astype(np.uint8))               # a different chain per start
    mix = rng.integers(3, 128, n).astype(np.uint8)
    mix[rng.random(n) < 0.3] = 1
    out["random_mix"] = stream(mix)
    short = rng.integers(3, 6, n).astype(np.uint8)
    short[rng.random(n) < 0.1] = 1
    out["short_mix"] = stream(short)
    seg = np.full(n, 3, dtype=np.uint8)
    seg[(np.arange(n) % 64) == 61] = 127                                                 # jumps over whole segments
    out["jump_over_segments"] = stream(seg)
    out["all_literal"] = stream(np.full(n, 1, dtype=np.uint8))                           # 9/8 of the buffer: store raw
    return n, out


def lzss_gold_inputs():
    """inputs of tests/golden/ref_lzss_gold.npz (CULZSS rows a13/a14): name -> bytes of one buffer (a multiple of 4096).
    Shared by the generator (tests/golden/make_lzss_gold.py) and the tests that read the fixture."""
    MiB = 1 << 20
    rng = np.random.default_rng(11)
    tail = np.random.default_rng(20260928).integers(0, 256, MiB, dtype=np.uint8)
    c = {
        "log_1m": log_bytes(MiB),
        "text_1m": text_bytes(MiB),
        "zeros_1m": np.zeros(MiB, dtype=np.uint8),
        "zipf_1m": zipf_bytes(MiB),                                   # store raw
        "float_256k": float_bytes(262144),                            # store raw
        "run_118685": np.concatenate([np.full(118685, 65, dtype=np.uint8), tail[: MiB - 118685]]),   # packed > buffer
        "run_119175": np.concatenate([np.full(119175, 65, dtype=np.uint8), tail[: MiB - 119175]]),   # packed == buffer
        "zeros_64k": np.zeros(65536, dtype=np.uint8),
        "spaces_then_text": np.concatenate([np.full(8192, 0x20, dtype=np.uint8), text_bytes(57344, seed=5)]),
        "period3_8k": np.tile(np.array([1, 2, 3], dtype=np.uint8), 2731)[:8192].copy(),
        "caret_tail_4k": np.concatenate([log_bytes(3968, seed=9), np.full(128, ord("^"), dtype=np.uint8)]),
        "repeat_across_last_chunk": np.tile(text_bytes(96, seed=3), 43)[:4096].copy(),
        "one_packet_random": rng.integers(0, 256, 4096, dtype=np.uint8),
    }
    for k in (1, 2, 3, 4):
        c["log_%dpkt" % k] = log_bytes(4096 * k, seed=40 + k)
    return c


def findmatch_gold_inputs():
    """inputs of tests/golden/ref_findmatch_gold.npz (CULZSS row a11, the match search): the buffers of
    lzss_gold_inputs() plus packets aimed at FindMatch's corners -- runs of 128 and more (length clamp to 127), a
    4-symbol alphabet (many equally long runs: the first one found must win), all spaces (matches against the ring's
    initial fill), periods 1 / 2 / 5, a run that starts inside the last 128-byte chunk (shortened scan + clamp),
    and the '^' fill meeting '^' data.  Shared by tests/golden/make_findmatch_gold.py and the tests."""
    c = dict(lzss_gold_inputs())
    rng = np.random.default_rng(4242)
    c["sym4_16k"] = rng.integers(0, 4, 16384, dtype=np.uint8) + 65
    c["spaces_8k"] = np.full(8192, 0x20, dtype=np.uint8)
    c["runs_ge128_8k"] = np.repeat(rng.integers(0, 256, 40, dtype=np.uint8), rng.integers(100, 400, 40))[:8192].copy()
    c["period1_2_5_12k"] = np.concatenate([np.full(4096, 7, dtype=np.uint8), np.tile(np.array([8, 9], dtype=np.uint8), 2048),
                                           np.tile(np.array([1, 2, 3, 4, 5], dtype=np.uint8), 820)[:4096]])
    last = log_bytes(4096, seed=77).copy()
    last[3968 + 40:] = last[3968 - 60:3968 - 60 + 88]                  # a repeat that begins 40 bytes into the last chunk
    c["repeat_inside_last_chunk"] = last
    car = text_bytes(4096, seed=8).copy()
    car[3900:3968] = ord("^")                                             # '^' data just ahead of the '^'-filled slot
    car[4000:] = ord("^")
    c["caret_data_4k"] = car
    c["random_64k"] = rng.integers(0, 256, 65536, dtype=np.uint8)
    return c
