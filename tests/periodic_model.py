"""CPU model of the periodic tier's closed form (csrc/bwt_periodic.hip) -- test infrastructure, never imported by the product.

The same layout the kernels use, in a dozen lines of Python: smallest period p of the block's beginning, first break e, tail
t = n - e, L = the multiple of p covering max(p, t), explicit zone Z L wide; the text of representatives
U = T[0 .. Z L + 2 p + 1) | 0xFF | T[e - Z L .. n) | zeros, its suffix array (naive), every rotation expanded into its class's
chain.  Compared with a naive suffix array of T it shows whether the LAYOUT is right, independent of any GPU: with Z = 1
(round 5's form) 'abaa' * k + 'baab' comes out wrong, with Z = 2 (PER_Z in csrc/glc_internal.h) nothing does.

ADVERSARIAL: (period, cut, tail) triples on which the Z = 1 layout is wrong, found by random search with this model
(tools/exp/find_periodic_adversaries.py); block = period * k + period[:cut] + tail.  The GPU tests build 64 KiB+ blocks of them."""

PER_Z = 2


def naive_sa(T):
    return sorted(range(len(T)), key=lambda i: T[i:])


def span(p, t):
    m = max(p, t)
    return p * ((m + p - 1) // p)


def closed_form_sa(T, Z=PER_Z, probe=4):
    """the tier's suffix array of T, or None where the tier would not take the block (probe: bytes the candidate test compares;
    16 in the kernel)"""
    n = len(T)
    emax = 0
    for p in range(1, n):
        if T[:probe] != T[p:p + probe]:
            continue
        e = next((i for i in range(p, n) if T[i] != T[i - p]), n)
        t = n - e
        L = Z * span(p, t)
        if e > emax and e >= L + 2 * p + 1:
            break
        emax = max(emax, e)
    else:
        return None
    la = L + 2 * p + 1
    U = T[:la] + b"\xff" + T[e - L:n] + bytes(8)
    x0, lb = la + 1, L + t
    X = e == n or T[e] < T[e - p]
    sa = []
    for q in naive_sa(U):
        if q < p:
            chain = list(range(q, e - L, p))
            sa += chain[::-1] if X else chain
        elif x0 <= q < x0 + lb:
            sa.append(e - L + (q - x0))
    return sa if len(sa) == n else None


ADVERSARIAL = [
    (b"abaa", 0, b"baab"),
    (b"\x00\x00\x00\x00\x00\x00\x01", 4, b"\x01\x00\x00\x00\x00\x01"), (b"\x00\x00\x00\x00\x00\x01", 2, b"\x00\x01\x00\x00\x00\x00\x01"),
    (b"\x00\x00\x00\x01", 2, b"\x01\x00\x00\x01"), (b"\x00\x00\x00\x01\x00", 2, b"\x01\x00\x00\x00\x01"),
    (b"\x00\x00\x00\x01\x00\x00\x01", 1, b"\x00\x01\x00\x00\x01"), (b"\x00\x00\x00\x01\x00\x01\x00", 0, b"\x00\x00\x01\x00\x01\x00\x00\x00\x01"),
    (b"\x00\x00\x00\x01\x01", 2, b"\x01\x01\x00\x00\x01"), (b"\x00\x00\x01\x00", 0, b"\x00\x01\x00\x00\x01"),
    (b"\x00\x00\x01\x00", 1, b"\x01\x00\x00\x01"), (b"\x00\x00\x01\x00\x00\x00\x01", 4, b"\x00\x01\x00\x00\x01\x00\x00\x00"),
    (b"\x00\x00\x01\x00\x01", 4, b"\x00\x01\x00\x01"), (b"\x00\x00\x01\x00\x01\x00", 3, b"\x00\x00\x00\x01\x00\x01"),
    (b"\x00\x01\x00\x00", 3, b"\x00\x01\x00\x00\x01"), (b"\x00\x01\x00\x00\x01\x00\x00", 0, b"\x01\x00\x00\x01"),
    (b"\x00\x01\x00\x00\x01\x01\x00", 1, b"\x01\x01\x00\x00\x01\x00\x01"), (b"\x00\x01\x00\x01\x00", 2, b"\x00\x00\x01\x00\x01"),
    (b"\x00\x01\x00\x01\x01", 2, b"\x01\x00\x01\x01"), (b"\x00\x01\x00\x01\x01", 2, b"\x01\x00\x01\x01\x00"),
    (b"\x00\x01\x01\x00\x01", 2, b"\x01\x00\x01\x01\x00\x01\x01\x00"), (b"\x00\x01\x01\x01\x00\x01\x01", 3, b"\x00\x01\x01\x01\x00\x00"),
    (b"\x00\x01\x01\x01\x01\x01", 3, b"\x01\x01\x00\x01\x01\x01\x01\x01"), (b"\x00\x01\x01\x01\x01\x01", 4, b"\x00\x01\x01\x01\x01\x01"),
    (b"\x00\x02\x02\x02\x02\x02\x02", 4, b"\x02\x00\x02\x02\x02\x02\x01"), (b"\x01\x00\x00\x01\x00", 3, b"\x01\x00\x00\x01\x00\x01"),
    (b"\x01\x00\x00\x01\x00\x01\x01", 0, b"\x00\x00\x01\x00\x01\x01\x01"), (b"\x01\x00\x00\x01\x01", 0, b"\x00\x00\x01\x01\x01"),
    (b"\x01\x00\x01\x00\x01", 3, b"\x01\x00\x01\x01"), (b"\x01\x00\x01\x00\x01\x01", 3, b"\x01\x01\x00\x01\x01"),
    (b"\x01\x00\x01\x01", 0, b"\x00\x01\x01\x01"), (b"\x01\x00\x01\x01\x00", 2, b"\x01\x01\x00\x01\x01\x00\x01\x01"),
    (b"\x01\x00\x01\x01\x00\x01\x01", 6, b"\x01\x00\x01\x01\x00\x01\x01\x01"), (b"\x01\x00\x01\x01\x01\x00", 1, b"\x01\x01\x00\x01\x01\x00"),
    (b"\x01\x00\x01\x01\x01\x00", 4, b"\x01\x00\x01\x01\x01\x00\x01\x01"), (b"\x01\x00\x01\x01\x01\x01", 5, b"\x00\x01\x01\x01\x01\x00"),
    (b"\x01\x01\x00\x00\x00", 4, b"\x01\x01\x00\x00\x01"), (b"\x01\x01\x00\x00\x01", 1, b"\x00\x00\x01\x01\x01"),
    (b"\x01\x01\x00\x00\x01\x01\x01", 5, b"\x01\x01\x00\x00\x01\x01\x01\x01\x01"), (b"\x01\x01\x00\x01", 1, b"\x00\x01\x01\x01"),
    (b"\x01\x01\x00\x01\x00", 2, b"\x00\x01\x01\x00\x01\x01"), (b"\x01\x01\x00\x01\x00\x01", 3, b"\x01\x01\x01\x00\x01\x01"),
    (b"\x01\x01\x01\x00", 1, b"\x01\x00\x01\x01\x01"), (b"\x01\x01\x01\x00", 2, b"\x00\x01\x01\x01"),
    (b"\x01\x01\x01\x00\x00", 1, b"\x01\x00\x00\x01\x01\x01"), (b"\x01\x01\x01\x00\x00", 2, b"\x00\x00\x01\x01\x01"),
    (b"\x01\x01\x01\x00\x01", 2, b"\x00\x01\x01\x01\x01"), (b"\x01\x01\x01\x00\x01\x00\x01", 2, b"\x00\x01\x00\x01\x01\x01\x01"),
    (b"\x02\x00\x02\x02", 3, b"\x02\x00\x02\x02\x02"),
]


def adversarial_block(per, cut, tail, n):
    """a block of exactly n bytes: per * k + per[:cut] + tail"""
    body = n - len(tail) - cut
    k, r = divmod(body, len(per))
    if r:                                  # keep the phase at the break: drop whole leading symbols of the first period instead
        return (per * (k + 1))[len(per) - r:] + per[:cut] + tail
    return per * k + per[:cut] + tail
