"""random search for (period, cut, tail) triples on which the periodic tier's ROUND-5 layout (explicit zone L wide, Z = 1) gives a
wrong suffix array -- how tests/periodic_model.ADVERSARIAL was made.  CPU only."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import periodic_model as M

rng = random.Random(11)
found = set()
while len(found) < 48:
    p, t, A = rng.randint(2, 7), rng.randint(1, 10), rng.randint(2, 3)
    per = bytes(rng.randrange(A) for _ in range(p))
    tail = bytes(rng.randrange(A) for _ in range(t))
    cut = rng.randint(0, p - 1)
    bad = 0
    for k in (10, 19):
        T = per * k + per[:cut] + tail
        s = M.closed_form_sa(T, Z=1)
        bad += s is not None and s != M.naive_sa(T)
    if bad == 2:
        found.add((per, cut, tail))
print(sorted(found))
