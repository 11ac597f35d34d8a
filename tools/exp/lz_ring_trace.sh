#!/bin/bash
# timeline (kernels + memory copies) of the CULZSS host-pointer ring: bash tools/exp/lz_ring_trace.sh
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/lt; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/lt -o t -- /tmp/ring_bench 32 8 > /tmp/lt.log 2>&1
tail -1 /tmp/lt.log
python - <<PY
import sqlite3, glob
db = glob.glob("/tmp/lt/**/*.db", recursive=True)
if not db: print("no db", open("/tmp/lt.log").read()[-600:])
else:
    c = sqlite3.connect(db[0])
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    ev = []
    for name, s, e in c.execute("select s.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id"):
        ev.append((s, e, "K " + name.split("(")[0][-28:]))
    mt = [t for t in tabs if "memory_copy" in t and not t.startswith("rocpd_info")]
    for t in mt:
        cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
        if "start" in cols and "end" in cols:
            sz = "size" if "size" in cols else ("bytes" if "bytes" in cols else "0")
            nm = "name" if "name" in cols else ("kind" if "kind" in cols else "0")
            for s, e, b, n in c.execute("select start, end, %s, %s from %s" % (sz, nm, t)):
                ev.append((s, e, "C %s %s B" % (n, b)))
    ev.sort()
    m = [(s, e) for s, e, n in ev if "k_lzss_match" in n]
    print("k_lzss_match starts, delta to the previous start (us) / duration (us):")
    print(" ".join("%d/%d" % ((m[i][0] - m[i - 1][0]) / 1e3, (m[i][1] - m[i][0]) / 1e3) for i in range(1, len(m))))
    # a window inside the threaded pass (passes: 3 warm-ups of 8, then seq, ring, threads of nbuf each)
    nb = (len(m) - 24) // 7 if len(m) > 24 else 8
    k0 = 24 + 2 * nb + nb // 2
    t_lo = m[min(k0, len(m) - 1)][0]
    win = [x for x in ev if x[0] >= t_lo][:60]
    t0 = win[0][0]
    for s, e, n in win:
        print("%9.1f us  +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
