#!/usr/bin/env python3
"""Per kernel: the sequence of global loads (L) and `s_waitcnt vmcnt` (W) in the gfx950 ISA.
(LW)xN = N loads each waited for before the next is issued = N serial memory latencies.
usage: loadwait.py file.hip [...]"""
import re, subprocess, sys, tempfile, os
for src in sys.argv[1:]:
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    "-o", out, src], check=True, capture_output=True)
    txt = open(out).read()
    parts = re.split(r'\n(_Z\w+):[^\n]*\n', txt)
    for i in range(1, len(parts), 2):
        name = parts[i]; body = parts[i + 1].split('s_endpgm')[0]
        seq = []
        for ln in body.split('\n'):
            if 'global_load' in ln or 'buffer_load' in ln: seq.append('L')
            elif 's_waitcnt' in ln and 'vmcnt' in ln: seq.append('W')
        sq = ''.join(seq)
        if sq:
            m = re.search(r'\d+([a-z_]\w*?)(E|I)', name)
            print('%-28s %s' % (name[7:35], re.sub(r'(LW){3,}', lambda m: '(LW)x%d' % (len(m.group(0)) // 2), sq)[:110]))
