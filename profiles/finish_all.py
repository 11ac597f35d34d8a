#!/usr/bin/env python3
"""After `gpurun -- bash profiles/collect_all.sh <tag>`: gpurun_out/<tag>_* -> the tracked summaries under profiles/, all under
the one tag.  usage: python profiles/finish_all.py <tag>
  <tag>_bench.json, <tag>_bench_full.json, <tag>_kernel_stats.md (headline + CULZSS + text256 + log256 + partly_deep),
  <tag>_culzss_bench.json, <tag>_culzss_ring.md, pmc_traffic.json (+ <tag>_pmc_traffic.json), <tag>_pmc_insts.{json,md},
  <tag>_isa_census.{json,md}"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "gpurun_out")


def db_in(d):
    f = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    return f[0] if f else None


def main():
    tag = sys.argv[1]
    run = lambda *a: subprocess.run([sys.executable] + list(a), check=True, cwd=ROOT)
    run(os.path.join(ROOT, "tools", "isa_census.py"), tag)
    run(os.path.join(HERE, "make_pmc_json.py"), tag)
    run(os.path.join(HERE, "make_pmc_insts.py"), tag)
    shutil.copy(os.path.join(HERE, "pmc_traffic.json"), os.path.join(HERE, tag + "_pmc_traffic.json"))
    full = os.path.join(OUT, tag + "_bench_full.json")
    if os.path.exists(full):
        shutil.copy(full, os.path.join(HERE, tag + "_bench_full.json"))
    md = os.path.join(HERE, tag + "_kernel_stats.md")
    for what, cmd in (("text256", "python tools/exp/text_batch.py text256 256 3   (256 DISTINCT 1 MiB text blocks, 4 calls of glcCompressBatch; the torch kernels are the data generator)"),
                      ("log256", "python tools/exp/text_batch.py log256 256 3   (256 DISTINCT 1 MiB log buffers)"),
                      ("pd", "python tools/exp/pd_batch.py 3 all   (bench.py's 64 partly_deep blocks, 3 calls of glcBwtBatch)")):
        db = db_in(os.path.join(OUT, "%s_%s_prof" % (tag, what)))
        if not db:
            print("no database for", what)
            continue
        open(md, "a").write("\n# rocprofv3 --kernel-trace --stats -- %s  (MI355X, %s)\n\n" % (cmd, tag))
        subprocess.run([sys.executable, os.path.join(HERE, "summarize_rocpd.py"), db, md], check=True, capture_output=True)
    cmd = os.path.join(HERE, tag + "_culzss_kernel_stats.md")
    if os.path.exists(cmd):                                    # one file for the tag: the CULZSS table joins the others
        open(md, "a").write("\n" + open(cmd).read())
        os.remove(cmd)
    ring = os.path.join(OUT, tag + "_ring.log")
    if os.path.exists(ring):
        open(os.path.join(HERE, tag + "_culzss_ring.md"), "w").write(
            "# CULZSS host-pointer ABI, tests/c_caller/culzss_ring_bench.c 256 16, three runs on one box (%s)\n\n```\n%s```\n" % (tag, open(ring).read()))
    print("done:", sorted(f for f in os.listdir(HERE) if f.startswith(tag)))


if __name__ == "__main__":
    main()
