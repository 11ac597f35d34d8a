#!/usr/bin/env python3
"""gpurun_out/<tag>_{prof,pmc_FETCH_SIZE,pmc_WRITE_SIZE} -> profiles/<tag>_kernel_stats.md,
profiles/<tag>_bench.json and profiles/pmc_traffic.json (read by bench.py for roofline.traffic).
usage: make_pmc_json.py <tag>"""
import glob
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from summarize_pmc import per_kernel  # noqa: E402

FETCH_CORR, WRITE_CORR = 2.0, 1.0     # calibration: see "calibration" below


def db_in(d):
    f = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    return f[0] if f else None


def main():
    tag = sys.argv[1]
    out = os.path.join(ROOT, "gpurun_out")
    prof = db_in(os.path.join(out, tag + "_prof"))
    if prof:
        md = os.path.join(HERE, tag + "_kernel_stats.md")
        open(md, "w").write("# rocprofv3 --kernel-trace --stats -- python bench.py --gib 4 --steps 1 --warmup 1 "
                            "--no-cpu-baseline --main-only --no-overlap-pass  (MI355X, %s)\n\n" % tag)
        subprocess.run([sys.executable, os.path.join(HERE, "summarize_rocpd.py"), prof, md], check=True,
                       capture_output=True)
    cprof = db_in(os.path.join(out, tag + "_culzss_prof"))
    if cprof:
        md = os.path.join(HERE, tag + "_culzss_kernel_stats.md")
        open(md, "w").write("# rocprofv3 --kernel-trace --stats -- python tools/bench_culzss.py --gib 1  (MI355X, %s; the run "
                            "also times smaller batches, see min/max)\n\n" % tag)
        subprocess.run([sys.executable, os.path.join(HERE, "summarize_rocpd.py"), cprof, md], check=True,
                       capture_output=True)
        cj = os.path.join(out, tag + "_culzss_bench.json")
        if os.path.exists(cj) and os.path.getsize(cj):
            shutil.copy(cj, os.path.join(HERE, tag + "_culzss_bench.json"))
    bj = os.path.join(out, tag + "_bench.json")
    if os.path.exists(bj) and os.path.getsize(bj):
        shutil.copy(bj, os.path.join(HERE, tag + "_bench.json"))
    fdb = db_in(os.path.join(out, tag + "_pmc_FETCH_SIZE"))
    wdb = db_in(os.path.join(out, tag + "_pmc_WRITE_SIZE"))
    if not (fdb and wdb):
        print("no PMC databases for", tag)
        return
    F = {r[0]: r for r in per_kernel(fdb, "FETCH_SIZE")}
    W = {r[0]: r for r in per_kernel(wdb, "WRITE_SIZE")}
    kernels = {}
    for k in sorted(set(F) | set(W)):
        f, w = F.get(k), W.get(k)
        kernels[k[:100]] = {
            "launches": (f or w)[1],
            "avg_fetch_KB_raw": f[2] if f else None, "avg_write_KB_raw": w[2] if w else None,
            "avg_hbm_bytes_per_launch": int(((f[2] if f else 0) * FETCH_CORR + (w[2] if w else 0) * WRITE_CORR) * 1024),
        }
    def find(sub):
        for k, v in kernels.items():
            if sub in k:
                return v
        return None
    o8 = find("k_rs_onesweep<8, false") or find("k_rs_onesweep<8")
    # per-launch HBM bytes of the kernels bench.py profiles live (its `roofline.traffic` reads this table), and the
    # whole encode's HBM bytes per input byte (every glc:: kernel of one 256-block batch / 256 MiB)
    short = {"k_fs_part": "k_fs_part2<", "k_fs_sort": "k_fs_sort_bwt(", "k_fs_hist": "k_fs_hist(", "k_fs_ties": "k_fs_ties(",
             "k_mtf_encode": "k_mtf_encode<", "k_huff_pack": "k_huff_pack(", "k_huff_build": "k_huff_build<",
             "k_mtf_chunk_lists": "k_mtf_chunk_lists(", "k_mtf_scan_lists": "k_mtf_scan_lists<",
             "k_rs_onesweep<8,false>": "k_rs_onesweep<8, false",
             # decoder
             "k_dec_huff": "k_dec_huff_lanes(", "k_dec_prepare": "k_dec_prepare(", "k_imtf_pos": "k_imtf_pos_deque(",
             "k_imtf_scan": "k_imtf_scan(", "k_imtf_apply": "k_imtf_apply(", "k_ibwt_hist": "k_ibwt_hist(",
             "k_rs_scan": "k_rs_scan<9>", "k_ibwt_lf": "k_ibwt_lf(", "k_ibwt_walk": "k_ibwt_walk<", "k_ibwt_rank": "k_ibwt_rank(",
             "k_ibwt_emit": "k_ibwt_emit("}
    per_launch = {}
    for nm, sub in short.items():
        v = find(sub)
        if v:
            per_launch[nm] = v["avg_hbm_bytes_per_launch"]
    enc = ["k_fs_hist", "k_fs_part", "k_fs_sort", "k_fs_ties", "k_mtf_chunk_lists", "k_mtf_scan_lists", "k_mtf_encode",
           "k_huff_build", "k_huff_pack"]
    enc_bytes = sum(per_launch.get(k, 0) for k in enc)
    res = {
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --gib 0.25 --steps 1 "
                   "--warmup 0 --no-cpu-baseline --no-verify --main-only --no-overlap-pass (one pass per counter; MI355X; tag %s)" % tag,
        "units": "rocprofv3 reports KB; bytes = KB*1024 * correction",
        "calibration": {
            "fetch": "a kernel that streams exactly 8 B x 2^28 suffix words (2097152 KB) reported FETCH_SIZE "
                     "1048597.5 KB -> correction x2.0000 (the guide's gfx950 half-count, confirmed for 8-B/lane loads)",
            "write": "k_sa_init_keys writes exactly 2097152 KB; WRITE_SIZE reported 2097152.0 KB -> correction x1.0000",
        },
        "fetch_correction": FETCH_CORR, "write_correction": WRITE_CORR,
        "collected": tag,
        "blocks_per_launch": 256,
        "hbm_bytes_per_launch": per_launch,
        "encode_hbm_bytes_per_input_byte": round(enc_bytes / float(256 << 20), 2) if enc_bytes else None,
        "encode_hbm_bytes_note": "sum over the encode kernels of one 256-block launch each / 256 MiB "
                                 "(BWT: k_fs_hist, k_fs_part, k_fs_sort, k_fs_ties; MTF; Huffman)",
        "decode_hbm_bytes_per_input_byte": round(sum(per_launch.get(k, 0) for k in (
            "k_dec_huff", "k_dec_prepare", "k_imtf_pos", "k_imtf_scan", "k_imtf_apply", "k_ibwt_hist", "k_rs_scan", "k_ibwt_lf",
            "k_ibwt_walk", "k_ibwt_rank", "k_ibwt_emit")) / float(256 << 20), 2),
        "k_rs_onesweep8_hbm_bytes_per_launch": o8["avg_hbm_bytes_per_launch"] if o8 else None,
        "kernels": kernels,
    }
    json.dump(res, open(os.path.join(HERE, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: v["avg_hbm_bytes_per_launch"] for k, v in kernels.items()}, indent=1))


if __name__ == "__main__":
    main()
