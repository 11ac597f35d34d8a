#!/usr/bin/env python3
"""gpurun_out/<tag>_pmc_insts.json (tools/exp/pmc_insts.sh) + gpurun_out/<tag>_valu_rate.json (tools/probes/valu_rate_probe)
-> profiles/<tag>_pmc_insts.{json,md}, profiles/<tag>_valu_rate.{json,md} and profiles/pmc_insts.json (the file bench.py
reads for `valu_issue_frac`: wave64 instructions per 64 input bytes of every glc:: kernel + the measured issue rates).
usage: make_pmc_insts.py <tag>"""
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
UNITS = float(256 << 20)          # the counters were collected on launches over 256 blocks of 1 MiB

SHORT = {"k_fs_part": "k_fs_part2<", "k_fs_sort": "k_fs_sort_bwt(", "k_fs_hist": "k_fs_hist(", "k_fs_ties": "k_fs_ties(",
         "k_mtf_encode": "k_mtf_encode<", "k_huff_pack": "k_huff_pack(", "k_huff_build": "k_huff_build<",
         "k_mtf_chunk_lists": "k_mtf_chunk_lists(", "k_mtf_scan_lists": "k_mtf_scan_lists<",
         "k_ibwt_walk": "k_ibwt_walk<", "k_imtf_pos": "k_imtf_pos_deque(", "k_dec_huff_lanes": "k_dec_huff_lanes(",
         "k_ibwt_lf": "k_ibwt_lf(", "k_ibwt_emit": "k_ibwt_emit(", "k_ibwt_hist": "k_ibwt_hist(",
         "k_imtf_apply": "k_imtf_apply(", "k_imtf_scan": "k_imtf_scan(", "k_ibwt_rank": "k_ibwt_rank(",
         "k_dec_prepare": "k_dec_prepare("}


def rate_summary(vr):
    """issue rate of every probed instruction at 8 waves per SIMD: wave64 instructions per SIMD per cycle"""
    # the chip clock during a launch = the longest wave's s_memtime ticks / the launch's event time; here the rate is
    # quoted per second and per SIMD, which needs no clock: Ginst/s over the whole chip / (4 SIMDs x CUs)
    simds = 4 * vr["cus"]
    out = {}
    for name, o in vr["ops"].items():
        best = max(o[w]["ginst_per_s_chip"] for w in o)
        out[name] = {"ginst_per_s_chip": best, "ginst_per_s_per_simd": round(best / simds, 4),
                     "cycles_per_inst_one_wave": o["w1"]["cycles_per_inst_per_wave"]}
    return out


def main():
    tag = sys.argv[1]
    out = os.path.join(ROOT, "gpurun_out")
    vr_path = os.path.join(out, tag + "_valu_rate.json")
    rates = None
    if os.path.exists(vr_path):
        vr = json.load(open(vr_path))
        shutil.copy(vr_path, os.path.join(HERE, tag + "_valu_rate.json"))
        rates = rate_summary(vr)
        fast = sorted(k for k, v in rates.items() if v["ginst_per_s_per_simd"] > 0.8 and k.startswith("v_"))
        slow = sorted(k for k, v in rates.items() if 0.4 < v["ginst_per_s_per_simd"] <= 0.8 and k.startswith("v_"))
        with open(os.path.join(HERE, tag + "_valu_rate.md"), "w") as f:
            f.write("# tools/probes/valu_rate_probe on MI355X (%s, %d CUs; tag %s)\n\n" % (vr["device"], vr["cus"], tag))
            f.write("64 instructions of one kind per loop trip on 8 independent registers, 1 / 2 / 4 / 8 waves per SIMD on every SIMD.\n"
                    "`G/s` = wave64 instructions per second over the whole chip (hipEvents), best of the four occupancies;\n"
                    "`per SIMD` = G/s / %d SIMDs (instructions per ns per SIMD); `1 wave` = shader cycles (s_memtime) per\n"
                    "instruction as ONE wave alone on its SIMD sees it.  The chip held ~2.35-2.40 GHz during these launches\n"
                    "(s_memtime ticks of the longest wave / event time), so 0.59 per ns per SIMD = one instruction per 4.0 cycles\n"
                    "and 1.03-1.13 = one per 2.1-2.3 cycles.\n\n" % (4 * vr["cus"]))
            f.write("| instruction | G/s chip | per SIMD per ns | cycles, 1 wave |\n|---|---|---|---|\n")
            for k, v in rates.items():
                f.write("| `%s` | %.0f | %.3f | %.2f |\n" % (k, v["ginst_per_s_chip"], v["ginst_per_s_per_simd"], v["cycles_per_inst_one_wave"]))
            f.write("\n**Two VALU classes.** ~2 cycles per wave64 instruction (needs two waves per SIMD: one wave alone issues every ~4.1): "
                    + ", ".join("`%s`" % k for k in fast) + ".\n\n~4 cycles: " + ", ".join("`%s`" % k for k in slow) + ".\n\n"
                    "`v_cndmask_b32` (VOP2, implicit vcc) with a vcc NOT produced by a preceding VALU compare runs at one per ~22 cycles; "
                    "behind `v_cmp` (the way compiled code uses it) the pair takes 2 x 4.  SALU: one instruction per cycle per CU "
                    "(0.59 per ns per SIMD-equivalent); `s_nop` is free at the issue stage.  LDS: a `ds_read_b32/b64/u8` wave "
                    "instruction per ~8 cycles per CU-quarter (0.28 per ns per SIMD), `b128`, writes and returning atomics half "
                    "of that, `ds_bpermute` a third.\n")
    # the issue-rate probe is not rerun with every counter collection: cite the newest summary there is
    vr_md = sorted(f for f in os.listdir(HERE) if f.endswith("_valu_rate.md"))
    vr_tag = vr_md[-1][:-len("_valu_rate.md")] if vr_md else tag
    pj = os.path.join(out, tag + "_pmc_insts.json")
    if not os.path.exists(pj):
        print("no", pj)
        return
    raw = json.load(open(pj))
    shutil.copy(pj, os.path.join(HERE, tag + "_pmc_insts.json"))
    per64 = {}
    rows = []
    for short, sub in SHORT.items():
        for name, k in raw["kernels"].items():
            if sub in name:
                p = k["per_launch"]
                us = sum(k["avg_us_under_counters"].values()) / len(k["avg_us_under_counters"])
                e = {c: round(p[c] / (UNITS / 64.0), 2) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
                                                                  "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM") if c in p}
                e["avg_us_under_counters_256_blocks"] = round(us, 1)
                e["launches"] = k["launches"]
                if "SQ_LDS_BANK_CONFLICT" in p and p.get("SQ_LDS_IDX_ACTIVE"):
                    e["lds_bank_conflict_frac"] = round(p["SQ_LDS_BANK_CONFLICT"] / p["SQ_LDS_IDX_ACTIVE"], 3)
                    e["SQ_LDS_IDX_ACTIVE"] = round(p["SQ_LDS_IDX_ACTIVE"] / (UNITS / 64.0), 2)     # LDS-busy cycles (summed over CUs) per 64 input bytes
                if "SQ_WAVE_CYCLES" in p and p.get("SQ_WAVES"):
                    e["wave_cycles_per_wave"] = round(p["SQ_WAVE_CYCLES"] / p["SQ_WAVES"] * 4)   # SQ counts in quad-cycles... see note
                per64[short] = e
                rows.append((short, e))
                break
    res = {"collected": tag, "command": raw["command"],
           "unit": "wave64 instructions per 64 input bytes (counter sum of a launch over 256 blocks of 1 MiB / 2^22)",
           "per_64_bytes": per64,
           "issue_rate": {"source": "profiles/%s_valu_rate.json (tools/probes/valu_rate_probe.hip)" % vr_tag,
                          "slow_class_cycles_per_inst": 4.0, "fast_class_cycles_per_inst": 2.1,
                          "slow_class": "DPP, SDWA, compares, carry ops, shifts left, min/max, multiplies, bit-field / permute ops, 3-operand integer ops, SGPR-operand forms",
                          "fast_class": "v_add_u32, v_sub_u32, v_and/or/xor_b32, v_lshrrev_b32, v_mov_b32, v_add_f32, v_fma_f32 (VGPR operands)",
                          "salu": "1 instruction per cycle per CU", "clock_GHz": 2.4}}
    json.dump(res, open(os.path.join(HERE, "pmc_insts.json"), "w"), indent=1)
    with open(os.path.join(HERE, tag + "_pmc_insts.md"), "w") as f:
        f.write("# per-kernel instruction counters (rocprofv3 --pmc, three passes, --kernel-trace only; MI355X; tag %s)\n\n`%s`\n\n" % (tag, raw["command"]))
        f.write("wave64 instructions per 64 input bytes; `us` = average launch duration under the counters (256 blocks of 1 MiB per launch);\n"
                "`VALU x 4 cyc` = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x launch cycles at 2.4 GHz): the share of the VALU issue slots if every\n"
                "instruction were of the 4-cycle class (profiles/%s_valu_rate.md); `SALU` = SQ_INSTS_SALU / (256 CUs x launch cycles).\n\n" % vr_tag)
        f.write("| kernel | VALU | SALU | LDS | VMEM rd | VMEM wr | us | VALU x 4 cyc | SALU pipe | LDS bank-conflict share |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for short, e in rows:
            cyc = e["avg_us_under_counters_256_blocks"] * 1e-6 * 2.4e9
            n64 = UNITS / 64.0
            f.write("| `%s` | %.1f | %.1f | %.1f | %.2f | %.2f | %.1f | %.2f | %.2f | %s |\n" % (
                short, e.get("SQ_INSTS_VALU", 0), e.get("SQ_INSTS_SALU", 0), e.get("SQ_INSTS_LDS", 0), e.get("SQ_INSTS_VMEM_RD", 0),
                e.get("SQ_INSTS_VMEM_WR", 0), e["avg_us_under_counters_256_blocks"],
                e.get("SQ_INSTS_VALU", 0) * n64 * 4 / (1024 * cyc), e.get("SQ_INSTS_SALU", 0) * n64 / (256 * cyc),
                e.get("lds_bank_conflict_frac", "-")))
    print(open(os.path.join(HERE, tag + "_pmc_insts.md")).read())


if __name__ == "__main__":
    main()
