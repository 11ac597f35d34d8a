#!/usr/bin/env python3
"""Static instruction census of the library's kernels, by issue class (runs in the build container: hipcc only).

    python tools/isa_census.py [tag]      -> profiles/<tag>_isa_census.json, profiles/isa_census.json, profiles/<tag>_isa_census.md

Every .hip source is compiled to gfx950 assembly (`hipcc --cuda-device-only -S`) and the instructions between a kernel's
label and its end are counted: VALU of the 2-cycle class (plain add / sub / and / or / xor / shift right / mov / f32 add /
fma with VGPR or literal operands: profiles/r03a_valu_rate.md), VALU of the 4-cycle class (everything else: DPP, SDWA,
compares, carries, shifts left, min / max, multiplies, bit-field and three-operand integer ops, v_readlane, an SGPR
operand on an otherwise fast instruction), SALU, LDS, VMEM.  A static count is not a dynamic one, but the kernels' hot
parts are straight-line code executed by every wave, so `valu_fast_frac` is the weight bench.py uses to turn a measured
SQ_INSTS_VALU into issue cycles: cycles = N * (f * 2.1 + (1 - f) * 4.0)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpu-lossless-compression_amd", "csrc")
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32",
        "v_add_f32", "v_fma_f32", "v_mov_b64"}


def classify(line):
    m = re.match(r"\s+([a-z_0-9]+)\s*(.*)", line)
    if not m:
        return None
    op, rest = m.group(1), m.group(2).split(";")[0]
    if op.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_sleep", "s_setprio", "s_code_end")):
        return "other"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        base = re.sub(r"_(e32|e64)$", "", op)
        if op.endswith(("_dpp", "_sdwa")) or "row_" in rest or "quad_perm" in rest or "dst_sel" in rest:
            return "valu_slow"
        if base in FAST:
            ops = [o.strip() for o in rest.split(",")]
            if any(re.match(r"^(s\d+|s\[\d+:\d+\]|vcc|exec|m0)", o) for o in ops[1:]):
                return "valu_slow"                             # an SGPR source puts a fast instruction in the slow class
            return "valu_fast"
        return "valu_slow"
    return None


def census_of(asm):
    out, cur, counts = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_ZN3glc\S+):", line)
        if m:
            cur, counts = m.group(1), dict(valu_fast=0, valu_slow=0, salu=0, lds=0, vmem=0, other=0)
            continue
        if cur and line.startswith(".Lfunc_end"):
            out[cur] = counts
            cur = None
            continue
        if cur:
            c = classify(line)
            if c:
                counts[c] += 1
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    res = {}
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", "-",
                            os.path.join(CSRC, f)], capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(r.stderr[-2000:])
        c = census_of(r.stdout)
        dm = demangle(list(c))
        for k, v in c.items():
            short = re.sub(r"^(void )?glc::", "", dm[k]).split("(")[0]
            valu = v["valu_fast"] + v["valu_slow"]
            v["valu_fast_frac"] = round(v["valu_fast"] / valu, 3) if valu else None
            v["source"] = f
            res[short] = v
    doc = {"collected": tag, "what": __doc__.split("\n\n")[2].replace("\n", " "), "fast_class": sorted(FAST),
           "cycles_per_instruction": {"fast": 2.1, "slow": 4.0}, "kernels": res}
    for path in (os.path.join(ROOT, "profiles", "%s_isa_census.json" % tag), os.path.join(ROOT, "profiles", "isa_census.json")):
        json.dump(doc, open(path, "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", "%s_isa_census.md" % tag), "w") as fo:
        fo.write("# static instruction census by issue class (tools/isa_census.py, tag %s)\n\n" % tag)
        fo.write("| kernel | VALU 2-cycle class | VALU 4-cycle class | fast share | SALU | LDS | VMEM |\n|---|---|---|---|---|---|---|\n")
        for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["valu_fast"] + kv[1]["valu_slow"])):
            fo.write("| `%s` | %d | %d | %s | %d | %d | %d |\n" % (k, v["valu_fast"], v["valu_slow"], v["valu_fast_frac"], v["salu"], v["lds"], v["vmem"]))
    print("wrote profiles/%s_isa_census.{json,md}: %d kernels" % (tag, len(res)))


if __name__ == "__main__":
    main()
