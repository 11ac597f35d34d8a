#!/bin/bash
# same-box bench A/B of library builds: bash tools/exp/ab_lib.sh <kernel-table key> tag tag ...   (libglc_<tag>.so beside the in-tree library)
key=$1; shift
for v in "$@" "$@"; do
  GLC_LIB=/root/repo/gpu-lossless-compression_amd/libglc_$v.so timeout 300 python bench.py --no-cpu-baseline --main-only --no-overlap-pass --no-verify > /tmp/o.json 2>/tmp/o.err
  python - <<PY
import json
s=open("/tmp/o.json").read(); d=json.loads(s[s.index('{"metric"'):])
print("$v", d["value"], "$key", d["kernels"]["$key"]["avg_launch_ms"], "decode", d["decode"]["one_plan_GBps"], d["decode"]["pipelined_plans_GBps"])
PY
done
