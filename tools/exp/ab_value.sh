# usage: ab_value.sh tag...   -- bench.py `value` (4 GiB, main leg only) with variants/libglc_<tag>.so ("main" = the in-tree library); env passes through
for v in "$@"; do
  unset GLC_LIB
  if [ $v != main ]; then export GLC_LIB=$PWD/gpu-lossless-compression_amd/variants/libglc_$v.so; fi
  python bench.py --gib 4 --steps 3 --main-only --no-cpu-baseline --no-verify --details /tmp/ab_$v.json > /tmp/ab_$v.line 2>/tmp/ab_$v.err || tail -3 /tmp/ab_$v.err
  python - <<PY
import json
try:
    j=json.load(open("/tmp/ab_$v.line")); print("$v", j["value"], j.get("value_no_stage_overlap_GBps"), j["kernel_ms_per_launch"])
except Exception as e: print("$v failed", e)
PY
done
