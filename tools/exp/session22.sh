#!/bin/bash
# k_fs_sort_bwt: flagged blocks leave on a scalar load.  text256 kernel stats, text-like leg, headline twice, the sorter tests
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bucket_sorter.py tests/test_gpu_sample_sorter.py tests/test_gpu_fuzz.py tests/test_gpu_bench_inputs.py -m gpu -x -q 2>&1 | tail -2
bash tools/exp/kstats.sh python $GRAFT_REPO_ROOT/tools/exp/text_batch.py text256 256 3 | grep -E "k_fs_sort_bwt|k_fs_part2|total kernel"
python tools/exp/text_batch.py text256 256 3 2>/dev/null | tail -3
for i in 1 2; do
timeout 600 python bench.py --steps 4 --warmup 1 --main-only --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j.get('kernel_ms_per_launch'))"
done
