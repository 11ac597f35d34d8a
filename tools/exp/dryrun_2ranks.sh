#!/bin/bash
# the N = 2 code path of bench.py on a one-GPU box (both ranks on device 0, gloo): the normal line, then the line when
# one rank's exchange leg fails or hangs (rank 0 prints without the gather fields, at the latest after --gather-timeout)
#   bash tools/exp/dryrun_2ranks.sh
export GLC_BENCH_ONE_DEVICE=1
R="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
timeout 600 $R --master-port 29517 bench.py --gpus 2 --gib 0.5 --rows 256 --steps 1 --warmup 1 --no-cpu-baseline 2>/tmp/d1.err | grep '{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ok  value', d['value'], 'with_gather', d.get('value_with_gather'), 'gather_ms', d.get('gather_ms'), d['gather_to_rank0'].get('root_decodes_gathered_blocks'))"
GLC_BENCH_BREAK_EXCHANGE=1 timeout 600 $R --master-port 29518 bench.py --gpus 2 --gib 0.5 --rows 256 --steps 1 --warmup 1 --no-cpu-baseline --gather-timeout 20 2>/tmp/d2.err | grep '{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('broken value', d['value'], 'with_gather', d.get('value_with_gather'), d['gather_to_rank0'])"
GLC_BENCH_BREAK_EXCHANGE=hang1 timeout 600 $R --master-port 29519 bench.py --gpus 2 --gib 0.5 --rows 256 --steps 1 --warmup 1 --no-cpu-baseline --gather-timeout 20 2>/tmp/d3.err | grep '{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hung   value', d['value'], 'with_gather', d.get('value_with_gather'), d['gather_to_rank0'])"
echo "exit codes: ${PIPESTATUS[@]}"; tail -2 /tmp/d2.err
