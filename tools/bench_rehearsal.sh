# Rehearsal of bench.py's multi-rank flows on a ONE-GPU box (two ranks sharing GPU 0, gloo control
# plane, host transport standing in for RCCL): (1) the sharded run succeeds; (2) rank 1's sharded
# child fails (injected) -> the in-process device-group fallback produces the line, labelled;
# (3) --mode group directly; (4) the bare launch `python bench.py --gpus 2`.   gpurun -- 'bash tools/bench_rehearsal.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/rehearsal; mkdir -p $O
run() { # name, extra env..., then args
  name=$1; shift
  timeout -k 10 300 env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT \
    bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --device 0 --ntrain 8192 --stall-timeout 120 $MODE > $O/$name.json 2> $O/$name.err
  echo "$name rc=$? $(python -c "
import json,sys
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('value', d['value'], '|', d['config']['parallelism'], '| fallback_from:', d.get('fallback_from'), '| check', (d.get('shard_check') or {}).get('ok'))
except Exception as e: print('no json', e)")"
}
PORT=29611 MODE="" run ok A=1 || true
PORT=29621 MODE="" run injected_failure GPX_BENCH_INJECT=fail:1 || true
PORT=29631 MODE="--mode group" run group_direct A=1 || true
# (4) bare launch, no torch.distributed.run in front: bench.py starts the launcher itself
timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --device 0 --ntrain 8192 --stall-timeout 120 > $O/bare_launch.json 2> $O/bare_launch.err
echo "bare_launch rc=$? $(python -c "
import json
try:
    d=json.loads(open('$O/bare_launch.json').read().strip().splitlines()[-1]); print('n_gpus', d['n_gpus'], 'value', d['value'], '|', d['config']['parallelism'], '| check', (d.get('shard_check') or {}).get('ok'))
except Exception as e: print('no json', e)")"
