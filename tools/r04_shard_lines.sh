# bench.py lines of the sharded path on ONE card: the sharded schedule on one rank (1-rank RCCL communicator), and the
# two-rank rehearsal over the host transport (bare launch), fp64 and mixed
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04_shard_lines; rm -rf $O; mkdir -p $O
timeout -k 10 300 python bench.py --mode shard --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/shard_one_rank.json 2> $O/shard_one_rank.err; echo "one rank rc=$?"
timeout -k 10 300 python bench.py --mode shard --dtype mixed --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/shard_one_rank_mixed.json 2> $O/shard_one_rank_mixed.err; echo "one rank mixed rc=$?"
timeout -k 10 400 python bench.py --gpus 4 --mode group --device 0 --steps 2 --warmup 1 --no-cpu-baseline --no-microbench --stall-timeout 200 > $O/group_four_ranks_one_card.json 2> $O/group_four_ranks_one_card.err; echo "group 4 rc=$?"
timeout -k 10 400 python bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --device 0 --ntrain 16384 --stall-timeout 120 > $O/bare_two_ranks.json 2> $O/bare_two_ranks.err; echo "bare 2 ranks rc=$?"
timeout -k 10 400 python bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --device 0 --ntrain 16384 --dtype float32 --workload C5 --stall-timeout 120 > $O/bare_two_ranks_f32.json 2> $O/bare_two_ranks_f32.err; echo "bare 2 ranks f32 rc=$?"
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_shard_lines/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['n_gpus'], round(d['ms_per_step'],1), round(d['value']), d['dtype'], d['config']['parallelism'], (d.get('shard_check') or {}).get('ok'), d['phases_ms']['chol'], d['phases_ms']['solve'], d['phases_ms']['predict_total'])
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-600:])
P
