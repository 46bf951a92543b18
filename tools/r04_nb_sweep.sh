# panel width 1024 vs 2048 (predict block following) over N, fp64 and fp32: where does the wider panel start to pay?
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04_nb_sweep; rm -rf $O; mkdir -p $O
for N in 8192 16384 24576 32768 49152; do
  for nb in 1024 2048; do
    GPX_NB_PRED=$nb python tools/c2_bench.py --no-profile --ntrain $N --block $nb --steps 5 --warmup 2 2>> $O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']
print('fp64 N=$N nb=$nb', round(d['ms_per_step'],2), 'chol', p['chol'], 'predict', p['predict_total'])" | tee -a $O/sweep.txt
  done
done
for nb in 1024 2048; do
  GPX_NB_PRED=$nb python bench.py --workload C5 --block $nb --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>> $O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms']
print('fp32 C5 nb=$nb', round(d['ms_per_step'],1), 'SYRK', round(d['roofline']['achieved'],1), 'chol', p['chol'], 'predict', p['predict_total'])" | tee -a $O/sweep.txt
done
