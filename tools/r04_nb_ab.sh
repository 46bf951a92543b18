# panel width A/B at C3 on the round-4 schedule: 1024 (default) vs 2048 (with the predict block width following)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04_nb_ab; rm -rf $O; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); p=d['phases_ms']
print('$2', round(d['ms_per_step'],1), 'SYRK', round(d['roofline']['achieved'],2), 'launches', d['roofline']['launches'], 'chol', p['chol'], 'diag', p['chol_diag'], 'strip', p['chol_strip'], 'trsm', p['chol_trsm'], 'predict', p['predict_total'])"; }
for rep in 1 2; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/nb1024_$rep.json 2> $O/err.log; line $O/nb1024_$rep.json "nb=1024"
GPX_NB_PRED=2048 python bench.py --block 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/nb2048_$rep.json 2>> $O/err.log; line $O/nb2048_$rep.json "nb=2048 pred=2048"
python bench.py --block 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/nb2048p1024_$rep.json 2>> $O/err.log; line $O/nb2048p1024_$rep.json "nb=2048 pred=1024(slab)"
python bench.py --block 1536 --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/nb1536_$rep.json 2>> $O/err.log; line $O/nb1536_$rep.json "nb=1536"
done
