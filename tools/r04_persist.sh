# the persistent (lock-step) form of the trailing update for EVERY panel at C3, leaving k CUs per XCD to the chain
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04_persist; rm -rf $O; mkdir -p $O
line() { python -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); p=d['phases_ms']
print('$2', round(d['ms_per_step'],1), 'SYRK', round(d['roofline']['achieved'],2), 'chol', p['chol'], 'diag', p['chol_diag'], 'strip', p['chol_strip'], 'predict', p['predict_total'])"; }
run() { env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/x.json 2>> $O/err.log; }
for rep in 1 2; do
run A=1; line $O/x.json "default"
run GPX_CU_SELF_RESERVE=1 GPX_RESV_ALL=1 GPX_RESV_FORM=0; line $O/x.json "persistent all panels k=1"
run GPX_CU_SELF_RESERVE=2 GPX_RESV_ALL=1 GPX_RESV_FORM=0; line $O/x.json "persistent all panels k=2"
run GPX_CU_SELF_RESERVE=1 GPX_RESV_ALL=1 GPX_RESV_FORM=1; line $O/x.json "turnover all panels k=1"
done
