cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04_shard_nb; rm -rf $O; mkdir -p $O
line() { python -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); p=d['phases_ms']
print('$2', round(d['ms_per_step'],1), 'chol', p['chol'], 'solve', p['solve'], 'predict', p['predict_total'], 'check', (d.get('shard_check') or {}).get('ok'))"; }
for rep in 1 2; do
for nb in 1024 2048; do
GPX_NB_SHARD=$nb timeout -k 10 300 python bench.py --mode shard --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $O/one_rank_$nb.json 2>> $O/err.log; line $O/one_rank_$nb.json "one rank nb_shard=$nb"
done; done
timeout -k 10 300 python -m pytest tests/test_group_gpu.py tests/test_shard_gpu.py -m gpu -x -q -k "oracle or c4_shape_on_one_gpu_world8" > $O/pytest_2048.log 2>&1 <<< "" ; true
GPX_NB_SHARD=2048 timeout -k 10 300 python - > $O/group4_2048.txt 2>&1 <<'P'
import os, numpy as np
from gaussianprocesspathmodelling_amd import GP
from oracle.gp_oracle import OracleGP, synthetic_problem
for repl in ("0", "1"):
    os.environ["GPX_SHARD_REPLICATE"] = repl
    X, y, Xs = synthetic_problem(20000, 3, 300, seed=5)
    ref = OracleGP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0).fit(X, y)
    mr, vr = ref.predict(Xs)
    with GP("matern52", (0.3, 0.2, 0.25), 1.5, 1e-2, jitter=0.0, devices=3, oversubscribe=True) as gp:
        m, v = gp.fit(X, y).predict(Xs)
        print("repl", repl, "nb_shard 2048, 3 ranks, N=20000: mean rel", float(np.max(np.abs(m - mr) / np.maximum(np.abs(mr), 1e-6))), "var rel", float(np.max(np.abs(v - vr) / np.maximum(vr, 1.5e-6))), "alpha", float(np.max(np.abs(gp.alpha_ - ref.alpha_)) / np.max(np.abs(ref.alpha_))))
P
cat $O/group4_2048.txt | tail -3
