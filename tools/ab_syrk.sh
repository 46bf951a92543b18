cd $GRAFT_REPO_ROOT
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > gpurun_out/r2_ab_$name.json 2> gpurun_out/r2_ab_$name.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/r2_ab_$name.json')); print('$name', round(d['ms_per_step'],1), round(d['roofline']['achieved'],2), round(d['roofline']['avg_launch_ms'],3), d['phases_ms']['chol'], d['phases_ms']['chol_trsm'], d['phases_ms']['trsm'], d['outputs_finite'])"
}
timeout -k 10 700 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
run paired GPX_NONE=0
run paired2 GPX_NONE=0
python tools/c2_bench.py
