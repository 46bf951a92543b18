#!/bin/bash
# GPU: 128-wide diagonal step (potf2_128_kernel) — parity tests, then A/B against the 64-wide stepping
set -e -o pipefail
mkdir -p gpurun_out
python tools/potf2_128_check.py
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_gp_parity_gpu.py tests/test_fp32_gpu.py tests/test_mixed_gpu.py tests/test_fuzz_gpu.py tests/test_delay_gpu.py -m gpu -x -q > gpurun_out/r03_t2.log 2>&1 || { tail -40 gpurun_out/r03_t2.log; exit 1; }
tail -3 gpurun_out/r03_t2.log
for s in 64 128 64 128; do
  echo "== C2, GPX_DIAG_STEP=$s"; GPX_DIAG_STEP=$s python tools/c2_bench.py 2>/dev/null | tail -1
done
for s in 64 128; do
  echo "== C3, GPX_DIAG_STEP=$s"; GPX_DIAG_STEP=$s python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'], d['phases_ms'])"
done
