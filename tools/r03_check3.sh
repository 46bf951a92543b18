#!/bin/bash
# GPU: fused trailing update (strip + rest in one launch, device-counter hand-over) — parity, then A/B
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 120 python tools/potf2_128_check.py
timeout -k 10 700 python -m pytest tests/test_kernels_gpu.py tests/test_gp_parity_gpu.py tests/test_fp32_gpu.py tests/test_mixed_gpu.py tests/test_fuzz_gpu.py tests/test_delay_gpu.py tests/test_full_size_gpu.py -m gpu -x -q > gpurun_out/r03_t3.log 2>&1 || { tail -40 gpurun_out/r03_t3.log; exit 1; }
tail -3 gpurun_out/r03_t3.log
for f in 0 1 0 1; do
  echo "== C3, GPX_FUSED_STRIP=$f"; GPX_FUSED_STRIP=$f timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'], d['roofline']['launches'], d['phases_ms'])"
done
for f in 0 1; do
  echo "== C2, GPX_FUSED_STRIP=$f"; GPX_FUSED_STRIP=$f python tools/c2_bench.py 2>/dev/null | tail -1
done
