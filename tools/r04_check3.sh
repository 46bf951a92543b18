# Round 4, third GPU check: the self-reserving trailing update (GPX_CU_SELF_RESERVE=k) — correctness (schedule variants
# bit-identical), then A/B at C2 (two calls, one pass) and C3, then a kernel trace of C2 with and without it.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04_check3; rm -rf $O; mkdir -p $O
set -x
timeout -k 10 600 python -m pytest tests/test_gp_parity_gpu.py tests/test_fit_predict_gpu.py tests/test_fp32_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "schedule or fit_predict or fp32 or potrf" > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.log
[ $rc -eq 0 ] || exit 1
{ bash tools/ab_sweep.sh GPX_CU_SELF_RESERVE "0 1 2 4 0 1 2 4" c2; bash tools/ab_sweep.sh GPX_CU_SELF_RESERVE "0 1 2 4 0 1 2 4" c2onepass; bash tools/ab_sweep.sh GPX_CU_SELF_RESERVE "0 1 2 0 1 2" c3; } > $O/ab.txt 2> $O/ab.err
cat $O/ab.txt
