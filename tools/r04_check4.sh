# C2 A/B of the self-reserving update's forms, then traces
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04_check4; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gp_parity_gpu.py -m gpu -x -q -k "schedule" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
{
for f in 1 0; do echo "== GPX_RESV_FORM=$f"; GPX_RESV_FORM=$f bash tools/ab_sweep.sh GPX_CU_SELF_RESERVE "0 1 2 4 0 4" c2; done
echo "== one pass, form 1"; bash tools/ab_sweep.sh GPX_CU_SELF_RESERVE "0 1 2 4 0 4" c2onepass
echo "== form 1, chain untouched"; GPX_RESV_CHAIN=0 bash tools/ab_sweep.sh GPX_CU_SELF_RESERVE "0 2 4" c2
} > $O/ab.txt 2>&1
cat $O/ab.txt
export C2_ARGS=--no-profile GAPS_ARGS=--main
for v in 0 4; do
  GPX_CU_SELF_RESERVE=$v bash tools/r03_c2trace.sh > $O/gaps_resv$v.txt 2> $O/gaps_resv$v.err
done
grep -h "mean step\|block boundaries" $O/gaps_*.txt
