# Round 4: the whole -m gpu suite (what the driver runs at round end), smoke(), then a 120-case fuzz sweep over two new seeds.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04_suite; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=12 > $O/gpu_suite.log 2>&1
rc=$?; echo "suite rc=$rc"; tail -20 $O/gpu_suite.log; [ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
for seed in 49; do timeout -k 10 400 python tools/fuzz_parity.py --cases 60 --seed $seed > $O/fuzz_$seed.log 2>&1; echo "fuzz $seed rc=$?"; tail -1 $O/fuzz_$seed.log; done
