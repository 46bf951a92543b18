# Round 4, GPU check 5: fp32 / mixed on the shard, the Comm rewrite (byte moves + typed reductions) on every transport, kbuild
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r04_check5; rm -rf $O; mkdir -p $O
set -x
timeout -k 10 900 python -m pytest tests/test_fp32_gpu.py tests/test_mixed_gpu.py tests/test_group_gpu.py tests/test_shard_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -s > $O/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -30 $O/pytest.log
