cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
bash tools/c3_trace.sh > /dev/null 2>&1
mkdir -p gpurun_out/r04_c3trace
python tools/c3_gaps.py gpurun_out/c3trace/trace.csv > gpurun_out/r04_c3trace/gaps.txt 2>&1
python tools/c3_main_stream.py gpurun_out/c3trace/trace.csv 65536 2048 > gpurun_out/r04_c3trace/main_stream.txt 2>&1
rm -rf gpurun_out/c3trace/*/ gpurun_out/c3trace/trace.csv
head -30 gpurun_out/r04_c3trace/main_stream.txt
