#!/bin/bash
# does a --pmc pass work WITHOUT any GPX_* switch (the library detects counter collection)?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_auto; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O -- python3 $R/tools/c2_bench.py --steps 2 > $O/run.json 2> $O/run.err; echo "rc=$?"
tail -c 200 $O/run.json
find $O -name "*counter_collection.csv" | head -2; find $O -name "*counter_collection.csv" -delete
