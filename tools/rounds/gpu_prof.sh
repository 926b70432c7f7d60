#!/bin/bash
# rocprofv3 kernel stats of bench.py at batch 256 (2 steps + 1 warm-up).  usage: gpu_prof.sh TAG
TAG=${1:-prof}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o prof -- python $GRAFT_REPO_ROOT/bench.py --batch 256 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -32 "$f" | cut -c1-150
find gpurun_out/${TAG}_prof -type f ! -name "*stats*" -delete 2>/dev/null
tail -1 gpurun_out/${TAG}_prof.log | cut -c1-300
