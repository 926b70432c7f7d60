#!/bin/bash
TAG=${1:-r3o}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
for lib in "" attn640; do
  if [ -n "$lib" ]; then export ANTMMF_HIP_LIB=$GRAFT_REPO_ROOT/ant-multi-modal-framework_amd/lib/libantmmf_hip_$lib.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_$lib -o p -- python $GRAFT_REPO_ROOT/tools/attn_bench.py x$lib 10 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_$lib -name "*kernel_stats.csv" | head -1); echo "--- lib=$lib"; grep "attn_" $f | cut -d, -f1-4 | cut -c1-120
done | tee $GRAFT_REPO_ROOT/gpurun_out/${TAG}_attn_threads_stats.txt
