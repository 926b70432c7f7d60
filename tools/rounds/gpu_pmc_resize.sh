#!/bin/bash
# PMC passes over the resize kernels (64 full-HD frames -> 224).
TAG=${1:-resize}
mkdir -p gpurun_out; export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc
mkdir -p $P
cd /tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/$name -o p -- python $GRAFT_REPO_ROOT/tools/resize_bench.py 64 > $P.$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/*_pmc/sq*/*counter_collection.csv")):
    if "resize" not in f: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-28:]
        if "resize" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    print("==", f)
    for k, d in agg.items():
        print(k, {c: round(v / cnt[(k, c)] / 1e6, 2) for c, v in d.items()}, "(millions per launch)")
PY
find gpurun_out/${TAG}_pmc -name "*.db" -delete 2>/dev/null
