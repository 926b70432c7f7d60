#!/bin/bash
# PMC passes (two SQ groups) over the attention microbench for the 32 x 32 forward (variant 0) and the 16 x 16 forward (variant 1)
TAG=${1:-r4h}
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for v in 0 1; do
P=$ROOT/gpurun_out/${TAG}_pmc_attn_v$v
mkdir -p $P
run() { name=$1; shift; ANTMMF_ATTN_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/$name -o p -- python $ROOT/tools/attn_bench.py pmc 1 > $P.$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES
done
cd $ROOT
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/${TAG}_pmc_attn_v*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-44:]
        if "attn_fwd" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", f)
    for k, d in agg.items():
        print(k, {c: round(v) for c, v in d.items()})
PY
find gpurun_out/${TAG}_pmc_attn_v* -name "*.db" -delete 2>/dev/null
