#!/bin/bash
# PMC passes over the attention kernels (forward + backward at 256 pairs, packed QKV).
TAG=${1:-attn}
mkdir -p gpurun_out; export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc
mkdir -p $P
cd /tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/$name -o p -- python $GRAFT_REPO_ROOT/tools/attn_layout_exp.py 256 > $P.$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/*_pmc/sq*/*counter_collection.csv")):
    if "attn" not in f: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-28:]
        if "attn" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", f)
    for k, d in agg.items():
        print(k, {c: round(v / 1e6, 1) for c, v in d.items()})
PY
find gpurun_out/${TAG}_pmc -name "*.db" -delete 2>/dev/null
