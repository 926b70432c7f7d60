#!/bin/bash
# PMC passes over the GEMM microbench (counters in their own runs, kernel-trace only).
TAG=${1:-r1e}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
P=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/$name -o p -- python $GRAFT_REPO_ROOT/tools/gemm_pmc.py 2 > $P.$name.log 2>&1; echo "$name rc=$?"; }
mkdir -p $P
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_WAVES
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
tag = os.environ.get("TAG_", "")
for f in sorted(glob.glob("gpurun_out/*_pmc/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    print("==", f)
    for k, d in agg.items():
        print(k, {c: round(v) for c, v in d.items()})
PY
find gpurun_out/${TAG}_pmc -name "*.db" -delete 2>/dev/null
