#!/bin/bash
# PMC passes over the attention microbench (counters in their own runs, kernel-trace only).
TAG=${1:-r2}
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
P=$ROOT/gpurun_out/${TAG}_pmc_attn
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/$name -o p -- python $ROOT/tools/attn_bench.py pmc 1 > $P.$name.log 2>&1; echo "$name rc=$?"; }
mkdir -p $P
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES
run sq3 SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VALU_TRANS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_EXP_GDS
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $ROOT
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/${TAG}_pmc_attn/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "attn" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", f)
    for k, d in agg.items():
        print(k, {c: round(v) for c, v in d.items()})
PY
find gpurun_out/${TAG}_pmc_attn -name "*.db" -delete 2>/dev/null
