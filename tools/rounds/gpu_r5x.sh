#!/bin/bash
# round 5: counter passes over the final attention kernels at the ViT-L/14 step's shapes (product library): SQ busy / wait / LDS / MFMA, FETCH_SIZE, WRITE_SIZE -- separate passes
TAG=${1:-r5x}
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
P=$ROOT/gpurun_out/${TAG}_pmc_attn
mkdir -p $P
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/$name -o p -- python $ROOT/tools/attn_bench.py pmc 1 > $P.$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE
cd $ROOT
python - <<PY | tee gpurun_out/${TAG}_pmc_attn_summary.txt
import csv, glob, collections
print("# rocprofv3 --pmc passes over tools/attn_bench.py (1024 x 16 heads x 64; 257 and 77 tokens; 3 launches of each kernel per pass: sums over the launches)")
for f in sorted(glob.glob("gpurun_out/${TAG}_pmc_attn/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "attn" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", f)
    for k, d in agg.items():
        print(k, {c: round(v) for c, v in d.items()})
PY
find gpurun_out/${TAG}_pmc_attn -name "*.db" -delete 2>/dev/null
