#!/bin/bash
# round 5: where the one-kernel attention backward spends its time -- SQ counter passes + timing-only ablations (lab library) on the 1024 x 16 x 257 shape
TAG=${1:-r5l}
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=$(pwd)
export ANTMMF_HIP_LIB=$ROOT/ant-multi-modal-framework_amd/lib/libantmmf_hip_lab.so
for v in 0 1 2 4 8 16 32 15 0; do echo "--- ANTMMF_ATTN_FUSED_ABL=$v"; ANTMMF_ATTN_FUSED_ABL=$v timeout 300 python tools/attn_bench.py abl$v 10 2>&1 | grep "bwd.N257" | cut -c1-200; done | tee gpurun_out/${TAG}_attn_bwd_one_kernel_ablations.txt
cd /tmp
P=$ROOT/gpurun_out/${TAG}_pmc_attn
mkdir -p $P
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/$name -o p -- python $ROOT/tools/attn_bench.py pmc 1 > $P.$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
cd $ROOT
python - <<PY | tee gpurun_out/${TAG}_pmc_attn_summary.txt
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/${TAG}_pmc_attn/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-48:]
        if "attn" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print("==", f)
    for k, d in agg.items():
        print(k, {c: round(v) for c, v in d.items()})
PY
find gpurun_out/${TAG}_pmc_attn -name "*.db" -delete 2>/dev/null
