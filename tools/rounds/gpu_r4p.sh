#!/bin/bash
# round 4: what the stores of the NT GEMM cost, in counters: rolling epilogue (8196) vs the same kernel without its stores (40964), dgrad_fc2 shape
TAG=${1:-r4p}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
P=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc
mkdir -p $P
rocprofv3 --list-avail > $P.avail.txt 2>&1
grep -o "TA_[A-Z0-9_a-z]*\|TCP_[A-Z0-9_a-z]*\|TD_[A-Z0-9_a-z]*\|SQ_INST_CYCLES[A-Z0-9_a-z]*\|SQ_INSTS_VMEM[A-Z_a-z]*\|SQ_WAIT[A-Z_a-z]*\|SQ_ACTIVE_INST[A-Z_a-z]*" $P.avail.txt | sort -u | tr '\n' ' ' | cut -c1-6000
echo
run() { v=$1; name=$2; shift; shift; ANTMMF_GEMM_VARIANT=$v timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/v${v}_$name -o p -- python $GRAFT_REPO_ROOT/tools/gemm_pmc2.py 4096 1024 6 > $P.v${v}_$name.log 2>&1; echo "v$v $name rc=$?"; }
for v in 8196 40964; do
run $v sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run $v sq2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE
# (TA_* / TCP_* / TD_* passes HANG on this pool until the timeout -- 3 x 300 s lost in round 4: not scheduled)
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/*_pmc/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-30:]
        if "gemm" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    print("==", f.split("/")[-2])
    for k, d in agg.items():
        print("  ", k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
find gpurun_out/${TAG}_pmc -name "*.db" -delete 2>/dev/null
grep -il "error\|invalid\|not found" gpurun_out/${TAG}_pmc.v*.log | head
