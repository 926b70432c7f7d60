#!/bin/bash
# round 5: MFMA utilisation of the product GEMM kernels from SQ counters (north_star: "evidenced by rocprof ... MFMA utilisation against the chip's peak"); counters in their own
# pass, kernel-trace only
TAG=${1:-r5i}
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
P=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc
mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $P/sq -o p -- python $GRAFT_REPO_ROOT/tools/gemm_pmc3.py 4 > $P.sq.log 2>&1; echo "sq rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r5i_pmc_gemm_mfma_util.txt
import csv, glob, collections
print("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python tools/gemm_pmc3.py 4  (product library; per-launch averages;")
print("# 263168-token shapes of the l14 step; MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE), the normalisation of profiles/r4_pmc_gemm_store_sq.txt)")
rows = collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/r5i_pmc/sq/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-34:]
        if "gemm" not in k: continue
        key = (k, r.get("Dispatch_Id") or r.get("Correlation_Id"))
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
# group consecutive dispatches of the same kernel into runs of 4 launches (one shape each)
runs, last = [], None
for (k, d), c in rows.items():
    if last is None or last[0] != k or len(last[1]) >= 4:
        last = [k, []]; runs.append(last)
    last[1].append(c)
for k, cs in runs:
    n = len(cs); avg = {c: sum(x.get(c, 0) for x in cs) / n for c in cs[0]}
    util = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, 128.0 * avg.get("GRBM_GUI_ACTIVE", 1))
    print(f"{k:36s} launches {n}  GUI_ACTIVE {avg.get('GRBM_GUI_ACTIVE', 0):12.0f}  MFMA_BUSY {avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):14.0f}  MFMA pipe busy {util:.3f}  wait share of wave cycles {avg.get('SQ_WAIT_INST_ANY', 0) / max(1.0, avg.get('SQ_WAVE_CYCLES', 1)):.3f}")
PY
find gpurun_out/${TAG}_pmc -name "*.db" -delete 2>/dev/null; find gpurun_out/${TAG}_pmc -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
