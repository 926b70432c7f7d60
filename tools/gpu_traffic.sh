#!/bin/bash
# HBM-side traffic of the GEMM kernels over ONE default bench step, from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
# separate passes, kernel-trace only -- MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots").  Writes gpurun_out/<tag>_gemm_traffic.json.
TAG=${1:-r1}
BATCH=${2:-1024}
mkdir -p gpurun_out; export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_traffic
mkdir -p $P
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/$c -o p -- python $GRAFT_REPO_ROOT/bench.py --batch $BATCH --steps 1 --warmup 1 --no-cpu-baseline > $P.$c.log 2>&1; echo "$c rc=$?"
done
cd $GRAFT_REPO_ROOT
python - "$TAG" "$BATCH" <<'PY'
import csv, glob, json, sys
tag, batch = sys.argv[1], int(sys.argv[2])
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}; n = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}; per = {}
for c in tot:
    for f in glob.glob(f"gpurun_out/{tag}_traffic/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_" not in k or r["Counter_Name"] != c:
                continue
            tot[c] += float(r["Counter_Value"]); n[c] += "tail_reduce" not in k   # (the reduce launch of a split tail round belongs to its GEMM call: bytes yes, launch no)
            d = per.setdefault(k.split("(")[0][-44:], {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
            d[c] += float(r["Counter_Value"]); d["launches"] += c == "FETCH_SIZE"
# units: KB; gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> x2 (guide); WRITE_SIZE matched the algorithmic bytes of the
# fc1 forward (526 vs 539 MB) in the calibration run, so it is used as reported
launches = max(1, n["FETCH_SIZE"])
read_b, write_b = tot["FETCH_SIZE"] * 1024 * 2, tot["WRITE_SIZE"] * 1024
out = {"workload": "l14", "per_gpu_batch": batch, "gemm_launches": launches, "steps_profiled": 2,
       "read_bytes_per_launch": read_b / launches, "write_bytes_per_launch": write_b / max(1, n["WRITE_SIZE"]),
       "traffic_bytes_per_launch": read_b / launches + write_b / max(1, n["WRITE_SIZE"]),
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), all gemm_* kernels of bench.py --steps 1 --warmup 1; FETCH_SIZE x2 (gfx950 correction); L2-miss traffic incl. Infinity-Cache hits",
       "per_kernel": {k: {"launches": v["launches"], "read_mb_per_launch": round(v["FETCH_SIZE"] * 2048 / max(1, v["launches"]) / 1e6, 1),
                          "write_mb_per_launch": round(v["WRITE_SIZE"] * 1024 / max(1, v["launches"]) / 1e6, 1)} for k, v in per.items()}}
json.dump(out, open(f"gpurun_out/{tag}_gemm_traffic.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
find gpurun_out/${TAG}_traffic -type f ! -name "*counter_collection.csv" -delete 2>/dev/null
find gpurun_out/${TAG}_traffic -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
