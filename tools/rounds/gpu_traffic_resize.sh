#!/bin/bash
# HBM-side traffic of the resize kernels (FETCH_SIZE / WRITE_SIZE in separate PMC passes, kernel-trace only).
TAG=${1:-r1_resize}
mkdir -p gpurun_out; export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_traffic
mkdir -p $P
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $P/$c -o p -- python $GRAFT_REPO_ROOT/tools/resize_bench.py 64 > $P.$c.log 2>&1; echo "$c rc=$?"
done
cd $GRAFT_REPO_ROOT
python - "$TAG" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/{tag}_traffic/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "resize" not in k or r["Counter_Name"] != c:
                continue
            d = per.setdefault(k.split("(")[0].replace("void ", ""), {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n_FETCH_SIZE": 0, "n_WRITE_SIZE": 0})
            d[c] += float(r["Counter_Value"]); d["n_" + c] += 1
out = {}
n, h, w, s = 64, 1080, 1920, 224
alg = {"resize_h": (n * h * w * 3, n * h * s * 3), "resize_v": (n * h * s * 3, n * s * s * 3 * 4)}
for k, d in per.items():
    # units: KB; FETCH_SIZE tallies 128-B requests at 64 B on gfx950 -> x2 (MI355X_MICROARCH.md); WRITE_SIZE as reported
    rd = d["FETCH_SIZE"] / max(1, d["n_FETCH_SIZE"]) * 1024 * 2
    wr = d["WRITE_SIZE"] / max(1, d["n_WRITE_SIZE"]) * 1024
    a = alg["resize_h" if "resize_h" in k else "resize_v"]
    out[k] = {"read_bytes_per_launch": int(rd), "write_bytes_per_launch": int(wr), "algorithmic_read": a[0], "algorithmic_write": a[1]}
json.dump({"workload": "64 x 1080x1920x3 -> 224x224 (tools/resize_bench.py 64)", "kernels": out}, open(f"gpurun_out/{tag}_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find gpurun_out/${TAG}_traffic -name "*.db" -delete 2>/dev/null
