"""Per-(kernel, grid) HBM-side bytes of the row / attention kernels out of rocprofv3 `--pmc` passes (counter_collection.csv files):

    python tools/row_pmc.py <dir with FETCH_SIZE/ and WRITE_SIZE/ sub-directories> [name-filter ...]

FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md) and WRITE_SIZE, KB -> MB per launch.  The persistent backward kernels launch the same grid for both towers: their
launches are split at the midpoint of the per-launch byte counts (image-tower tensors are 3.3 x the text tower's)."""
import csv
import glob
import sys


def main():
    base = sys.argv[1]
    keys = sys.argv[2:] or ["ln_", "attn_"]
    data = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{base}/{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if r["Counter_Name"] != c or not any(x in k for x in keys):
                    continue
                short = k.split("(")[0].replace("void ", "").replace("unsigned short", "bf16")[:66]
                grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
                data.setdefault((short, grid), {}).setdefault(c, []).append(float(r["Counter_Value"]) * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6)
    print(f"{'kernel':66s} {'grid':>9s} {'part':>6s} {'calls':>6s} {'read MB':>9s} {'write MB':>9s}")
    for (short, grid), d in sorted(data.items()):
        fe, wr = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
        if not fe:
            continue
        mid = (min(fe) + max(fe)) / 2 if max(fe) > 1.8 * min(fe) else None
        parts = [("all", fe, wr)] if mid is None else [("small", [x for x in fe if x < mid], None), ("large", [x for x in fe if x >= mid], None)]
        if mid is not None and wr:
            wmid = (min(wr) + max(wr)) / 2
            parts = [("small", parts[0][1], [x for x in wr if x < wmid]), ("large", parts[1][1], [x for x in wr if x >= wmid])]
        for name, f_, w_ in parts:
            if f_:
                print(f"{short:66s} {grid:>9s} {name:>6s} {len(f_):6d} {sum(f_) / len(f_):9.1f} {(sum(w_) / len(w_)) if w_ else float('nan'):9.1f}")


if __name__ == "__main__":
    main()
