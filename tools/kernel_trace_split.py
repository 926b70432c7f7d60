"""Per-(kernel, grid) averages out of a rocprofv3 `*kernel_trace.csv` -- the stats file lumps the image-tower and text-tower launches of a kernel together, this keeps them apart
(the grid of a row kernel is its row count / 4), so that in-step durations can be set against the isolated ones of tools/ln_bench.py and tools/attn_bench.py.

    python tools/kernel_trace_split.py <kernel_trace.csv> [name-filter ...]      (default filter: ln_ attn_ colsum)
"""
import csv
import sys


def sequence(path, needle, before=2, after=1, limit=40):
    """`--seq <needle>`: the launches around every launch whose name contains `needle`, in stream order -- what ran in front of a kernel that is slower in the step than in a loop"""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (r.get("Kernel_Name") or "").split("(")[0].replace("void ", "")[:60], r.get("Grid_Size_X") or r.get("Grid_Size") or "?"))
    rows.sort()
    hits = [i for i, r in enumerate(rows) if needle in r[2]]
    hits = hits[len(hits) // 3:]                            # steady state only (behind the warm-up steps)
    by_dur = sorted(hits, key=lambda i: rows[i][1] - rows[i][0])
    hits = sorted(by_dur[-limit * 3 // 4:] + by_dur[:limit // 4])   # mostly the LONGEST instances (the image tower's), a few of the shortest
    for i in hits:
        line = []
        for j in range(max(0, i - before), min(len(rows), i + after + 1)):
            s0, e0, nm, grid = rows[j]
            gap = (s0 - rows[j - 1][1]) * 1e-3 if j else 0.0
            line.append(f"{'>>' if j == i else '  '}{nm}[{grid}] {(e0 - s0) * 1e-6:.3f} ms (gap {gap:.1f} us)")
        print(" | ".join(line))


def main():
    path = sys.argv[1]
    if len(sys.argv) > 3 and sys.argv[2] == "--seq":
        return sequence(path, sys.argv[3])
    keys = sys.argv[2:] or ["ln_", "attn_", "colsum"]
    acc = {}
    with open(path) as f:
        rd = csv.DictReader(f)
        for r in rd:
            name = r.get("Kernel_Name") or r.get("Name") or ""
            if not any(k in name for k in keys):
                continue
            grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
            wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?"
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            short = name.split("(")[0].replace("void ", "").replace("unsigned short", "bf16")
            a = acc.setdefault((short, grid, wg), [0, 0.0, 1e9, 0.0])
            a[0] += 1
            a[1] += dur
            a[2] = min(a[2], dur)
            a[3] = max(a[3], dur)
    print(f"{'kernel':72s} {'grid':>10s} {'wg':>5s} {'calls':>6s} {'avg ms':>8s} {'min':>8s} {'max':>8s} {'total ms':>9s}")
    for (short, grid, wg), (n, tot, lo, hi) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        if tot < 0.5:
            continue
        print(f"{short[:72]:72s} {grid:>10s} {wg:>5s} {n:6d} {tot / n:8.4f} {lo:8.4f} {hi:8.4f} {tot:9.2f}")


if __name__ == "__main__":
    main()
