"""Aggregate a rocprofv3 `*kernel_stats.csv` into the step's kernel families, in ms per step.

    python tools/kernel_families.py profiles/r5_bench_l14_kernel_stats.csv [steps]

`steps` = number of train steps the profiled command ran (warm-up included; default: the call count of the AdamW kernel, which runs once per step).
Prints one line per family and the per-kernel averages of the row / attention kernels (the figures DESIGN.md section 7 quotes).
"""
import csv
import sys

FAMILIES = (
    ("gemm", ("gemm_", "splitk_reduce")),
    ("layernorm", ("ln_fwd", "ln_bwd", "ln_partials")),
    ("attention", ("attn_",)),
    ("colsum", ("colsum",)),
    ("adamw", ("adamw",)),
)


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return "other"


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if not steps:
        steps = next((int(r["Calls"]) for r in rows if "adamw" in r["Name"]), 1)
    tot = {}
    for r in rows:
        f = family(r["Name"])
        tot[f] = tot.get(f, 0.0) + float(r["TotalDurationNs"]) * 1e-6
    all_ms = sum(tot.values())
    print(f"{path}: {steps} steps, {all_ms / steps:.1f} ms of kernels per step")
    for fam in [f for f, _ in FAMILIES] + ["other"]:
        ms = tot.get(fam, 0.0) / steps
        print(f"  {fam:10s} {ms:8.2f} ms/step  {100.0 * ms * steps / all_ms:5.1f} %")
    for r in rows:
        if family(r["Name"]) in ("layernorm", "attention") and float(r["TotalDurationNs"]) * 1e-6 / steps > 0.5:
            short = r["Name"].split("(")[0].replace("void ", "").replace("unsigned short", "bf16")
            print(f"    {short:70s} calls/step {int(r['Calls']) / steps:6.1f}  avg {float(r['AverageNs']) * 1e-6:7.4f} ms  {float(r['TotalDurationNs']) * 1e-6 / steps:7.2f} ms/step")


if __name__ == "__main__":
    main()
