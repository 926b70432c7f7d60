"""Summary of tools/overlap_probe.py runs:  python tools/overlap_summary.py gpurun_out/r6b_overlap_probe_G*.jsonl

For every CU split G / 256 - G and every (GEMM, row / attention kernel) pair: the sum of the two sides' rates while they share the chip, each normalised by its rate
on the whole chip (the G = 256 file's plain-stream times).  1.00 = no better than running them one after the other; the verdict's bar for building the two-stream
step was 1.12.  The last line per G weights the pairs by the B kernels' share of the l14 step (ms per step from profiles/r5_bench_l14_kernel_stats.csv)."""
import json
import sys

STEP_MS = {"ln_fwd_d": 17.5, "ln_fwd_4d_gelu": 29.3, "ln_bwd_4d_gelu": 40.9, "ln_bwd_renorm_d": 46.8, "attn_fwd": 20.0, "attn_bwd": 49.3}


def main():
    runs = {}
    for path in sys.argv[1:]:
        rows = [json.loads(l) for l in open(path) if l.startswith("{")]
        if rows:
            runs[rows[-1]["G"]] = rows
    ref = runs[256]
    a_ref = {r["kernel"]: r["ms_plain_stream"] for r in ref if r["side"] == "A"}
    b_ref = {r["kernel"]: r["ms_plain_stream"] for r in ref if r["side"] == "B"}
    box = next((r for r in ref if r["side"] == "box"), {})
    print(f"whole chip (G = 256 process, plain stream): power cap {box.get('power_cap_w')} W")
    for r in ref:
        if r["side"] == "A":
            print(f"  {r['kernel']:16s} {r['ms_plain_stream']:7.3f} ms  {r['tf_on_share']:7.1f} TF  clock {r['clock_plain']} MHz  {r.get('watts_plain')} W")
        if r["side"] == "B":
            print(f"  {r['kernel']:16s} {r['ms_plain_stream']:7.3f} ms  {r['tbs_plain_stream']:6.2f} TB/s  {r.get('watts_plain')} W")
    for G in sorted(runs, reverse=True):
        rows = runs[G]
        print(f"\nG = {G} CUs for the GEMM, {256 - G if G < 256 else 'the same 256'} for the row / attention kernel")
        for r in rows:
            if r["side"] == "A":
                print(f"  alone on its share: {r['kernel']:12s} {r['ms_on_share']:7.3f} ms ({a_ref[r['kernel']] / r['ms_on_share']:.2f} of its whole-chip rate) clock {r['clock_share']} MHz {r.get('watts_share')} W")
            if r["side"] == "B":
                print(f"  alone on its share: {r['kernel']:16s} {r['ms_on_share']:7.3f} ms ({b_ref[r['kernel']] / r['ms_on_share']:.2f}) {r['tbs_on_share']:.2f} TB/s {r.get('watts_share')} W")
        by_a = {}
        for r in rows:
            if r["side"] != "pair":
                continue
            ra, rb = a_ref[r["A"]] / r["A_ms_beside_B"], b_ref[r["B"]] / r["B_ms_beside_A"]
            print(f"  {r['A']:10s} + {r['B']:16s}: A {ra:.3f} + B {rb:.3f} = {ra + rb:.3f}   (A {r['A_ms_beside_B']:.3f} ms at {r['clock_beside']} MHz, B {r['B_ms_beside_A']:.3f} ms, {r.get('watts')} W"
                  f"{'' if r.get('covered', True) else ', NOT covered'})")
            by_a.setdefault(r["A"], []).append((STEP_MS.get(r["B"], 0.0), ra + rb))
        for a, lst in by_a.items():
            w = sum(x for x, _ in lst)
            if w:
                print(f"  {a}: step-weighted sum of rates {sum(x * s for x, s in lst) / w:.3f}")


if __name__ == "__main__":
    main()
