"""Can the MFMA-bound and the HBM-bound halves of the step share the chip?  (VERDICT r5, next #1.)

Two HIP streams created with hipExtStreamCreateWithCUMask: stream A owns G CUs (G / 8 per XCD: the mask bits interleave over the XCDs first) and loops a GEMM of the
step, stream B owns the other 256 - G CUs and loops one of the step's row / attention kernels.  Each side is timed (i) on the whole chip, (ii) alone on its CU share,
(iii) with the other side running beside it for the WHOLE of its loop (the partner's loop is made 1.6 x as long).  The persistent kernels are sized for their share
through the lab library's grid knobs, which are read once per process -- so one process per G:

    ANTMMF_HIP_LIB=.../libantmmf_hip_lab.so ANTMMF_GEMM_PERSIST_WGS=$G ANTMMF_WGRAD_WGS=$G ANTMMF_ATTN_PERSIST_WGS=$((256-G)) ANTMMF_ROW_CUS=$((256-G)) \
        python tools/overlap_probe.py $G [tag]

G = 256: both streams unmasked, full grids (what two plain streams give today).  One JSON line per kernel alone and per (A kernel, B kernel) pair, with the GEMM's
in-kernel shader clock and the board power (hwmon) over each loop.  tools/overlap_summary.py turns the files into the number that would have to reach 1.12:
rate_A(beside B) / rate_A(whole chip) + rate_B(beside A) / rate_B(whole chip).
"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ant-multi-modal-framework_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from antmmf.hip import _lib, ops  # noqa: E402
from _power import PowerSampler, power_cap_w  # noqa: E402


def hip_runtime():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    raise RuntimeError("libamdhip64 is not mapped")


def masked_stream(hip, lo, hi):
    """A stream whose kernels may only run on the CUs whose mask bits lie in [lo, hi) (bit i = XCD i % 8, CU i / 8 of it)."""
    words = (ctypes.c_uint32 * 8)()
    for i in range(lo, hi):
        words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value)


def loop_ms(stream, fn, n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with PowerSampler() as ps:
        with torch.cuda.stream(stream):
            a.record()
            for _ in range(n):
                fn()
            b.record()
        torch.cuda.synchronize()
    loop_ms.watts = ps.mean_w
    return a.elapsed_time(b) / n


def pair_ms(sa, fa, na, sb, fb, nb):
    """Both loops at once (host enqueues them interleaved in proportion); per-launch ms of each side from its own stream's events, the ms by which B's loop outlasted
    A's (negative: A ran alone at the end) and the board power over the run."""
    ea0, ea1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    with PowerSampler() as ps:
        with torch.cuda.stream(sa):
            ea0.record()
        with torch.cuda.stream(sb):
            eb0.record()
        ia = ib = 0
        while ia < na or ib < nb:
            if ib >= nb or (ia < na and ia * nb <= ib * na):
                with torch.cuda.stream(sa):
                    fa()
                ia += 1
            else:
                with torch.cuda.stream(sb):
                    fb()
                ib += 1
        with torch.cuda.stream(sa):
            ea1.record()
        with torch.cuda.stream(sb):
            eb1.record()
        torch.cuda.synchronize()
    return ea0.elapsed_time(ea1) / na, eb0.elapsed_time(eb1) / nb, ea1.elapsed_time(eb1), ps.mean_w


def covered_pair(sa, fa, ta, sb, fb, tb, measure, span=150.0):
    """Per-launch ms of side `measure` ("A" / "B") while the OTHER side runs for the whole of its loop: the partner's launch count is doubled until its loop ends after
    the measured one (a side that slows down 3 x beside its partner would otherwise finish its loop alone)."""
    n_meas = max(4, int(span / (ta if measure == "A" else tb)))
    n_part = max(4, int(1.6 * span / (tb if measure == "A" else ta)))
    for _ in range(5):
        if measure == "A":
            t_a, t_b, b_after_a, watts = pair_ms(sa, fa, n_meas, sb, fb, n_part)
            if b_after_a >= 0:
                return t_a, watts, n_part
        else:
            t_a, t_b, b_after_a, watts = pair_ms(sa, fa, n_part, sb, fb, n_meas)
            if b_after_a <= 0:
                return t_b, watts, n_part
        n_part *= 2
    return (t_a if measure == "A" else t_b), watts, -n_part   # (negative count: coverage not reached)


def gemm_clock_mhz():
    lib = _lib.load()
    fn = getattr(lib, "antmmf_debug_gemm_clock", None)
    if fn is None:
        return None
    buf = (ctypes.c_ulonglong * 2)()
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = ctypes.c_int
    torch.cuda.synchronize()
    if fn(buf) != 0 or buf[1] == 0:
        return None
    return round(buf[0] / buf[1] * 100.0, 1)   # shader ticks per 100-MHz tick


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    tag = sys.argv[2] if len(sys.argv) > 2 else "r6"
    only = os.environ.get("OVERLAP_ONLY", "")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    hip = hip_runtime()
    full = torch.cuda.Stream()
    if G < 256:
        sa, sb = masked_stream(hip, 0, G), masked_stream(hip, G, 256)
    else:
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    BF = torch.bfloat16
    T, d, F = 1024 * 257, 1024, 4096

    def rnd(*shape, dtype=BF, scale=1.0):
        return (torch.randn(*shape, device=dev) * scale).to(dtype)

    x_d, x_f = rnd(T, d), rnd(T, F)
    y_d, y_f = torch.empty_like(x_d), torch.empty_like(x_f)
    w1, w1t = rnd(F, d, scale=0.03), rnd(d, F, scale=0.03)
    b1 = rnd(F, dtype=torch.float32)
    gw = torch.zeros(F, d, device=dev)
    ws = {}

    def a_fc1():
        ops.gemm(x_d, w1, bias=b1, out=y_f)                      # 263168 x 4096 x 1024 + bias (NT rolling kernel)

    def a_dgrad():
        ops.gemm(x_f, w1t, out=y_d)                              # 263168 x 1024 x 4096 plain (dgrad_fc1)

    def a_wgrad():
        ops.gemm_wgrad_(gw, x_f, x_d, 1)                         # 4096 x 1024 over 263168 tokens (TN)

    # B side: the row kernels and the attention kernels of the image tower
    g_d, be_d = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev) * 0.1
    g_f, be_f = torch.rand(F, device=dev) + 0.5, torch.randn(F, device=dev) * 0.1
    dy_d, dy_f = rnd(T, d), rnd(T, F)
    dr_d = rnd(T, d)     # the residual-branch gradient: a tensor of its own (the runs of profiles/r6_two_stream_overlap_* passed dres = dy: four distinct streams, not five)
    _, m_d, r_d = ops.layernorm_fwd(x_d, g_d, be_d, 1e-5)
    _, m_f, r_f = ops.layernorm_fwd(x_f, g_f, be_f, 1e-5, act="gelu")
    dg_d, db_d, dxs_d = torch.zeros(d, device=dev), torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    dg_f, db_f, dxs_f = torch.zeros(F, device=dev), torch.zeros(F, device=dev), torch.zeros(F, device=dev)
    qkv = rnd(1024, 257, 3 * d)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    o, lse = ops.attention_fwd(q, k, v, 16, 0.125, None)
    do = rnd(1024, 257, d)
    dqkv = torch.empty_like(qkv)
    e_d, e_f = T * d * 2, T * F * 2
    B = {
        "ln_fwd_d": (lambda: ops.layernorm_fwd(x_d, g_d, be_d, 1e-5), 2 * e_d, 4),
        "ln_fwd_4d_gelu": (lambda: ops.layernorm_fwd(x_f, g_f, be_f, 1e-5, act="gelu"), 2 * e_f, 1),
        "ln_bwd_4d_gelu": (lambda: ops.layernorm_bwd(dy_f, x_f, m_f, r_f, g_f, dg_f, db_f, act="gelu", dxsum=dxs_f), 3 * e_f, 1),
        "ln_bwd_renorm_d": (lambda: ops.layernorm_bwd_renorm(dy_d, x_d, m_d, r_d, g_d, be_d, dg_d, db_d, dres=dr_d, dxsum=dxs_d), 5 * e_d, 3),
        "attn_fwd": (lambda: ops.attention_fwd(q, k, v, 16, 0.125, None), 4 * e_d, 1),
        "attn_bwd": (lambda: ops.attention_bwd(q, k, v, o, lse, do, 16, 0.125, None, dq=dqkv[..., :d], dk=dqkv[..., d:2 * d], dv=dqkv[..., 2 * d:]), 8 * e_d, 1),
    }
    A = {"fc1_bias": (a_fc1, 2.0 * T * F * d), "dgrad_fc1": (a_dgrad, 2.0 * T * F * d), "wgrad_fc1": (a_wgrad, 2.0 * T * F * d)}
    if only:
        keep = set(only.split(","))
        A = {k_: v_ for k_, v_ in A.items() if k_ in keep} or A
        B = {k_: v_ for k_, v_ in B.items() if k_ in keep} or B
    out = []

    def emit(dct):
        dct.update(G=G, tag=tag)
        print(json.dumps(dct), flush=True)
        out.append(dct)

    # (i) / (ii): every kernel alone -- on an ordinary stream (whole chip; the grid knobs of this process still apply) and on its CU share; ~0.4 s loops so that the
    # board-power average settles
    alone = {}
    emit({"side": "box", "power_cap_w": power_cap_w()})
    for name, (fn, flop) in A.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t_full = loop_ms(full, fn, 200)
        w_full, clk_full = loop_ms.watts, gemm_clock_mhz()
        t_share = loop_ms(sa, fn, 200)
        w_share, clk = loop_ms.watts, gemm_clock_mhz()
        alone[name] = t_share
        emit({"side": "A", "kernel": name, "ms_plain_stream": round(t_full, 4), "ms_on_share": round(t_share, 4), "tf_on_share": round(flop / t_share / 1e9, 1),
              "clock_plain": clk_full, "clock_share": clk, "watts_plain": w_full, "watts_share": w_share})
    for name, (fn, nbytes, _) in B.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        n = max(20, int(300.0 / max(loop_ms(full, fn, 10), 1e-3)))
        t_full = loop_ms(full, fn, n)
        w_full = loop_ms.watts
        n = max(20, min(n, int(300.0 / max(loop_ms(sb, fn, 5), 1e-3))))
        t_share = loop_ms(sb, fn, n)
        alone[name] = t_share
        emit({"side": "B", "kernel": name, "ms_plain_stream": round(t_full, 4), "ms_on_share": round(t_share, 4), "tbs_plain_stream": round(nbytes / t_full / 1e9, 3),
              "tbs_on_share": round(nbytes / t_share / 1e9, 3), "watts_plain": w_full, "watts_share": loop_ms.watts})
    # (iii) pairs: A measured while B runs for all of A's loop, then B measured while A runs for all of B's loop
    for an, (fa, flop) in A.items():
        for bn, (fb, nbytes, _) in B.items():
            ta, tb = alone[an], alone[bn]
            ta_pair, w_a, cov_a = covered_pair(sa, fa, ta, sb, fb, tb, "A")
            clk = gemm_clock_mhz()
            tb_pair, w_b, cov_b = covered_pair(sa, fa, ta, sb, fb, tb, "B")
            emit({"side": "pair", "A": an, "B": bn, "A_ms_alone_share": round(ta, 4), "A_ms_beside_B": round(ta_pair, 4), "A_tf_beside_B": round(flop / ta_pair / 1e9, 1),
                  "B_ms_alone_share": round(tb, 4), "B_ms_beside_A": round(tb_pair, 4), "B_tbs_beside_A": round(nbytes / tb_pair / 1e9, 3), "clock_beside": clk,
                  "watts": w_a, "watts_b_run": w_b, "covered": cov_a > 0 and cov_b > 0})
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/{tag}_overlap_probe_G{G}.jsonl", "w") as f:
        for dct in out:
            f.write(json.dumps(dct) + "\n")


if __name__ == "__main__":
    main()
