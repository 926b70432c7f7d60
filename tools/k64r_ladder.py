"""Counted-vmcnt ladders of gemm_nt_k64r_kernel (csrc/gemm.hip): replays ONE wave's vector-memory issue order over several output tiles and, for every
wait of the K loop, counts the operations issued after the operation that wait has to cover (gfx9 retires a wave's vector-memory operations in issue
order, so `s_waitcnt vmcnt(N)` with N = that count is exact; anything smaller is safe and waits longer, anything larger is a race).  Per (role, phase) the
MINIMUM over all occurrences (first tile, later tiles, nk = 3 .. 9) is printed, so one constant serves every position.  Run: python tools/k64r_ladder.py

Issue order per phase ph of a K-tile (role T0 = first K-tile of an output tile, TE = last, TR = the others):
    DMA pieces 2 ph, 2 ph + 1
    hook:  T0, ph 0: [4 stores: row quarter 3 of the previous tile] [4 residual loads: quarter 3 of this tile]
           T0, ph q: wait for the residual vectors of quarter q (issued in the previous tile's TE, or in T0 ph 0 for q = 3), initialise its accumulators
           TE, ph 0: [1 DMA: the next tile's bias strip]
           TE, ph q + 1 (q = 0 .. 2): 4 stores (quarter q; 8 for the two-output activation epilogue) then 4 residual loads (quarter q of the NEXT tile)
    end-of-phase wait: phase ph + 1's fragment reads  (ph 0 .. 2: P quarter ph + 1 of this K-tile = piece ph + 1 of the previous K-tile;
                                                       ph 3: P quarter 0 of the next K-tile = piece 0 of this K-tile; its Q pieces are older)
The prologue (K-tile 0 whole, Q of K-tile 1, bias strip, residual quarters 0 - 2 of the first tile) is drained with vmcnt(0) and not modelled."""


def ladder(res, bias, nks=(3, 4, 5, 6, 9), tiles=3, stores=4):
    waits, inits, biasw = {}, {}, {}
    for nk in nks:
        ops = []
        pos = {}

        def role(t):
            return "T0" if t == 0 else ("TE" if t == nk - 1 else "TR")
        gt = 0
        rv_last = {}      # (tile, quarter) -> index of its last load
        bias_pos = {}
        for tile in range(tiles):
            for t in range(nk):
                k = role(t)
                for ph in range(4):
                    for idx in (2 * ph, 2 * ph + 1):
                        pos[(gt, idx)] = len(ops); ops.append("piece")
                    if k == "T0" and ph == 0:
                        if tile > 0: ops += ["store"] * stores
                        if res:
                            ops += ["resload"] * 4; rv_last[(tile, 3)] = len(ops) - 1
                        if bias and tile in bias_pos:   # the bias strip is read behind this point
                            biasw["T0"] = min(biasw.get("T0", 99), len(ops) - 1 - bias_pos[tile])
                    if k == "T0" and res and (tile, ph) in rv_last:
                        inits[ph] = min(inits.get(ph, 99), len(ops) - 1 - rv_last[(tile, ph)])
                    if k == "TE" and ph == 0 and bias:
                        ops.append("biasdma"); bias_pos[tile + 1] = len(ops) - 1
                    if k == "TE" and ph >= 1:
                        ops += ["store"] * stores
                        if res:
                            ops += ["resload"] * 4; rv_last[(tile + 1, ph - 1)] = len(ops) - 1
                    need = {0: (gt - 1, 1), 1: (gt - 1, 2), 2: (gt - 1, 3), 3: (gt, 0)}[ph]
                    if need in pos:
                        waits[(k, ph)] = min(waits.get((k, ph), 99), len(ops) - 1 - pos[need])
                gt += 1
    return waits, inits, biasw


def ladder_gate(nks=(3, 4, 5, 6, 9), tiles=3, stores=4):
    """EPI 8 (out = acc * gate, round 5): no bias, no residual.  The gate vectors of row quarter q (4 loads) are requested ONE phase before the store of that quarter, into
    the registers the previous quarter's vectors have just left (two quarters in flight -- two phases of lead -- put the kernel at 256 VGPRs + 24 B of scratch):
        TE ph 0: gate loads q0
        TE ph q + 1 (q = 0 .. 2): [wait q] multiply + 4 stores q, gate loads q + 1
        T0 ph 0 of the next tile: [wait q3] multiply + 4 stores q3
    Returns W[(role, ph)] and GINIT[q] = operations issued behind quarter q's last gate load at the point where it is consumed."""
    waits, ginit = {}, {}
    for nk in nks:
        ops, pos, gl = [], {}, {}

        def role(t):
            return "T0" if t == 0 else ("TE" if t == nk - 1 else "TR")
        gt = 0
        for tile in range(tiles):
            for t in range(nk):
                k = role(t)
                for ph in range(4):
                    for idx in (2 * ph, 2 * ph + 1):
                        pos[(gt, idx)] = len(ops); ops.append("piece")

                    def consume(q, tl):
                        if (tl, q) in gl:
                            ginit[q] = min(ginit.get(q, 99), len(ops) - 1 - gl[(tl, q)])

                    def load(q, tl):
                        ops.extend(["gateload"] * 4); gl[(tl, q)] = len(ops) - 1
                    if k == "T0" and ph == 0 and tile > 0:
                        consume(3, tile - 1); ops.extend(["store"] * stores)
                    if k == "TE":
                        if ph == 0:
                            load(0, tile)
                        else:
                            consume(ph - 1, tile); ops.extend(["store"] * stores); load(ph, tile)
                    need = {0: (gt - 1, 1), 1: (gt - 1, 2), 2: (gt - 1, 3), 3: (gt, 0)}[ph]
                    if need in pos:
                        waits[(k, ph)] = min(waits.get((k, ph), 99), len(ops) - 1 - pos[need])
                gt += 1
    return waits, ginit


if __name__ == "__main__":
    for (res, bias) in ((0, 0), (0, 1), (1, 0), (1, 1)):
        w, i, b = ladder(res, bias)
        print("EPI=%d (RES=%d BIAS=%d)" % (bias + 2 * res, res, bias))
        print("   W = {" + ", ".join("{" + ", ".join(str(w[(k, ph)]) for ph in range(4)) + "}" for k in ("T0", "TR", "TE")) + "}   (T0, TR, TE)")
        if i: print("   INIT = {%s}" % ", ".join(str(i[q]) for q in range(4)))
        if b: print("   BIASW = %d" % b["T0"])
    w, i, b = ladder(0, 1, stores=8)
    print("EPI=5 (bias + activation, two outputs: 8 stores per row quarter)")
    print("   W = {" + ", ".join("{" + ", ".join(str(w[(k, ph)]) for ph in range(4)) + "}" for k in ("T0", "TR", "TE")) + "}   (T0, TR, TE)")
    print("   BIASW = %d" % b["T0"])
    w, gi = ladder_gate()
    print("EPI=8 (gate: out = acc * gate)")
    print("   W = {" + ", ".join("{" + ", ".join(str(w[(k, ph)]) for ph in range(4)) + "}" for k in ("T0", "TR", "TE")) + "}   (T0, TR, TE)")
    print("   GINIT = {%s}" % ", ".join(str(gi[q]) for q in range(4)))
