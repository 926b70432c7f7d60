"""Audit of the hand-ordered registers of gemm_nt_k64r_kernel in hipcc's output (run after every edit of the kernel):
   python tools/k64r_audit.py <file.s>      (file.s from `hipcc -save-temps`, any translation unit that holds the kernel)
The residual vectors are loaded by inline asm, so hipcc believes their destination registers are written when the asm statement ends; it may copy,
spill or reuse them before the data lands.  For every such load this script walks the instruction text from the load to the fence that retires it
(`s_waitcnt vmcnt(N)` with the N of tools/k64r_ladder.py's INIT table, across the tile loop's back edge for the loads issued in the last K-tile) and
reports any instruction that names one of the destination registers.  Also checks: no scratch, 192 MFMAs, no v_accvgpr moves."""
import re
import sys

INIT = {0: 30, 1: 22, 2: 14, 3: 6}


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit(lines, name, gate=False):
    K = [l.strip() for l in lines]
    # instructions written by the kernel's inline asm sit between ;;#ASMSTART / ;;#ASMEND: the residual loads under audit are those (round 5: the cell tail behind the
    # walk adds compiler-generated loads of its own -- a bias vector, 8-byte residual pieces -- which hipcc tracks itself)
    in_asm, asm_line = False, set()
    for i, l in enumerate(K):
        if l.startswith(";;#ASMSTART"):
            in_asm = True
        elif l.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm:
            asm_line.add(i)
    code = [(i, l) for i, l in enumerate(K) if l and not l.startswith(";") and not l.startswith(".") or l.startswith(".LBB")]
    text = [l for _, l in code]
    from_asm = [i in asm_line for i, _ in code]
    nm = sum(1 for l in text if l.startswith("v_mfma"))
    bad = 0
    if any("scratch_" in l for l in text):
        print(name, "SCRATCH in use"); bad += 1
    if any(l.startswith("v_accvgpr") for l in text):
        print(name, "v_accvgpr moves"); bad += 1
    loads = [i for i, l in enumerate(text) if l.startswith("global_load_dwordx4") and from_asm[i]]
    if not loads:
        print(name, "mfma", nm, "no residual loads"); return bad
    # loop header = target of the last backward s_branch
    labels = {l[:-1].split(":")[0]: i for i, l in enumerate(text) if l.startswith(".LBB")}
    back = [(i, labels[m.group(1)]) for i, l in enumerate(text) for m in [re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)] if m and m.group(1) in labels and labels[m.group(1)] < i]
    tile_back = max(back, key=lambda p: p[0] - p[1])   # the widest backward branch = the tile loop
    end_i, head_i = tile_back
    if gate:   # EPI 8: 16 gate loads in the last K-tile (quarter q one phase ahead of its store), each retired by vmcnt(2); quarter 3 is consumed behind the back edge
        assert len(loads) == 16, (name, len(loads))
        groups = [("TE gate q%d" % q, loads[4 * q:4 * q + 4], 2, q == 3) for q in range(4)]
    else:
        assert len(loads) == 28, (name, len(loads))   # 12 prologue + 4 (T0) + 12 (TE)
        groups = [("prologue q%d" % q, loads[4 * q:4 * q + 4], 0, False) for q in range(3)]
        groups.append(("T0 q3", loads[12:16], INIT[3], False))
        groups += [("TE q%d" % q, loads[16 + 4 * q:20 + 4 * q], INIT[q], True) for q in range(3)]
    for tag, idxs, n, wraps in groups:
        dest = set()
        for i in idxs:
            dest |= regs(re.findall(r"v\[\d+:\d+\]", text[i])[0])
        last = idxs[-1]
        # instruction ranges to scan
        spans = []
        def scan_until_fence(a, b):
            for j in range(a, b):
                m = re.match(r"s_waitcnt vmcnt\((\d+)\)", text[j])
                if m and int(m.group(1)) == n and (tag.startswith("prologue") or j > a):
                    return j
            return None
        if wraps:
            f = scan_until_fence(head_i, end_i)
            spans = [(last + 1, end_i + 1), (head_i, f)]
        else:
            f = scan_until_fence(last + 1, len(text))
            spans = [(last + 1, f)]
        if f is None:
            print(name, tag, "fence vmcnt(%d) not found" % n); bad += 1; continue
        for a, b in spans:
            for j in range(a, b):
                if text[j].startswith("global_load_dwordx4") and j in idxs:
                    continue
                used = set()
                for tk in re.findall(r"v\[\d+:\d+\]|v\d+", text[j]):
                    used |= regs(tk)
                if used & dest:
                    print(name, tag, "TOUCHED before its fence:", text[j]); bad += 1
    print(name, "mfma", nm, "loads", len(loads), "problems", bad)
    return bad


if __name__ == "__main__":
    S = open(sys.argv[1]).read().split("\n")
    total = 0
    for e in (0, 1, 2, 3, 8, 10):   # (10 = 8 + the column sums of the stored rows)
        st = [i for i, l in enumerate(S) if l.startswith("_Z19gemm_nt_k64r_kernelILi%dE" % e)]
        if not st:
            continue
        en = [i for i, l in enumerate(S) if i > st[0] and ".amdhsa_kernel" in l][0]
        total += audit(S[st[0]:en], "k64r<%d>" % e, gate=(e in (8, 10)))
    sys.exit(1 if total else 0)
