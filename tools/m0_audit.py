"""Audit of M0 around the hand-issued LDS-DMA pieces of the attention kernels:  python tools/m0_audit.py <attention device .s>

`attn_dma_piece` sets M0 and issues `global_load_lds_dwordx4` from ONE inline-asm statement that names "m0" as clobbered.  Checked in hipcc's output, per kernel:
  * every asm-issued `global_load_lds_dwordx4` has its own `s_mov_b32 m0, ...` in front of it inside the same asm block;
  * every COMPILER-issued reader of M0 (its own `global_load_lds_*` / `buffer_load ... lds`, `s_movrel*` / `v_movrel*`, `s_sendmsg`) finds a compiler-issued write of M0
    between itself and the nearest asm block above it in program text (i.e. the compiler never relies on a value of M0 across one of the asm statements).
Prints one line per kernel that issues asm LDS-DMA; exit code = number of problems."""
import re
import sys

M0_READ = re.compile(r"^(global_load_lds_|buffer_load_\w+ .*\blds\b|s_movrel|v_movrel|s_sendmsg\b)")
M0_WRITE = re.compile(r"^s_(mov_b32|add_u32|add_i32|lshl_b32|or_b32|and_b32) m0\b")


def main():
    txt = open(sys.argv[1]).read().split("\n")
    bad, kernel, in_asm = 0, None, False
    stats = {}
    last_asm_dma, compiler_write_since = None, True
    pending_mov = False
    for raw in txt:
        l = raw.strip()
        m = re.match(r"^(_Z\w+):", l)
        if m and not l.startswith(".L"):
            kernel, in_asm, compiler_write_since, pending_mov = m.group(1), False, True, False
            continue
        if l.startswith(";;#ASMSTART"):
            in_asm, pending_mov = True, False
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not l or l.startswith(";") or l.startswith("."):
            continue
        st = stats.setdefault(kernel, dict(asm_dma=0, compiler_m0_reads=0, problems=0))
        if in_asm:
            if l.startswith("s_mov_b32 m0"):
                pending_mov = True
            elif l.startswith("global_load_lds_dwordx4"):
                st["asm_dma"] += 1
                if not pending_mov:
                    st["problems"] += 1
                pending_mov = False
                compiler_write_since = False      # from here on M0 holds the asm's value
        else:
            if M0_WRITE.match(l):
                compiler_write_since = True
            elif M0_READ.match(l):
                st["compiler_m0_reads"] += 1
                if not compiler_write_since:
                    st["problems"] += 1
    for k, st in stats.items():
        if st["asm_dma"]:
            print(f"m0 audit {k[:90]}: asm LDS-DMA pieces {st['asm_dma']}, compiler readers of M0 {st['compiler_m0_reads']}, problems {st['problems']}")
            bad += st["problems"]
    return bad


if __name__ == "__main__":
    sys.exit(main())
