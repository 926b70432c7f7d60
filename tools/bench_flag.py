"""bench.py with module flags of antmmf.hip.functional changed first (same-box A/Bs of step-level choices):

    python tools/bench_flag.py ATTN_BWD_SUMS=0 -- --no-cpu-baseline --steps 5

Everything after `--` goes to bench.py unchanged; the JSON line is bench.py's own, the flags are echoed on stderr."""
import os
import runpy
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "ant-multi-modal-framework_amd"))
sys.path.insert(0, ROOT)


def main():
    args = sys.argv[1:]
    cut = args.index("--") if "--" in args else len(args)
    flags, rest = args[:cut], args[cut + 1:]
    from antmmf.hip import functional as HF
    for f in flags:
        name, val = f.split("=", 1)
        if not hasattr(HF, name):
            raise SystemExit(f"antmmf.hip.functional has no flag {name}")
        cur = getattr(HF, name)
        setattr(HF, name, type(cur)(int(val)) if isinstance(cur, (bool, int)) else type(cur)(val))
        print(f"[bench_flag] {name} = {getattr(HF, name)!r}", file=sys.stderr)
    sys.argv = [os.path.join(ROOT, "bench.py")] + rest
    runpy.run_path(sys.argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
