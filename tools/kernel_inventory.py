"""Inventory of the device kernels inside a built library: name, VGPRs, AGPRs, SGPRs, scratch bytes per lane, static LDS -- read from the code objects themselves.

    python tools/kernel_inventory.py [ant-multi-modal-framework_amd/lib/libantmmf_hip.so] [--json out.json] [--grep k64r]

hipcc embeds one clang offload bundle per translation unit in the .hip_fatbin section; every gfx950 entry is an ELF whose AMDGPU metadata note (msgpack, printed
as YAML by llvm-readelf --notes) lists each kernel's resource usage.  Used by tests/test_host_logic.py::test_product_kernel_resources (a register-allocation
regression -- a kernel of the step picking up scratch, or losing a wave of occupancy -- shows up here without a GPU) and for the kernel counts quoted in DESIGN.md."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path, arch="gfx950"):
    data = open(path, "rb").read()
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", data, i + len(MAGIC))[0]
        q = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if arch in triple and size:
                yield data[i + off:i + off + size]
        pos = i + len(MAGIC)


def kernels(path):
    out = {}
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count:", "\n" + txt)[1:]:
            blk = ".agpr_count:" + blk
            g = lambda key, d=0: (re.search(r"\.%s:\s*(\S+)" % key, blk) or [None, d])[1]
            name = g("name", "?")
            out[name] = dict(vgpr=int(g("vgpr_count")), agpr=int(blk.split(".agpr_count:")[1].split()[0]), sgpr=int(g("sgpr_count")),
                             scratch=int(g("private_segment_fixed_size")), lds=int(g("group_segment_fixed_size")), spill_vgpr=int(g("vgpr_spill_count")))
    return out


def demangle(names):
    import shutil

    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not tool:
        return {n: n for n in names}
    p = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.splitlines()))


def main():
    skip = {i + 1 for i, a in enumerate(sys.argv) if a in ("--grep", "--json")}
    args = [a for i, a in enumerate(sys.argv) if i > 0 and not a.startswith("--") and i not in skip]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = args[0] if args else os.path.join(root, "ant-multi-modal-framework_amd", "lib", "libantmmf_hip.so")
    ks = kernels(path)
    dm = demangle(list(ks))
    pat = sys.argv[sys.argv.index("--grep") + 1] if "--grep" in sys.argv else None
    rows = sorted(((re.sub(r"^void ", "", dm[n]).split("(")[0], v) for n, v in ks.items()), key=lambda r: r[0])
    for n, v in rows:
        if pat is None or pat in n:
            print(f"{n:90s} vgpr {v['vgpr']:3d} agpr {v['agpr']:3d} sgpr {v['sgpr']:3d} scratch {v['scratch']:4d} lds {v['lds']}")
    print(f"# {len(ks)} kernels in {os.path.basename(path)}; with scratch: {sum(1 for v in ks.values() if v['scratch'])}; using AGPRs: {sum(1 for v in ks.values() if v['agpr'])}")
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump({n: v for n, v in rows}, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
