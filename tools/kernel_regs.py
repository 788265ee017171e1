#!/usr/bin/env python3
"""Register / LDS / scratch footprint of every gfx950 kernel in libpymf_hip.so (or another fat binary).

    python tools/kernel_regs.py [path.so] [substring ...]      # substrings filter the demangled names
Reads the code object out of the clang offload bundle and its AMDGPU metadata notes (llvm-readelf)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_objects(so):
    """Every gfx950 code object of the file (one offload bundle per translation unit)."""
    data = open(so, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(b"__CLANG_OFFLOAD_BUNDLE__", pos)
        if i < 0:
            break
        pos = i + 24
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, s, ts = struct.unpack_from("<QQQ", data, off)
            off += 24
            t = data[off:off + ts].decode(errors="replace")
            off += ts
            if "gfx950" in t and s > 0:
                out.append(data[i + o:i + o + s])
    if not out:
        raise SystemExit("no gfx950 code object in " + so)
    return out


def kernels(so):
    txt = ""
    for co in code_objects(so):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt += subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    rows = []
    for k in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
        k = ".agpr_count:" + k

        def g(key):
            m = re.search(r"\.%s:\s+(\S+)" % key, k)
            return m.group(1) if m else "?"
        rows.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=g("agpr_count"), lds=g("group_segment_fixed_size"),
                         scratch=g("private_segment_fixed_size"), wg=g("max_flat_workgroup_size")))
    dem = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for r, d in zip(rows, dem):
        r["dem"] = re.sub(r"^void ", "", re.sub(r"\(.*", "", d))
    return rows


if __name__ == "__main__":
    args = sys.argv[1:]
    so = args.pop(0) if args and args[0].endswith((".so", ".co")) or (args and os.path.exists(args[0])) else \
        os.path.join(ROOT, "pymf_amd", "csrc", "libpymf_hip.so")
    rows = kernels(so)
    print("%d kernels in %s" % (len(rows), so))
    for r in sorted(rows, key=lambda r: r["dem"]):
        if not args or any(a in r["dem"] for a in args):
            print("%-78s vgpr %3s agpr %3s lds %6s scratch %4s wg %4s" % (r["dem"][:78], r["vgpr"], r["agpr"], r["lds"], r["scratch"], r["wg"]))
