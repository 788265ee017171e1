"""Build libpymf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is tied to its sources by a SHA-256 over every file that goes into it (csrc/*.h, csrc/*.hip,
include/pymf_hip.h) and the compiler flags, stored next to the binary (libpymf_hip.so.srchash):
`up_to_date()` and `pymf_amd._lib.load()` compare it with the sources at hand, so a binary built from other
sources is rebuilt by build() and REFUSED by load() -- a stale library cannot pass the tests silently."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "libpymf_hip.so")
STAMP = OUT + ".srchash"
DEPS = sorted(f for f in os.listdir(HERE) if f.endswith((".h", ".hip"))) + \
       [os.path.join("..", "..", "include", "pymf_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result",
         # keep MFMA accumulators selectable from the whole 512-entry VGPR/AGPR file: the fused
         # kernel holds 352 accumulator registers, more than the 256 AGPRs alone
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def source_hash():
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for d in DEPS:
        p = os.path.join(HERE, d)
        h.update(b"\0" + d.encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def built_hash():
    try:
        with open(STAMP) as f:
            return f.read().strip()
    except OSError:
        return None


def up_to_date():
    return os.path.exists(OUT) and built_hash() == source_hash()


UNITS = ["pmf_nnls_quad_tu.hip", "pmf_api.hip", "pmf_nnls_tu.hip", "pmf_fused_tu.hip"]   # slowest first     # translation units, compiled side by side (no device code crosses them)


def build(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    want = source_hash()
    if os.path.exists(STAMP):
        os.remove(STAMP)
    cflags = [f for f in FLAGS if f != "-shared"]
    objs, procs = [], []
    for u in UNITS:
        obj = os.path.join(HERE, u.replace(".hip", ".o"))
        cmd = [hipcc] + cflags + ["-c", os.path.join(HERE, u), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd, cwd=HERE)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            for _, other in procs:
                if other.poll() is None:
                    other.kill()
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT, "-lrccl"]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link, cwd=HERE)
    for o in objs:
        os.remove(o)
    if source_hash() != want:
        raise RuntimeError("sources changed while libpymf_hip.so was being built: build again")
    with open(STAMP, "w") as f:
        f.write(want + "\n")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
