"""Build libpymf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "pmf_api.hip")
OUT = os.path.join(HERE, "libpymf_hip.so")
DEPS = sorted(f for f in os.listdir(HERE) if f.endswith((".h", ".hip"))) + \
       [os.path.join("..", "..", "include", "pymf_hip.h")]


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    for d in DEPS:
        p = os.path.join(HERE, d)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return False
    return True


def build(force=False, verbose=True):
    if not force and up_to_date():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-result",
           # keep MFMA accumulators selectable from the whole 512-entry VGPR/AGPR file: the fused
           # kernel holds 352 accumulator registers, more than the 256 AGPRs alone
           "-mllvm", "-amdgpu-mfma-vgpr-form=1", SRC, "-o", OUT, "-lrccl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=HERE)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
