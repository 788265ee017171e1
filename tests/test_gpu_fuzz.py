"""Seeded samples of the randomised sweeps under tests/sweeps/ (odd shapes, dtypes, layouts, every kernel class) as part of the
GPU suite: each sweep compares the library with the float64 oracles case by case and prints `bad N` at the end."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,args", [("fuzz_nmf.py", ["2024", "40"]),          # NMF / SNMF / BNMF / RNMF: fused, split, coop, tiled
                                       ("fuzz_als.py", ["2025", "25"]),          # NMFALS / NMFNNLS up to 64 bases
                                       ("fuzz_als.py", ["2026", "10", "wide"]),  # ... 65-128 bases (k_nnqp_wave)
                                       ("fuzz_tiled.py", ["2027", "20"]),        # two-pass kernels, stream forms on and off: bit-identical
                                       ("fuzz_wide.py", ["2028", "12"]),         # every inverse kernel, > 128 bases
                                       ("fuzz_misc.py", ["2029"]),               # streamed passes, NNDSVD
                                       ("fuzz_sequences.py", ["2030", "50"]),    # random call sequences on one object against its oracle twin
                                       ("fuzz_sequences.py", ["2031", "30"]),
                                       ("fuzz_abi_sequences.py", ["2040", "50"]),  # ... of C-ABI calls on one context, resident and streamed passes mixed
                                       ("threads_probe.py", []),                 # 24 host threads, one object each: bit-identical to sequential
                                       ("stress_leaks.py", []),                  # 180 contexts, 20 000-iteration loops: memory comes back
                                       ("huge_probe.py", ["20971520"]),          # V of 5.4e9 elements (> 2^32): W rows vs the oracle, one-pass vs two-pass
                                       ("huge_probe2.py", ["20971520"]),         # ... NMFALS (KKT), SNMF on it; NMF 1 024 x 1 000 000
                                       ("huge_probe3.py", ["20971520", "256"]),  # ... 256 bases: W of 5.4e9 elements on the wide-base path
                                       ("wide_scan.py", ["--quick"]),                     # 32 768 ... 1 000 000 columns: chunked accumulation chains
                                       ("degenerate_values.py", [])])            # zero / constant / low-rank data, zero bases, 1e-6 ... 1e+12
def test_seeded_sample_of_the_randomised_sweeps(tool, args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sweeps", tool)] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=600, cwd=ROOT)
    out = p.stdout.decode("utf-8", "replace")
    tail = "\n".join(out.strip().splitlines()[-15:])
    assert p.returncode == 0, tail
    assert out.strip().splitlines()[-1].strip() == "bad 0", tail


@pytest.mark.parametrize("args", [["2032", "40"], ["2033", "40"]])
def test_call_sequences_with_the_digest_of_data_beside_the_loop(args):
    """The random call sequences once more with the late data check (round 5: the digest of `data` on a second thread while
    pmf_factorize runs; stop + restore + upload + restart when the bytes changed) forced on for data of any size."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sweeps", "fuzz_sequences.py")] + args, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600, cwd=ROOT, env=dict(os.environ, PYMF_LATE_CHECK_MIN_BYTES="0"))
    out = p.stdout.decode("utf-8", "replace")
    tail = "\n".join(out.strip().splitlines()[-15:])
    assert p.returncode == 0, tail
    assert out.strip().splitlines()[-1].strip() == "bad 0", tail
