#!/usr/bin/env python3
"""More seeds of the multi-rank call-sequence sweep (tests/_dist_sequence_worker.py) than the GPU suite runs:
python3 tests/sweeps/run_dist_sequences.py [first_seed] [n_seeds] [ranks] [transport]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import test_dist_ranks as t

s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4
size = int(sys.argv[3]) if len(sys.argv) > 3 else 2
transport = sys.argv[4] if len(sys.argv) > 4 else "ipc"
bad = 0
for seed in range(s0, s0 + ns):
    try:
        t._spawn_workers("_dist_sequence_worker.py", size, extra_args=["--seed", str(seed), "--cases", "16"],
                         extra_env={"PYMF_DIST_TRANSPORT": transport, "LOCAL_RANK": "0"})
        print("seed %d ok" % seed, flush=True)
    except AssertionError as e:
        bad += 1
        print("seed %d FAILED:\n%s" % (seed, "\n".join(str(e).splitlines()[-12:])), flush=True)
print("bad %d" % bad)
