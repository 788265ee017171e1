"""N > 1 path on CPU: two gloo ranks exercise pymf_amd.dist and the sharded formulation."""
import os
import socket
import subprocess
import sys

from conftest import ROOT
from pymf_amd.dist import shard_rows


def test_shard_rows_partition():
    for m in (1, 7, 64, 1048576, 1000003):
        for size in (1, 2, 3, 4, 8):
            spans = [shard_rows(m, r, size) for r in range(size)]
            assert spans[0][0] == 0 and spans[-1][1] == m
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            lens = [hi - lo for lo, hi in spans]
            assert max(lens) - min(lens) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_world():
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode())
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out)
        assert "rank %d ok" % rank in out
