"""N > 1 path on CPU: world_size-2 (and 4) ranks exercise pymf_amd.dist -- the socket rendezvous that
replaced torch.distributed/gloo in the product -- and the row-sharded formulation; once under the
driver's own launcher (`python -m torch.distributed.run`, which only provides the env)."""
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


import pytest


@pytest.mark.parametrize("size", [2, 4])
def test_rank_world(size):
    port = _free_port()
    procs = []
    for rank in range(size):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size),
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


def test_world_under_the_torchrun_launcher():
    """The driver launches bench.py with `python -m torch.distributed.run`: the workers must find
    each other from the env it sets (and never import torch themselves)."""
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_dist_worker.py"), "--assert-no-torch"]
    p = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    assert "rank 0 ok" in out and "rank 1 ok" in out


def _run_world(size, args):
    port = _free_port()
    procs = []
    for rank in range(size):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py")] + list(args),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode())
    return procs, outs


def test_a_collective_called_on_one_rank_only_fails_loudly_on_every_rank():
    """the TCP star carries (sequence number, tag) in every frame; a rank-0-only collective paired with
    the peers' next unrelated one raises CollectiveMismatch everywhere -- no mixed payloads, no hang."""
    procs, outs = _run_world(3, ["--mismatch"])
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "rank %d ok (mismatch caught)" % rank in out, "rank %d:\n%s" % (rank, out)


def test_a_rank_stuck_in_a_start_up_step_ends_its_process_with_status_70():
    """ncclCommInitRank / the IPC self-test are time-boxed (dist.Watchdog): the stuck rank leaves
    with status 70 and a message, the launcher sees a failed job instead of a hang."""
    procs, outs = _run_world(2, ["--watchdog"])
    assert procs[0].returncode == 0 and "rank 0 ok" in outs[0], outs[0]
    assert procs[1].returncode == 70 and "still in 'the step under test'" in outs[1], outs[1]


def _spawn_workers(script, size, extra_args=(), extra_env=None, timeout=600):
    port = _free_port()
    procs = []
    for rank in range(size):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", script)] + list(extra_args),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode())
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out)
        assert "rank %d ok" % rank in out
    return outs


@pytest.mark.parametrize("size", [2, 3])
def test_class_level_multi_rank_path_on_cpu(size):
    """pymf_amd.NMF under WORLD_SIZE > 1 with a test double for the device context: the host logic
    (global row span, W0 slice of rank 0's stream, replicated H0, shapes) against the unsharded oracle."""
    _spawn_workers("_dist_class_worker.py", size, extra_args=["--fake"],
                   extra_env={"PYMF_DIST_TRANSPORT": "host"})       # no RCCL id: there is no GPU here


@pytest.mark.gpu
def test_class_level_multi_rank_path_on_one_gpu():
    """Two ranks sharing GPU 0, the real library, cross-rank sums through the host transport
    (PYMF_DIST_TRANSPORT=host -> pmf_set_host_allreduce): NMF / SNMF / RNMF / NNDSVD objects fed
    their row blocks must reproduce the unsharded oracle."""
    _spawn_workers("_dist_class_worker.py", 2, extra_env={"PYMF_DIST_TRANSPORT": "host", "LOCAL_RANK": "0"})


@pytest.mark.gpu
@pytest.mark.parametrize("size", [2, 4, 8])
def test_one_shot_ipc_allreduce_processes_on_one_gpu(size):
    """the one-shot all-reduce (pmf_ipc.h: every rank writes its partial of (W^T V | W^T W) into
    every peer's IPC-mapped receive area and adds the N partials in rank order) with two, four and EIGHT processes (the node size the receive areas are laid out for) sharing GPU 0 --
    RCCL refuses that set-up ("Duplicate GPU detected").  The class-level worker must reproduce the unsharded oracle,
    keep H bit-identical across the ranks, leave an early-exiting loop at the same iteration on both ranks, AND give
    the same bits as the host transport (both add in rank order): the worker prints digests that are compared here."""
    outs_ipc = _spawn_workers("_dist_class_worker.py", size, extra_args=["--digest"],
                              extra_env={"PYMF_DIST_TRANSPORT": "ipc", "LOCAL_RANK": "0"})
    outs_host = _spawn_workers("_dist_class_worker.py", size, extra_args=["--digest"],
                               extra_env={"PYMF_DIST_TRANSPORT": "host", "LOCAL_RANK": "0"})
    # round 5: by default the loop's exchange is FOLDED into the slab-reduce launch (push) and the H-step launch (wait + sum in
    # rank order); PYMF_DIST_FOLD=0 runs it as round 4's launch of its own -- all three must give the same bits
    outs_nofold = _spawn_workers("_dist_class_worker.py", size, extra_args=["--digest"],
                                 extra_env={"PYMF_DIST_TRANSPORT": "ipc", "LOCAL_RANK": "0", "PYMF_DIST_FOLD": "0"})
    for a, b, c in zip(outs_ipc, outs_host, outs_nofold):
        da = [l for l in a.splitlines() if l.startswith("digest ")]
        db = [l for l in b.splitlines() if l.startswith("digest ")]
        dc = [l for l in c.splitlines() if l.startswith("digest ")]
        assert da and da == db and da == dc, (da, db, dc)
        assert any("one-shot IPC all-reduce" in l and "ipc 0 " not in l and "(0 of them folded" not in l
                   for l in a.splitlines() if l.startswith("collective ")), a
        assert any("one-shot IPC all-reduce" in l and "ipc 0 " not in l and "(0 of them folded" in l
                   for l in c.splitlines() if l.startswith("collective ")), c
        assert all("one-shot" not in l for l in b.splitlines() if l.startswith("collective ")), b


# ---- who may join the rendezvous (pymf_amd/dist.py: bind, challenge-response, frame cap) ----------------
def test_non_loopback_rendezvous_needs_a_secret(monkeypatch):
    from pymf_amd import dist
    monkeypatch.delenv("PYMF_DIST_SECRET", raising=False)
    with pytest.raises(RuntimeError, match="PYMF_DIST_SECRET"):
        dist._key("10.1.2.3", "29500", 2, "10.1.2.3")
    monkeypatch.setenv("PYMF_DIST_SECRET", "s3cret")
    k1 = dist._key("10.1.2.3", "29500", 2, "10.1.2.3")
    monkeypatch.setenv("PYMF_DIST_SECRET", "other")
    assert k1 != dist._key("10.1.2.3", "29500", 2, "10.1.2.3")
    monkeypatch.delenv("PYMF_DIST_SECRET")
    assert dist._key("127.0.0.1", "29500", 2, "127.0.0.1")           # loopback: derived from the launcher's env


def test_multi_node_launch_with_a_loopback_master_addr_fails_at_once(monkeypatch):
    """Advisor (round 3): MASTER_ADDR = the host's own name resolves to 127.0.1.1 on Debian/Ubuntu -- a multi-node job would
    wait 300 s for ranks that can never connect."""
    from pymf_amd import dist
    for kk, vv in (("WORLD_SIZE", "4"), ("LOCAL_WORLD_SIZE", "2"), ("RANK", "0"), ("LOCAL_RANK", "0"),
                   ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(_free_port()))):
        monkeypatch.setenv(kk, vv)
    monkeypatch.setattr(dist, "_PEERS", None)
    with pytest.raises(RuntimeError, match="loopback"):
        dist.init_from_env()


def test_wrong_key_and_strangers_cannot_join_and_frames_are_capped():
    """Rank 0's accept loop against: a peer with the wrong key, a peer that sends garbage, then the real rank 1.
    Only the last one is admitted; afterwards an over-long frame header is refused."""
    import struct
    import threading
    from pymf_amd import dist
    key, bad = b"k" * 32, b"x" * 32
    port = _free_port()
    result = {}

    def serve():
        try:
            result["peers"] = dist._serve("127.0.0.1", port, 2, key, timeout=30.0)
        except Exception as e:       # pragma: no cover
            result["err"] = e

    th = threading.Thread(target=serve)
    th.start()
    with pytest.raises(RuntimeError):
        dist._join("127.0.0.1", port, 1, bad, timeout=1.0)          # never acknowledged
    s = socket.create_connection(("127.0.0.1", port), timeout=5.0)
    s.sendall(b"GET / HTTP/1.0\r\n\r\n" + b"\0" * 64)                 # a stranger
    s.close()
    mine = dist._join("127.0.0.1", port, 1, key, timeout=20.0)
    th.join(timeout=30.0)
    assert "err" not in result and list(result["peers"]) == [1]
    try:
        mine[0].sendall(struct.pack("<Q", dist._MAX_FRAME + 1))
        with pytest.raises(ConnectionError, match="PYMF_DIST_MAX_FRAME"):
            dist._recv(result["peers"][1])
    finally:
        mine[0].close()
        result["peers"][1].close()
        dist._LISTENER.close()
        dist._LISTENER = None


@pytest.mark.gpu
def test_class_level_multi_rank_path_over_rccl():
    """One RCCL rank per GPU (needs a box with >= 2 GPUs; the 1-GPU boxes skip it): the class-level worker with the
    default transport -- every cross-rank sum of the refactored allreduce_sum() helper goes through ncclAllReduce:
    (P | S) per iteration, ||V||^2, RNMF's error, the float64 V^T V of the Gram-space SNMF loop, NNDSVD's Gram and
    split norms."""
    from pymf_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs >= 2 GPUs: one RCCL rank per GPU")
    _spawn_workers("_dist_class_worker.py", 2)


@pytest.mark.gpu
@pytest.mark.parametrize("size,transport,seed", [(2, "ipc", 1), (2, "host", 2), (3, "ipc", 3)])
def test_random_call_sequences_under_a_multi_rank_world(size, transport, seed):
    """Round 4: the call-sequence sweep (tests/sweeps/fuzz_sequences.py found three stale-state bugs at one rank) with the rows
    sharded over processes that share GPU 0: factorize() with every flag combination, the hooks, frobenius_norm(), W / H / data
    edits -- gathered W, the replicated H (bit-identical across the ranks) and ferr against the unsharded oracle after every step."""
    _spawn_workers("_dist_sequence_worker.py", size, extra_args=["--seed", str(seed), "--cases", "14"],
                   extra_env={"PYMF_DIST_TRANSPORT": transport, "LOCAL_RANK": "0"})
