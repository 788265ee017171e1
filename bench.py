#!/usr/bin/env python3
"""bench.py -- factorize() iterations/sec of pymf's hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            [--config cfg4|cfg2|cfg3|cfg5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Default workload (BASELINE.json metric, configs[3] = "cfg4"): NMF multiplicative update on a dense
float32 V of 1,048,576 x 256 at k = 64; the rows of V and W are sharded over the N ranks (fixed
total problem => "strong" scaling), H is replicated, and one RCCL all-reduce of (W^T V | W^T W)
-- 80 KiB -- runs per iteration.  A "step" is one factorize() iteration: update_w + update_h
(compute_err=False, the pure update path; the compute_err=True rate is reported alongside).
Inputs follow BASELINE.md section 3: V = RandomState(1234).random_sample((m, n)).astype(float32),
np.random.seed(42), W0 then H0 from np.random.random.  All inputs are resident in HBM before the
timed region starts.  The secondary BASELINE configs run through the same harness and schema:
  --config cfg2   NMF     65,536 x 512,  k = 32
  --config cfg3   NMFALS  262,144 x 1024, k = 64
  --config cfg5   SNMF    scipy.sparse CSR 4,194,304 x 128 (1 % nnz), k = 128

Prints ONE JSON line on rank 0 (contract in the task description), including
  "roofline":     live HIP-event timing of the path's dominant m-sized kernel against the roofline
                  that bounds it (fp32 MFMA or HBM), with SURVEY's algorithmic flops/bytes and the
                  flops the kernel really executes
  "cpu_baseline": the NumPy oracle (oracle/, the reference's op order) timed on this host's cores.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0
RIDGE_FLOP_PER_BYTE = PEAK_F32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
PREROLL_MS_DEFAULT = 300.0   # measured (profiles/r02_ramp.md): a fresh box needs ~30 launches to reach its clock

CONFIGS = {
    # name: (algo, m, n, k, BASELINE.json configs[] index, description)
    "cfg4": ("NMF", 1048576, 256, 64, 3, "pymf.NMF multiplicative update, dense fp32 V"),
    "cfg2": ("NMF", 65536, 512, 32, 1, "pymf.NMF multiplicative update, dense fp32 V"),
    "cfg3": ("NMFALS", 262144, 1024, 64, 2, "pymf.NMFALS alternating least squares, dense fp32 V"),
    "cfg5": ("SNMF", 4194304, 128, 128, 4, "pymf.SNMF on scipy.sparse CSR V (1 % nnz)"),
}


def gen_rows(seed_state, m_total, ncols, lo, hi, chunk=65536):
    """rows [lo, hi) of RandomState.random_sample((m_total, ncols)), generated in chunks so a
    rank never holds the whole float64 matrix (the legacy stream is sequential)."""
    out = np.empty((hi - lo, ncols), dtype=np.float32)
    r = 0
    while r < hi:
        c = min(chunk, m_total - r)
        blk = seed_state.random_sample((c, ncols))
        a, b = max(r, lo), min(r + c, hi)
        if a < b:
            out[a - lo:b - lo] = blk[a - r:b - r]
        r += c
    # advance the stream past the remaining rows so later draws (H0) match the reference order
    rem = m_total - r
    while rem > 0:
        c = min(chunk, rem)
        seed_state.random_sample((c, ncols))
        rem -= c
    return out


def gen_csr(m, n, density, lo, hi, fast):
    """scipy.sparse.random(m, n, density, 'csr', float32, random_state=1234), rows [lo, hi).
    (about a minute and 4.4 GB of host memory at cfg5 size.)  fast=True: Poisson row lengths and uniform
    column draws -- same density, seconds; a shape-faithful stand-in, NOT the BASELINE stream."""
    if not fast:
        import scipy.sparse as sp
        V = sp.random(m, n, density=density, format="csr", dtype=np.float32, random_state=1234)
        V = V[lo:hi]
        return V.indptr.astype(np.int64), V.indices.astype(np.int32), V.data.astype(np.float32)
    rs = np.random.RandomState(1234)
    nnz_row = np.minimum(rs.poisson(density * n, size=m), n).astype(np.int64)
    indptr = np.concatenate([[0], np.cumsum(nnz_row)])
    nnz = int(indptr[-1])
    indices = rs.randint(0, n, size=nnz).astype(np.int32)       # duplicates add up, as in V.toarray()
    vals = rs.random_sample(nnz).astype(np.float32)
    a, b = int(indptr[lo]), int(indptr[hi])
    return (indptr[lo:hi + 1] - a).astype(np.int64), indices[a:b], vals[a:b]


def _blas_threads():
    threads = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info
        for info in threadpool_info():
            if info.get("user_api") == "blas":
                threads = int(info.get("num_threads", threads))
    except Exception:
        pass
    return threads


def _time_loop(step, budget_s, min_iters=3, max_iters=50):
    step()                                    # warm-up iteration
    t0 = time.time()
    it = 0
    while True:
        step()
        it += 1
        dt = time.time() - t0
        if (it >= min_iters and dt > budget_s * 0.5) or dt > budget_s or it >= max_iters:
            return it, dt


def cpu_baseline(cfg, m, n, k, budget_s=24.0, sample=None):
    """The oracle (NumPy restatement of the reference, its op order) on this host's cores.
    cfg4 / cfg2: FULL size, reference-default operands (float64 W/H, float32 V) = `value`, the all-float32
    variant alongside.  cfg3 / cfg5: a bounded row sample, scaled -- stated in `sample`."""
    import oracle
    algo = CONFIGS[cfg][0]
    out = {"unit": "iter/s", "cores": _blas_threads(), "kind": "port", "host_cpus": os.cpu_count() or 1}
    # (BASELINE.md section 3 says "all host cores": the BLAS build caps its pool -- OpenBLAS: 64 threads on these boxes)
    out["threads_note"] = "%d BLAS threads (the BLAS build's thread cap) of %d CPUs" % (out["cores"], out["host_cpus"])
    host = "numpy %s, host cpu_count=%d" % (np.__version__, os.cpu_count() or 1)
    if algo == "NMF":
        V = gen_rows(np.random.RandomState(1234), m, n, 0, m)
        np.random.seed(42)
        W = np.random.random((m, k))
        H = np.random.random((k, n))
        W32, H32 = W.astype(np.float32), H.astype(np.float32)

        def step64():
            oracle.nmf_update_w(V, W, H)
            oracle.nmf_update_h(V, W, H)

        def step32():
            oracle.nmf_update_w(V, W32, H32)
            oracle.nmf_update_h(V, W32, H32)
        it, dt = _time_loop(step64, budget_s * 0.7)
        # the float64-default factors after 1 + it iterations from W0 / H0: what the device is compared with at FULL size
        # (main(): `parity_full_size`); the timing loop is the parity run -- nothing is thrown away
        keep = {"iters": 1 + it, "W": W.copy(), "H": H.copy(),
                "ferr": float(oracle.frobenius_norm(V, W, H))}       # nmf.py:100-114 at full size, float64
        it32, dt32 = _time_loop(step32, budget_s * 0.3)
        out["_oracle_factors"] = keep
        out.update(value=it / dt, value_all_fp32=it32 / dt32,
                   sample="oracle NMF (reference op order, float64 W/H, float32 V) at FULL size %dx%d k=%d: "
                          "1 warm-up + %d iterations in %.2f s; all-float32 variant: %d iterations in %.2f s; %s"
                          % (m, n, k, it, dt, it32, dt32, host))
    elif algo == "NMFALS" and sample is not None:
        # The QPs timed here ARE the full-size parity check: `sample` holds, from the device's first iteration
        # at FULL size from the seeded W0 / H0, (a) 256 rows of V spread over the matrix and the device's W rows for them --
        # the row QPs of nmfals.py:85-97 depend on H0 and the row alone --, (b) for 256 columns the Hessian W1^T W1 and the
        # right-hand sides W1^T v_i over ALL 262 144 rows of the device's W1 (float64 on the host) and the device's H columns.
        ms, ns = len(sample["rows"]), len(sample["cols"])
        H0 = np.asarray(sample["H0"], dtype=np.float64)
        HAw = np.float64(np.dot(H0, H0.T))                                    # nmfals.py:93
        t0 = time.time()
        Wo = np.empty((ms, k))
        for q in range(ms):
            Wo[q] = oracle.nnqp_solve(HAw, np.float64(np.dot(-H0, sample["Vrows"][q].astype(np.float64))))   # :88-90
        tw = time.time() - t0
        t0 = time.time()
        Ho = np.empty((k, ns))
        for q in range(ns):
            Ho[:, q] = oracle.nnqp_solve(sample["HA_h"], -sample["FA_cols"][:, q])    # :73-75 (HA = W^T W, FA = -W^T v)
        th = time.time() - t0
        per_iter = tw * m / ms + th * n / ns
        Wd, Hd = sample["Wd_rows"].astype(np.float64), sample["Hd_cols"].astype(np.float64)
        out["_parity"] = {
            "iters": 1, "rows": int(ms), "cols": int(ns),
            "relW_rows": float(np.linalg.norm(Wd - Wo) / np.linalg.norm(Wo)),
            "relH_cols": float(np.linalg.norm(Hd - Ho) / np.linalg.norm(Ho)),
            "max_abs_W": float(np.max(np.abs(Wd - Wo))), "max_abs_H": float(np.max(np.abs(Hd - Ho))),
            "zero_pattern_mismatches_W": int(np.sum((Wd == 0) != (Wo == 0))),
            "tolerance": 1e-4,
            "against": "oracle.nnqp_solve (exact float64 active set; nmfals.py:70-97) on %d rows / %d columns spread over the "
                       "FULL %dx%d problem, k=%d, first iteration from the seeded W0 / H0: the row QPs with HA = H0 H0^T, the "
                       "column QPs with HA = W1^T W1 and W1^T v_i over all rows of the device's W1 (float64 on the host); "
                       "||X_gpu - X_ref||_F / ||X_ref||_F over the sampled rows / columns" % (ms, ns, m, n, k)}
        out.update(value=1.0 / per_iter,
                   sample="oracle NMFALS (exact active-set QP per row / column, float64, one Python-level solve per "
                          "sub-problem like the reference's cvxopt calls): %d of %d W rows in %.2f s, %d of %d H columns "
                          "in %.2f s, scaled to %d + %d QPs of dimension %d per iteration; %s"
                          % (ms, m, tw, ns, n, th, m, n, k, host))
    elif algo == "NMFALS":                        # no host copy of V at hand (device fill): a sample problem, timing only
        ms = min(m, 256)
        V = gen_rows(np.random.RandomState(1234), m, n, 0, ms)
        np.random.seed(42)
        W = np.random.random((m, k))[:ms]
        H = np.random.random((k, n))
        o = oracle.NMFALSOracle(V, num_bases=k)
        o.W, o.H = W.copy(), H.copy()
        t0 = time.time()
        o.update_w()
        tw = time.time() - t0
        ns = min(n, 256)
        o2 = oracle.NMFALSOracle(V[:, :ns], num_bases=k)
        o2.W, o2.H = o.W.copy(), H[:, :ns].copy()
        t0 = time.time()
        o2.update_h()
        th = time.time() - t0
        out.update(value=1.0 / (tw * m / ms + th * n / ns),
                   sample="oracle NMFALS on a sample problem (%d rows, %d columns), scaled to %d + %d QPs of dimension %d; %s"
                          % (ms, ns, m, n, k, host))
    else:                                         # SNMF on the densified sample (the reference cannot take CSR)
        ms = min(m, 131072)
        ip, ix, vv = gen_csr(m, n, 0.01, 0, ms, fast=True)
        import scipy.sparse as sp
        Vd = np.asarray(sp.csr_matrix((vv, ix, ip), shape=(ms, n)).toarray(), dtype=np.float32)
        np.random.seed(42)
        W = np.random.random((ms, k))
        H = np.random.random((k, n))
        o = oracle.SNMFOracle(Vd, num_bases=k)
        o.W, o.H = W, H

        early = {}

        def step():
            o.update_w()
            o.update_h()
            early["n"] = early.get("n", 0) + 1
            if early["n"] == 2:                   # the state after two iterations, kept beside the final one
                early["W"], early["H"] = np.array(o.W), np.array(o.H)
        it, dt = _time_loop(step, budget_s)
        # what the device is compared with (main(): `parity_full_size`): the same 131 072-row block as a problem of its own,
        # same W0 / H0, same 1 + it iterations -- H is replicated, so its H is what any rank of the sharded run iterates on
        out["_oracle_factors"] = {"iters": 1 + it, "W": np.array(o.W), "H": np.array(o.H), "csr": (ip, ix, vv), "ms": ms,
                                  "W2": early.get("W"), "H2": early.get("H")}
        out.update(value=(it / dt) * ms / float(m),
                   sample="oracle SNMF (snmf.py:67-91) on V.toarray() of the first %d of %d rows (the reference has no "
                          "sparse path), n=%d k=%d: %d iterations in %.2f s, scaled by %d/%d (cost is linear in rows); %s"
                          % (ms, m, n, k, it, dt, ms, m, host))
    return out


_THREAD_ENV = ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "GOTO_NUM_THREADS")


def cpu_baseline_in_child(cfg, m, n, k):
    """cpu_baseline() in a CHILD process that never touches the GPU, with the launchers' BLAS thread caps taken out of its
    environment: torch.distributed.run exports OMP_NUM_THREADS=1 to every rank (bench.py's own launcher: cpu_count / N) and a
    BLAS pool that started with one thread cannot be grown afterwards (threadpoolctl reports the new limit, the GEMM runs at
    the old speed -- measured) -- "the host's cores" would be one core.  Returns (cpu_baseline dict, oracle factors or None)."""
    import shutil
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="pymf_bench_cpu_")
    try:
        env = dict((a, b) for a, b in os.environ.items() if a not in _THREAD_ENV)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", tmp, "--config", cfg,
                        "--rows", str(m), "--cols", str(n), "--bases", str(k)], env=env, check=True, stdout=sys.stderr)
        cb = json.load(open(os.path.join(tmp, "cpu_baseline.json")))
        fac = None
        fp = os.path.join(tmp, "factors.npz")
        if os.path.exists(fp):
            z = np.load(fp)
            fac = {"iters": int(z["meta"][0]), "ferr": float(z["meta"][1]), "W": z["W"], "H": z["H"]}
        return cb, fac
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _cpu_baseline_child(tmp, cfg, m, n, k):
    cb = cpu_baseline(cfg, m, n, k)
    fac = cb.pop("_oracle_factors", None)
    cb.pop("_parity", None)
    if fac is not None and "ferr" in fac:          # the dense NMF configs: what the sharded device run is compared with
        np.savez(os.path.join(tmp, "factors.npz"), W=fac["W"], H=fac["H"], meta=np.array([float(fac["iters"]), fac["ferr"]]))
    cb["process"] = "a child of rank 0 without the launcher's BLAS thread caps (%s)" % ", ".join(_THREAD_ENV[:2])
    with open(os.path.join(tmp, "cpu_baseline.json"), "w") as fh:
        json.dump(cb, fh)


def sharded_parity_nmf(ctx, dist, fac, m, n, k, lo, hi):
    """`parity_full_size` of the ROW-SHARDED run (what the sharding changes is the m-reduction of
    nmf.py:124-125).  Collective.  Rank 0 holds the float64 oracle's factors after `iters` iterations of the FULL problem from
    the seeded W0 / H0 (cpu_baseline's timing loop); every rank restarts its shard from the same W0 rows / H0, runs the same
    number of iterations through the real exchange, and compares ITS row block of W with the oracle's (scattered by rank 0) and
    the replicated H with the oracle's H.  Also: a digest of every rank's device H -- the design's central invariant is that H
    is bit-identical on all ranks (the N partials are added in rank order everywhere)."""
    import hashlib
    w = dist.world()
    head = dist.broadcast_array(np.array([float(fac["iters"]), float(fac["ferr"])]) if w.rank == 0 else None)
    iters, ferr_ref = int(head[0]), float(head[1])
    Href = dist.broadcast_array(fac["H"] if w.rank == 0 else None)
    los, his = dist.allgather_int(lo), dist.allgather_int(hi)
    blob = dist.scatter_bytes([np.ascontiguousarray(fac["W"][a:b]).tobytes() for a, b in zip(los, his)] if w.rank == 0 else None,
                              tag="oracleW")
    Wref = np.frombuffer(blob, dtype=np.float64).reshape(hi - lo, k)
    np.random.seed(42)
    rs = np.random.mtrand._rand
    ctx.set_w(gen_rows(rs, m, k, lo, hi))
    ctx.set_h(np.random.random((k, n)))
    _, done, _ = ctx.factorize(iters, compute_err=False)
    Wd, Hd = ctx.get_w(), ctx.get_h()
    ferr_d = ctx.frobenius()                       # collective: ||V||^2 and the trace terms are summed over the ranks
    d = Wd.astype(np.float64) - Wref
    mine = np.array([float(np.sum(d * d)), float(np.sum(Wref * Wref)), float(np.max(np.abs(d))),
                     float(np.linalg.norm(Hd - Href) / np.linalg.norm(Href)), float(ferr_d), float(done)])
    rows = [np.frombuffer(b, dtype=np.float64) for b in dist.allgather_bytes(mine.tobytes(), tag="parity")]
    digests = dist.allgather_bytes(hashlib.sha256(np.ascontiguousarray(Hd).tobytes()).digest(), tag="Hdigest")
    if w.rank != 0:
        return None
    rel_rank = [float(np.sqrt(r[0] / r[1])) for r in rows]
    return {"iters": int(min(r[5] for r in rows)), "ranks": w.size,
            "relW": float(np.sqrt(sum(r[0] for r in rows) / sum(r[1] for r in rows))),
            "relW_per_rank": rel_rank, "relW_max_over_ranks": max(rel_rank),
            "max_abs_W": max(float(r[2]) for r in rows),
            "relH": float(rows[0][3]), "relH_max_over_ranks": max(float(r[3]) for r in rows),
            "h_identical_across_ranks": all(dg == digests[0] for dg in digests),
            "relferr": float(abs(rows[0][4] - ferr_ref) / ferr_ref), "ferr_gpu": float(rows[0][4]), "ferr_ref": ferr_ref,
            "ferr_identical_across_ranks": all(r[4] == rows[0][4] for r in rows),
            "tolerance": 2e-5, "tolerance_ferr": 1e-5,
            "against": "oracle (NumPy restatement of nmf.py:122-132, float64 W/H, float32 V) on the UNSHARDED %dx%d problem, k=%d, on "
                       "rank 0; the device run is sharded over %d ranks through the real per-iteration exchange, same seeded W0/H0, "
                       "same iteration count; every rank compares its own row block of W (scattered by rank 0); "
                       "||X_gpu - X_ref||_F / ||X_ref||_F" % (m, n, k, w.size)}


def _visible_devices(timeout=180.0):
    """The number of GPUs a rank process will see, counted by a CHILD process (pmf_device_count): the launcher itself makes
    no HIP call (a process that has initialised the GPU must not be the parent that replaces / outlives the ranks)."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); from pymf_amd import _lib; print(_lib.device_count())" % ROOT)
    try:
        p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
        return int(p.stdout.decode().strip().splitlines()[-1]) if p.returncode == 0 else 0
    except (subprocess.TimeoutExpired, ValueError, IndexError):
        return 0


def launch_ranks(args, child=None):
    """`python bench.py --gpus N` with N > 1 and no launcher (WORLD_SIZE unset): start the N rank processes ourselves -- one
    per GPU, RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as `torch.distributed.run`
    would -- relay rank 0's JSON line as our own stdout, and return the job's exit status: non-zero if any rank failed
    (the others are then killed by PID) or the wall-clock limit passed.  Fewer visible GPUs than ranks is an error, never a
    silent 1-GPU line (--debug-share-gpu: every rank on device 0, a plumbing check).  This process never touches the GPU."""
    import socket
    import subprocess
    import threading
    n = int(args.gpus)
    if not args.debug_share_gpu:
        have = _visible_devices()
        if have < n:
            print("bench.py: --gpus %d but %d GPU(s) are visible to pmf_device_count(); refusing to print a line that would "
                  "not be an N-GPU measurement (use --debug-share-gpu for the plumbing check on one GPU)" % (n, have),
                  file=sys.stderr)
            return 2
    # (picked, released, re-bound by rank 0: another process may take the port in between -- pymf_amd.dist then serves on the next
    #  free port of its span and the joiners, who scan the same span and authenticate by HMAC, find it there)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, out0 = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYMF_BENCH_SELF_LAUNCHED="1")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
        procs.append(subprocess.Popen(child or ([sys.executable, os.path.abspath(__file__)] + sys.argv[1:]), env=env,
                                      stdout=subprocess.PIPE, stderr=None))
    stray = []                                # JSON lines printed by ranks other than 0 (there must be none)

    def _drain(r):
        for line in procs[r].stdout:
            text = line.decode("utf-8", "replace")
            if r == 0:
                out0.append(text)
            else:                             # a non-zero rank's stdout goes to our stderr, its JSON lines are counted
                if text.startswith("{"):
                    stray.append(r)
                sys.stderr.write(text)
    drains = [threading.Thread(target=_drain, args=(r,), daemon=True) for r in range(n)]
    for t in drains:
        t.start()
    deadline = time.time() + float(args.launch_timeout)
    status, why = 0, ""
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            status, why = 3, "rank %d exited with status %d" % bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            status, why = 4, "the ranks did not finish within --launch-timeout %.0f s" % args.launch_timeout
            break
        time.sleep(0.05)
    if status:
        for p in procs:                       # exactly the processes started above, by PID
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass
        print("bench.py: %s; the other ranks were stopped" % why, file=sys.stderr)
    for t in drains:
        t.join(timeout=10)
    lines = [l for l in out0 if l.startswith("{")]
    if status == 0 and (len(lines) != 1 or stray):
        print("bench.py: rank 0 printed %d JSON lines, other ranks %d (ranks %s): exactly one line, from rank 0, is the contract"
              % (len(lines), len(stray), sorted(set(stray))), file=sys.stderr)
        status = 5
    if status == 0:
        sys.stdout.write(lines[0])
        sys.stdout.flush()
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg4",
                    help="BASELINE.json workload (default cfg4 = the headline metric)")
    ap.add_argument("--rows", dest="m", type=int, default=0)
    ap.add_argument("--cols", dest="n", type=int, default=0)
    ap.add_argument("--bases", dest="k", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-class-rate", action="store_true",
                    help="skip config.class_factorize_iters_per_sec (pymf_amd.<Class>.factorize on the same inputs)")
    ap.add_argument("--preroll-ms", type=float, default=PREROLL_MS_DEFAULT,
                    help="disclosed device pre-conditioning: untimed iterations of the same loop for about this "
                         "many ms of device time BEFORE the W counted warm-up steps (the chip's clock/power "
                         "state needs more than 5 launches to settle on a fresh box); 0 disables")
    ap.add_argument("--fill", choices=["numpy", "device", "fast"], default="numpy",
                    help="numpy: BASELINE.md protocol; device: counter-based fill of dense inputs (fast start-up); "
                         "fast: cfg5 only, a Poisson/uniform CSR stand-in of the same density (seconds, not a minute)")
    ap.add_argument("--snmf-gram", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="SNMF: -1 library default (Gram-space loop), 0 one pass over V per iteration")
    ap.add_argument("--debug-share-gpu", action="store_true",
                    help="plumbing check on a 1-GPU box: every rank uses device 0, the per-iteration sums cross the ranks "
                         "through the one-shot IPC all-reduce (RCCL refuses two ranks on one GPU); NOT a measurement")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="pmf_set_option(NAME, VALUE) on the bench context (A/B measurements of tuning knobs)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="--gpus N > 1 without a launcher: wall-clock limit (s) for the N rank processes this script starts")
    ap.add_argument("--cpu-baseline-child", metavar="DIR", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_child:                    # (rank 0's helper at N > 1: the oracle on the host's cores, no GPU, no ranks)
        algo_name, m, n, k, _, _ = CONFIGS[args.config]
        _cpu_baseline_child(args.cpu_baseline_child, args.config, args.m or m, args.n or n, args.k or k)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (never touching the GPU in THIS process) and exit with the job's status
        sys.exit(launch_ranks(args))

    from pymf_amd import _lib, dist
    if args.debug_share_gpu:
        # ranks sharing GPU 0 cannot form an RCCL communicator ("Duplicate GPU detected"): the per-iteration sums take the
        # one-shot IPC all-reduce (two processes, one GPU), anything larger the host transport
        os.environ["PYMF_DIST_TRANSPORT"] = "ipc"
    w = dist.init_from_env()
    if w.size != args.gpus and w.rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, w.size),
              file=sys.stderr)
    algo_name, m, n, k, cfg_index, cfg_desc = CONFIGS[args.config]
    m, n, k = args.m or m, args.n or n, args.k or k
    algo = getattr(_lib, "ALGO_" + algo_name)
    lo, hi = w.row_range(m)
    if w.size > 1 and not args.debug_share_gpu and _lib.device_count() < w.size:
        if w.rank == 0:
            print("bench.py: %d ranks but %d visible GPU(s): not an N-GPU measurement (--debug-share-gpu for the plumbing "
                  "check on one GPU)" % (w.size, _lib.device_count()), file=sys.stderr)
        sys.exit(2)
    # RCCL communicator (time-boxed), with the one-shot IPC all-reduce in front of it where it passes its self-test
    ctx = dist.make_context(algo, hi - lo, n, k, share_gpu=args.debug_share_gpu)
    for ov in args.option:
        ctx.set_option(ov.split("=")[0], int(ov.split("=")[1]))
    data = "synthetic"
    nnz_local = 0
    host = {}                    # host copies for the class-level measurement (1 rank, numpy fill)
    want_class = (w.size == 1 and not args.no_class_rate)
    if args.config == "cfg5":
        ip, ix, vv = gen_csr(m, n, 0.01, lo, hi, fast=(args.fill != "numpy"))
        nnz_local = int(vv.shape[0])
        ctx.set_v_csr(ip, ix, vv)
        if want_class:
            import scipy.sparse as sp
            host["V"] = sp.csr_matrix((vv, ix, ip), shape=(hi - lo, n))
        del ip, ix, vv
        if args.fill != "numpy":
            data = "synthetic (Poisson/uniform CSR stand-in of the BASELINE density)"
        # headline loop: W written in every iteration (snmf_gram = 2); --snmf-gram overrides
        ctx.set_option("snmf_gram", args.snmf_gram if args.snmf_gram >= 0 else 2)
    elif args.fill == "device":
        ctx.fill_v_uniform(1234, lo)
    else:
        V = gen_rows(np.random.RandomState(1234), m, n, lo, hi)
        ctx.set_v_dense(V)
        if want_class:
            host["V"] = V
        del V
    if args.fill == "device" and args.config != "cfg5":
        ctx.fill_w_uniform(42, lo)
        ctx.fill_h_uniform(43)
    else:
        np.random.seed(42)
        rs = np.random.mtrand._rand            # the global legacy stream, as np.random.random uses
        W0 = gen_rows(rs, m, k, lo, hi)
        H0 = np.random.random((k, n))
        ctx.set_w(W0)
        ctx.set_h(H0)
        if want_class and "V" in host:
            host["W"], host["H"] = W0, H0
        if algo_name == "NMFALS":
            host["W0"], host["H0"] = W0, H0      # for the from-the-random-start rate (below)
        del W0, H0

    # ---- the COLD figure: the very first factorize(K) of this context on resident data -- no pre-roll, no
    # warm-up, the chip at whatever clock it idles at, one-time work of a first call included; printed as a top-level sibling
    # of the pre-rolled `value` so that both clock states are one glance apart ----
    ctx.synchronize()
    dist.barrier()
    tc = time.perf_counter()
    _, done_c, _ = ctx.factorize(args.steps, compute_err=False)
    ctx.synchronize()
    cold_dt = dist.allreduce_max(time.perf_counter() - tc)
    cold_rate = done_c / cold_dt

    # ---- disclosed pre-conditioning (untimed, reported as "preroll_ms" / "preroll_iters") ----
    preroll_iters, preroll_ms = 0, 0.0
    state_dependent = algo_name == "NMFALS"      # the cost of an ALS iteration depends on how settled the active sets are
    if args.preroll_ms > 0:
        ctx.factorize(2, compute_err=False)          # (one-time work of a first call -- V^T V of the Gram-space loop, buffers -- ...)
        ctx.factorize(2, compute_err=False)          # ... stays out of the per-iteration estimate
        # every iteration carries a collective: ALL ranks must run the same count (the slowest rank's estimate)
        per = dist.allreduce_max(max(ctx.last_loop_ms() / 2.0, 1e-3))
        preroll_iters = int(min(max(args.preroll_ms / per, 1), 20000))
        ctx.factorize(preroll_iters, compute_err=False)
        preroll_ms = ctx.last_loop_ms()
        preroll_iters += 4
        if state_dependent and "W0" in host:
            # BASELINE.md section 3 / SURVEY 8(d): warm-up and timed iterations start from the SEEDED W0 / H0.  The pre-roll
            # only brings the chip to its clock; the factors go back to the seeded start, so `value` includes the expensive
            # first iterations (every row QP still changing its active set) -- the settled rate is reported beside it.
            ctx.set_w(host["W0"])
            ctx.set_h(host["H0"])

    elif state_dependent and "W0" in host:       # (no pre-roll: the cold call above still moved the factors)
        ctx.set_w(host["W0"])
        ctx.set_h(host["H0"])

    # ---- warm-up (untimed) ----
    if args.warmup > 0:
        _, done, _ = ctx.factorize(args.warmup, compute_err=False)
        assert done == args.warmup

    # ---- timed region: exactly K steps, barrier + device sync on both sides ----
    # Live HIP events around the dominant kernel: on this stack a timed launch costs the loop ~5 us (two more queue packets + the
    # dispatch's completion signal; tools/loop_probe.py: 59.3 -> 64.4 us per iteration at cfg2 with every launch timed), so only
    # every `timed_every`-th launch of the timed region carries a pair -- about one per millisecond, at least 8 per region
    timed_every = 1
    if algo_name != "NMFALS" and preroll_iters > 4:          # (NMFALS: the counting pass pairs launches with counters one to one)
        est_ms = max(preroll_ms / max(preroll_iters - 4, 1), 1e-3)
        timed_every = max(1, min(args.steps // 8, int(np.ceil(1.0 / est_ms))))
    timed_every = int(dist.allreduce_max(timed_every))
    ctx.set_option("profile_every", timed_every)
    ctx.profile_enable(True)
    ctx.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    _, done, conv = ctx.factorize(args.steps, compute_err=False)
    ctx.synchronize()
    dt = time.perf_counter() - t0         # this rank's K steps are done; the closing barrier (a TCP round
    dist.barrier()                        # trip through rank 0) is not part of anybody's K steps
    assert done == args.steps and conv < 0, "timed run was shortened (%d of %d)" % (done, args.steps)
    rank_dt = dist.allgather_float(dt)    # every rank's own K steps; `value` uses the slowest
    dt = max(rank_dt)
    stats = ctx.kernel_stats()
    coll_ms, coll_n = ctx.collective_ms() if w.size > 1 else (0.0, 0)
    all_ms = ctx.kernel_launch_ms()
    launch_ms = np.sort(all_ms)
    first_ms = [round(float(x), 4) for x in all_ms[:5]]
    ctx.profile_enable(False)
    per_rank_kernel_ms = dist.allgather_float(stats["mean_ms"])        # every rank's shard kernel: a slow GPU shows
    # ---- N > 1: the OTHER transport in the same run.  `value` above rode on the one-shot exchange when it
    # passed its self-test; the same K steps once more with the per-iteration sum carried by the transport behind it --
    # ncclAllReduce (RCCL) on a multi-GPU node, the host round trip where the ranks share a GPU (RCCL refuses that set-up) ----
    other_transport = None
    if w.size > 1 and algo_name == "NMF":
        behind = ("host transport (TCP star through rank 0; the ranks share GPU 0 and RCCL refuses two ranks on one GPU) -- "
                  "NOT RCCL, a plumbing figure" if dist.transport() in ("host", "ipc") else "ncclAllReduce (RCCL)")
        if dist.LAST_SETUP.get("oneshot") == "passed":
            ctx.set_option("oneshot_allreduce", 0)                     # every rank, at the same point of its call sequence
            ctx.factorize(max(args.warmup, 2), compute_err=False)
            ctx.profile_enable(True)
            ctx.synchronize()
            dist.barrier()
            t6 = time.perf_counter()
            _, done_o, _ = ctx.factorize(args.steps, compute_err=False)
            ctx.synchronize()
            d6 = time.perf_counter() - t6
            dist.barrier()
            d6 = dist.allreduce_max(d6)
            oc_ms, oc_n = ctx.collective_ms()
            ctx.profile_enable(False)
            ctx.set_option("oneshot_allreduce", 1)
            ctx.factorize(2, compute_err=False)                        # (back on the main path, and both slots exercised again)
            other_transport = {"transport": behind, "iters_per_sec": done_o / d6, "ms_per_step": d6 / max(done_o, 1) * 1e3,
                               "steps": done_o, "collective_mean_ms": oc_ms if oc_n else None, "collective_launches": oc_n}
        else:
            other_transport = {"transport": behind, "skipped": "the one-shot exchange is not in front of it (%s): `value` itself "
                               "was carried by this transport" % dist.LAST_SETUP.get("oneshot")}
    # NMFALS: the rate once the active sets have settled (about 300 iterations from the seeded start), same K steps
    settled, qp_counts = None, None
    if state_dependent:
        more = max(0, 300 - (args.warmup + args.steps))
        if more:
            ctx.factorize(more, compute_err=False)
        ctx.synchronize()
        dist.barrier()
        t5 = time.perf_counter()
        _, done_q, _ = ctx.factorize(args.steps, compute_err=False)
        ctx.synchronize()
        d5 = time.perf_counter() - t5
        dist.barrier()
        d5 = dist.allreduce_max(d5)
        settled = {"iters_per_sec": done_q / d5, "ms_per_step": d5 / max(done_q, 1) * 1e3, "steps": done_q,
                   "after_iterations": args.warmup + args.steps + more}
        # COUNTING PASS (same run, outside every timed region): the deterministic loop once more from the seeded start on the
        # counting instantiation of the QP kernel (it costs the kernel 8 %, so the timed loops carry no counters) -- the
        # counts of exactly the W half steps of the warm-up + timed region, and of the settled K steps
        if "W0" in host and "quad" in stats["name"]:
            ctx.set_w(host["W0"])
            ctx.set_h(host["H0"])
            ctx.set_option("nnqp_count", 1)
            ctx.nnqp_counters(reset=True)
            ctx.factorize(args.warmup + args.steps, compute_err=False)
            qp_counts = ctx.nnqp_counters(reset=True)
            if more:
                ctx.factorize(more, compute_err=False)
            ctx.nnqp_counters(reset=True)
            ctx.factorize(args.steps, compute_err=False)
            settled["qp_counts"] = ctx.nnqp_counters(reset=True)
            ctx.set_option("nnqp_count", 0)

    # secondary: the API-default compute_err=True rate (not the headline value; the reference has no
    # error on sparse data, nmf.py:109-112)
    rate_err = None
    if args.config != "cfg5":
        ne = max(2, min(args.steps, 20))
        ctx.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        _, done_e, conv_e = ctx.factorize(ne, compute_err=True)
        ctx.synchronize()
        dt_e = time.perf_counter() - t1
        dist.barrier()
        rate_err = done_e / dist.allreduce_max(dt_e)

    # NMFALS: `value` is the rate once the pre-roll has run (the active sets of most rows have settled: one solve per
    # row QP).  Beside it: the first iterations from the random start, where every row QP still changes its set.
    rate_from_start = None
    if algo_name == "NMFALS" and "W0" in host:
        ctx.set_w(host["W0"])
        ctx.set_h(host["H0"])
        ns0 = max(2, min(args.steps, 10))
        ctx.synchronize()
        dist.barrier()
        t4 = time.perf_counter()
        _, done_s, _ = ctx.factorize(ns0, compute_err=False)
        ctx.synchronize()
        d4 = time.perf_counter() - t4
        dist.barrier()
        rate_from_start = {"iters_per_sec": done_s / dist.allreduce_max(d4), "steps": ns0,
                           "note": "iterations 1 .. %d from the BASELINE random W0 / H0, no pre-roll" % ns0}

    # cfg5: `value` is the loop that writes W = V M in EVERY iteration (what the reference's update_w does,
    # snmf.py:67-70).  Beside it: the Gram-space loop that materialises W once per factorize() -- as an
    # amortised rate for this K, taken apart into its fixed cost and its per-iteration cost, and COLD (fresh
    # V: the one-time C = V^T V inside the measured call) -- and round 1's one-pass CSR kernel.
    gram_loop, rate_pass_per_iter = None, None
    if args.config == "cfg5" and args.snmf_gram == -1:
        def timed(steps):
            ctx.synchronize()
            dist.barrier()
            t2 = time.perf_counter()
            ctx.factorize(steps, compute_err=False)
            ctx.synchronize()
            d = time.perf_counter() - t2
            dist.barrier()
            return dist.allreduce_max(d)
        ctx.set_option("snmf_gram", 1)
        ctx.factorize(2, compute_err=False)
        t_k, t_2k = timed(args.steps), timed(2 * args.steps)
        per_it = max(t_2k - t_k, 0.0) / args.steps
        ctx.invalidate_v()                       # forget C = V^T V (and everything else derived from V)
        t_cold = timed(args.steps)
        gram_loop = {"iters_per_sec_amortised_over_steps": args.steps / t_k, "steps": args.steps,
                     "per_iteration_ms": per_it * 1e3, "fixed_ms_per_factorize": max(t_k - args.steps * per_it, 0.0) * 1e3,
                     "cold_factorize_ms": t_cold * 1e3,
                     "cold_iters_per_sec": args.steps / t_cold,
                     "note": "W = V M written once per factorize(); cold = first call on fresh data, C = V^T V formed inside"}
        ctx.set_option("snmf_gram", 0)
        ctx.factorize(2, compute_err=False)
        rate_pass_per_iter = args.steps / timed(args.steps)
        ctx.set_option("snmf_gram", 2)

    # class-level rate: pymf_amd.<Class>(data).factorize(K) on the same inputs, SECOND call on an object whose
    # data and factors are resident (the first call uploads); default settings, and with check_data off
    class_rate = None
    if want_class and w.rank == 0 and "W" in host:
        import pymf_amd
        per_step_s = dt / args.steps
        kc = args.steps if per_step_s > 2e-3 else max(args.steps, 100)
        cls = {"NMF": pymf_amd.NMF, "NMFALS": pymf_amd.NMFALS, "SNMF": pymf_amd.SNMF}[algo_name]
        mdl = cls(host["V"], num_bases=k)
        mdl.W, mdl.H = host.pop("W"), host.pop("H")
        class_rate = {"steps": kc}
        if algo_name == "SNMF":
            class_rate["note"] = ("the class runs the library's default SNMF loop: Gram-space iterations, W = V M written ONCE per "
                                  "factorize() -- compare with config.gram_space_loop_w_once_per_factorize, not with `value`")
        try:
            # FIRST call of a fresh object with pre-set factors: context creation, digests, upload of data / W / H, loop
            kf = 50
            t3 = time.perf_counter()
            mdl.factorize(niter=kf, compute_err=False)
            first_call = dict((k_, round(float(v_), 3)) for k_, v_ in mdl.last_call_ms.items())
            first_call.update(steps=kf, wall_ms=round((time.perf_counter() - t3) * 1e3, 3),
                              note="fresh pymf_amd.%s(V, k) with W, H assigned, factorize(%d, compute_err=False): ctx = context "
                                   "creation, upload = digests + host->device copies of data, W, H, loop = device loop" % (algo_name, kf))
            class_rate["first_call"] = first_call
            for key, chk in (("iters_per_sec", True), ("iters_per_sec_check_data_off", False)):
                mdl.check_data = chk
                d3, ov = None, None
                for _rep in range(3):            # best of three calls (a 256-thread host digests 1 GiB in 5 ms -- or 70)
                    t3 = time.perf_counter()
                    mdl.factorize(niter=kc, compute_err=False)
                    d = time.perf_counter() - t3
                    if d3 is None or d < d3:
                        d3, ov = d, (d - mdl._ctx.last_loop_ms() * 1e-3) * 1e3
                class_rate[key] = kc / d3
                class_rate["call_overhead_ms" + ("" if chk else "_check_data_off")] = ov
            _ = mdl.W                            # the read that refreshes the host array (not inside the rate)
        finally:
            if mdl._ctx is not None:
                mdl._ctx.close()
        del mdl
    als_sample = None
    if (algo_name == "NMFALS" and w.size == 1 and not args.no_cpu_baseline and "W0" in host and "V" in host
            and not hasattr(host["V"], "tocsr")):
        # one iteration at FULL size from the seeded start, kept for the oracle (cpu_baseline times exactly these QPs)
        ctx.set_w(host["W0"]); ctx.set_h(host["H0"])
        ctx.update_w()
        Wd = ctx.get_w()
        ctx.update_h()
        Hd = ctx.get_h()
        rows = np.arange(0, m, max(m // 256, 1))[:256]
        cols = np.arange(0, n, max(n // 256, 1))[:256]
        HA_h = np.zeros((k, k)); FA = np.zeros((k, len(cols)))
        for r0 in range(0, m, 32768):                      # float64 on the host, in row chunks
            Wc = Wd[r0:r0 + 32768].astype(np.float64)
            HA_h += Wc.T.dot(Wc)
            FA += Wc.T.dot(host["V"][r0:r0 + 32768][:, cols].astype(np.float64))
        als_sample = {"rows": rows, "cols": cols, "Vrows": host["V"][rows].copy(), "H0": np.array(host["H0"]),
                      "Wd_rows": Wd[rows].copy(), "Hd_cols": Hd[:, cols].copy(), "HA_h": HA_h, "FA_cols": FA}
        del Wd, Hd
    host.clear()

    # ---- N > 1: the host-CPU number and the parity of the SHARDED run in the same line.  Rank 0 times the
    # oracle on the host's cores AFTER every timed region (the other ranks wait in the barrier behind it: the rendezvous sockets
    # block without a time limit, and nothing is being measured any more); the sharded parity run follows on all ranks ----
    cb_multi, par_multi = None, None
    if w.size > 1 and not args.no_cpu_baseline:
        fac_m = None
        if w.rank == 0:
            cb_multi, fac_m = cpu_baseline_in_child(args.config, m, n, k)
            cb_multi["note"] = "rank 0 of %d, after the timed regions; the other ranks idle in a barrier meanwhile" % w.size
        dist.barrier()
        if algo_name == "NMF" and args.fill == "numpy":
            par_multi = sharded_parity_nmf(ctx, dist, fac_m, m, n, k, lo, hi)
        elif w.rank == 0:
            par_multi = {"skipped": "the sharded full-size comparison is built for the dense NMF configs with the BASELINE (numpy) fill; "
                                    "%s / --fill %s at N > 1 is covered by tests/test_dist_ranks.py on small problems" % (args.config, args.fill)}
        del fac_m

    if w.rank == 0:
        mean_s = stats["mean_ms"] * 1e-3
        fl, ex, by = stats["flops_per_launch"], stats["executed_flops_per_launch"], stats["bytes_per_launch"]
        mfma_bound = by > 0 and fl / by > RIDGE_FLOP_PER_BYTE
        if mean_s > 0 and mfma_bound:
            ach, peak, unit, bound = fl / mean_s / 1e12, PEAK_F32_MFMA_TFLOPS, "TFLOP/s", "mfma"
        elif mean_s > 0:
            ach, peak, unit, bound = by / mean_s / 1e9, PEAK_HBM_GBS, "GB/s", "hbm"
        else:
            ach, peak, unit, bound = 0.0, PEAK_HBM_GBS, "GB/s", "hbm"
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("%s@%dx%dx%d/%d" % (stats["name"], m, n, k, w.size))
            except Exception:
                traffic = None
        if w.size == 1:
            collective = "none"
        elif algo_name == "SNMF":
            collective = "sum of V^T V once per factorize(); transports: " + ctx.collective_name
        else:
            collective = "sum of (W^T V | W^T W), %d B/iter; transports: %s" % (4 * k * (n + k), ctx.collective_name)
        out = {
            "metric": "factorize_iters_per_sec",
            "value": args.steps / dt,
            "unit": "iter/s",
            "n_gpus": w.size,
            "steps": args.steps,
            "warmup": args.warmup,
            "preroll_ms": preroll_ms,
            "preroll_iters": preroll_iters,
            "ms_per_step": dt / args.steps * 1e3,
            "cold_iters_per_sec": cold_rate,       # the first factorize(K) on resident data: no pre-roll, no warm-up (see preroll_ms)
            "rank_ms_per_step": [x / args.steps * 1e3 for x in rank_dt],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 MFMA contractions + f64 QP solves" if algo_name == "NMFALS" else "f32",
            "data": data,
            "config": {"workload": "%s (BASELINE.json configs[%d]): %s %dx%d, k=%d, rows sharded over %d GPU(s), "
                                   "compute_err=False" % (args.config, cfg_index, cfg_desc, m, n, k, w.size),
                       "m": m, "n": n, "k": k, "algo": algo_name, "path": ctx.path_name,
                       "collective": collective,
                       "collective_setup": dict(dist.LAST_SETUP) if w.size > 1 else None,   # transport, ranks, one-shot self-test verdict
                       "ipc_selftest": dist.LAST_SETUP.get("oneshot") if w.size > 1 else None,
                       "launched_by": ("bench.py itself (N rank processes, no launcher)" if os.environ.get("PYMF_BENCH_SELF_LAUNCHED")
                                       else "external launcher (WORLD_SIZE in the environment)") if w.size > 1 else None,
                       "collective_mean_ms": coll_ms if coll_n else None,      # HIP events around the per-iteration sum, this rank
                       "collective_launches": coll_n,
                       "other_transport_same_run": other_transport,
                       "rccl_only_iters_per_sec": (other_transport or {}).get("iters_per_sec") if w.size > 1 and dist.transport() == "rccl" else None,
                       "rccl_collective_mean_ms": (other_transport or {}).get("collective_mean_ms") if w.size > 1 and dist.transport() == "rccl" else None,
                       "compute_err_true_iters_per_sec": rate_err,
                       "class_factorize": class_rate,
                       "first_call": (class_rate or {}).get("first_call"),
                       "settled": settled,
                       "settled_iters_per_sec": settled["iters_per_sec"] if settled else None,
                       "from_random_start": rate_from_start},
            "roofline": {"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                         "traffic": traffic,
                         "traffic_source": ("profiles/traffic.json: rocprofv3 PMC passes of this kernel on this shape "
                                            "(2 x FETCH_SIZE + WRITE_SIZE per launch), collected by tools/pmc_configs.sh -- "
                                            "a recorded constant, not counted in this run") if traffic is not None
                                           else "none recorded for this kernel / shape / rank count",
                         "kernel": stats["name"], "launches": stats["launches"],
                         "launches_in_timed_region": args.steps if algo_name != "NMFALS" else stats["launches"],
                         "timed_every": timed_every,
                         "mean_kernel_ms": stats["mean_ms"],
                         "per_rank_kernel_ms": per_rank_kernel_ms,      # rank order; `achieved` is rank 0's shard kernel
                         "min_kernel_ms": float(launch_ms[0]) if len(launch_ms) else None,
                         "median_kernel_ms": float(np.median(launch_ms)) if len(launch_ms) else None,
                         "max_kernel_ms": float(launch_ms[-1]) if len(launch_ms) else None,
                         "first_launches_ms": first_ms,
                         "flops_per_launch": fl,
                         "executed_flops_per_launch": ex,
                         "executed_TFLOPs": ex / mean_s / 1e12 if mean_s > 0 else 0.0,
                         "algorithmic_bytes_per_launch": by,
                         "achieved_hbm_GBs": by / mean_s / 1e9 if mean_s > 0 else 0.0,
                         "kernel_share_of_timed_region": (args.steps if algo_name != "NMFALS" else stats["launches"]) * mean_s / dt if dt > 0 else 0.0},
        }
        if algo_name == "NMFALS":
            # The QP kernel runs float64 VALU code: neither HBM nor MFMA bounds it.  Against the 78.6 TFLOP/s float64 vector
            # peak: the FMA flop k_nnqp_quad EXECUTES, from the counts of this run's counting pass (the warm-up + timed
            # iterations repeated from the seeded start; pmf_nnqp_counters: wave tasks of 4 problems, passes, sum over the passes of the largest system
            # among a wave's four problems = the size ns its frame-padded elimination runs over).  Per problem: y0 = B f once
            # (2 * 64 * 64), then per pass the correction over the small set and the product with the solution (2 * 64 * ns
            # each), the LDL^T of the ns x ns block (2 ns^3 / 3) and the two triangular solves (2 * 2 ns^2).
            qps = float(hi - lo)
            quad = "quad" in stats["name"]
            if quad and qp_counts is not None:
                nl = max(stats["launches"] + args.warmup, 1)        # W half steps the counters cover
                fma_flop, tasks, passes, nssum = 0.0, 0, 0, 0
                for fr in ("frame16", "frame32"):
                    cnt = qp_counts[fr]
                    if cnt["passes"] == 0:
                        continue
                    ns = cnt["sum_largest_system"] / float(cnt["passes"])
                    fma_flop += 4.0 * (cnt["wave_tasks"] * 2.0 * 64.0 * 64.0 +
                                       cnt["passes"] * (4.0 * 64.0 * ns + 2.0 * ns ** 3 / 3.0 + 4.0 * ns * ns))
                    tasks += cnt["wave_tasks"]; passes += cnt["passes"]; nssum += cnt["sum_largest_system"]
                fma_flop /= nl
                src = ("counting pass of this run: the same %d W half steps (warm-up + timed) repeated from the seeded start on the "
                       "counting instantiation of the kernel (pmf_nnqp_counters)" % nl)
                out["roofline"]["qp_counts"] = dict(qp_counts, w_half_steps=nl,
                                                    passes_per_wave_task=passes / float(max(tasks, 1)),
                                                    mean_largest_system=nssum / float(max(passes, 1)))
            else:                                 # lane-per-variable kernel: DESIGN.md 3.4, profiles/r02_pmc_summary.csv
                fma_flop = 2.0 * 2.24e5 * qps * (k / 64.0) ** 3
                src = "modelled: profiles/r02_pmc_summary.csv (k_nnqp, 2.24e5 FMA per QP at k = 64)"
            if mean_s > 0:
                out["roofline"].update(bound="valu_f64", achieved=fma_flop / mean_s / 1e12, peak=78.6, unit="TFLOP/s",
                                       frac=fma_flop / mean_s / 1e12 / 78.6, flops_per_launch=fma_flop,
                                       executed_flops_per_launch=fma_flop, executed_TFLOPs=fma_flop / mean_s / 1e12)
                out["roofline"]["flop_source"] = src
                if quad and (m, n, k, w.size) == (262144, 1024, 64, 1):
                    # VALU issue slots: a RECORDED instruction count (rocprofv3 SQ_INSTS_VALU of a settled W-step launch on
                    # exactly this shape: 1.16e8 = 442 per problem) -- a model for any other state, labelled as such
                    valu_instr = 1.16e8
                    out["roofline"]["valu_issue_modelled"] = {"wave_instructions_per_launch_recorded": valu_instr,
                                                              "achieved_Ginstr_per_s": valu_instr / mean_s / 1e9,
                                                              "peak_Ginstr_per_s": 1024 * 2.4 / 4.0,
                                                              "frac": valu_instr / mean_s / 1e9 / (1024 * 2.4 / 4.0),
                                                              "source": "profiles/r03_pmc_cfg3_nnqp_quad.txt (settled state); not counted in this run"}
            out["roofline"]["note"] = ("float64 VALU kernel: `achieved` = executed FMA flop of the row QPs (live device counts) / live launch "
                                       "time of the 16-slot-frame launch plus the 32-slot one behind it; the kernel is bound by the dependent "
                                       "chains of its factorisations and triangular solves at two / three waves per SIMD, not by flops")
        if args.config == "cfg5":
            out["config"]["nnz_local"] = nnz_local
            out["config"]["gram_space_loop_w_once_per_factorize"] = gram_loop
            out["config"]["one_pass_csr_kernel_iters_per_sec"] = rate_pass_per_iter
            out["config"]["loop"] = ("k x n sized Gram-space iteration (P = M^T (V^T V), S = P M) + W = V M written in EVERY "
                                     "iteration, as the reference's update_w does; the write of iteration i runs on a stream of its "
                                     "own beside the k x n sized kernels of iteration i + 1 (option snmf_w_pipe; 0 = stream order)"
                                     if stats["name"].startswith("k_csr_w") else "one pass over the CSR rows per iteration")
        if cb_multi is not None:
            out["cpu_baseline"] = cb_multi
            out["parity_full_size"] = par_multi
        if w.size == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.config, m, n, k, sample=als_sample)
            fac = cb.pop("_oracle_factors", None)
            par = cb.pop("_parity", None)
            out["cpu_baseline"] = cb
            if par is not None:
                out["parity_full_size"] = par
            if fac is not None and "csr" in fac:
                # cfg5: the oracle's row block fed to the device as its own CSR problem, same seeded
                # W0 / H0, same iteration count, the headline loop (W written every iteration)
                ip, ix, vv = fac["csr"]
                c2 = _lib.Context(algo, fac["ms"], n, k, device=0 if args.debug_share_gpu else w.local_rank)
                try:
                    c2.set_v_csr(ip, ix, vv)
                    c2.set_option("snmf_gram", args.snmf_gram if args.snmf_gram >= 0 else 2)
                    np.random.seed(42)
                    c2.set_w(np.random.random((fac["ms"], k)))
                    c2.set_h(np.random.random((k, n)))
                    after2 = None
                    if fac.get("W2") is not None and fac["iters"] > 2:
                        c2.factorize(2, compute_err=False)
                        W2, H2 = c2.get_w(), c2.get_h()
                        after2 = {"relW": float(np.linalg.norm(W2 - fac["W2"]) / np.linalg.norm(fac["W2"])),
                                  "relH": float(np.linalg.norm(H2 - fac["H2"]) / np.linalg.norm(fac["H2"]))}
                        del W2, H2
                        _, done_p, _ = c2.factorize(fac["iters"] - 2, compute_err=False)
                        done_p += 2
                    else:
                        _, done_p, _ = c2.factorize(fac["iters"], compute_err=False)
                    Wd, Hd = c2.get_w(), c2.get_h()
                finally:
                    c2.close()
                rows = np.arange(0, fac["ms"], 257)
                out["parity_full_size"] = {
                    "iters": int(done_p),
                    "relW": float(np.linalg.norm(Wd - fac["W"]) / np.linalg.norm(fac["W"])),
                    "relW_rows_sampled": float(np.linalg.norm(Wd[rows] - fac["W"][rows]) / np.linalg.norm(fac["W"][rows])),
                    "relH": float(np.linalg.norm(Hd - fac["H"]) / np.linalg.norm(fac["H"])),
                    "H_min": float(Hd.min()), "after_2_iterations": after2,
                    "tolerance": {"H": 2e-5, "W": 5e-5},
                    "tolerance_note": "k = n = 128: H H^T of a square H has cond ~ 1e7 and W = V inv(H H^T) H amplifies any error of H by "
                                      "sigma_max / sigma_min ~ 3e3 (DESIGN.md 4.1).  Rounds 1-5 stored H in float32 between the iterations "
                                      "(W: 2e-4 after 50 iterations, tolerance 1e-3); since round 6 the device keeps SNMF's H in float64 "
                                      "(pmf_inv.h: k_snmf_h_f64; the reference's H is float64, nmf.py:120) and only M = H^T inv(H H^T) is "
                                      "rounded for the float32 product W = V M",
                    "against": "oracle SNMF (snmf.py:67-91, float64 W/H) on V.toarray() of a %d-row block of the cfg5 matrix (n=%d, "
                               "k=%d) fed to the device as its own CSR problem: same seeded W0/H0, same %d iterations; H is "
                               "replicated in the row-sharded run, so this is the H every rank iterates on for that block; "
                               "||X_gpu - X_ref||_F / ||X_ref||_F (W crosses zero: Frobenius-relative only)" % (fac["ms"], n, k, fac["iters"])}
                del Wd, Hd, fac
            elif fac is not None and args.fill == "numpy":
                # FULL-SIZE parity: the device from the same W0 / H0 for the same number of iterations as the
                # float64-default oracle just ran for its timing -- every row of W, all of H
                np.random.seed(42)
                rs = np.random.mtrand._rand
                ctx.set_w(gen_rows(rs, m, k, 0, m))
                ctx.set_h(np.random.random((k, n)))
                _, done_p, _ = ctx.factorize(fac["iters"], compute_err=False)
                Wd, Hd = ctx.get_w(), ctx.get_h()
                ferr_d = ctx.frobenius()                   # frobenius_norm() of the same state (nmf.py:100-114)
                out["parity_full_size"] = {
                    "iters": int(done_p),
                    "relferr": float(abs(ferr_d - fac["ferr"]) / fac["ferr"]), "ferr_gpu": float(ferr_d), "ferr_ref": fac["ferr"],
                    "tolerance_ferr": 1e-5,
                    "relW": float(np.linalg.norm(Wd - fac["W"]) / np.linalg.norm(fac["W"])),
                    "relH": float(np.linalg.norm(Hd - fac["H"]) / np.linalg.norm(fac["H"])),
                    "max_abs_W": float(np.max(np.abs(Wd - fac["W"]))),
                    "tolerance": 2e-5,
                    "against": "oracle (NumPy restatement of nmf.py:122-132, float64 W/H, float32 V) at %dx%d, k=%d, same "
                               "seeded W0/H0, same iteration count; ||X_gpu - X_ref||_F / ||X_ref||_F" % (m, n, k)}
                del Wd, Hd, fac
        print(json.dumps(out), flush=True)
    ctx.close()
    dist.shutdown()


if __name__ == "__main__":
    main()
