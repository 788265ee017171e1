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


def cpu_baseline(cfg, m, n, k, budget_s=24.0):
    """The oracle (NumPy restatement of the reference, its op order) on this host's cores.
    cfg4 / cfg2: FULL size, reference-default operands (float64 W/H, float32 V) = `value`, the all-float32
    variant alongside.  cfg3 / cfg5: a bounded row sample, scaled -- stated in `sample`."""
    import oracle
    algo = CONFIGS[cfg][0]
    out = {"unit": "iter/s", "cores": _blas_threads(), "kind": "port"}
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
        it32, dt32 = _time_loop(step32, budget_s * 0.3)
        out.update(value=it / dt, value_all_fp32=it32 / dt32,
                   sample="oracle NMF (reference op order, float64 W/H, float32 V) at FULL size %dx%d k=%d: "
                          "1 warm-up + %d iterations in %.2f s; all-float32 variant: %d iterations in %.2f s; %s"
                          % (m, n, k, it, dt, it32, dt32, host))
    elif algo == "NMFALS":
        ms = min(m, 256)                          # rows of the W half step in the sample
        V = gen_rows(np.random.RandomState(1234), m, n, 0, ms)
        np.random.seed(42)
        W = np.random.random((m, k))[:ms]
        H = np.random.random((k, n))
        o = oracle.NMFALSOracle(V, num_bases=k)
        o.W, o.H = W.copy(), H.copy()
        t0 = time.time()
        o.update_w()
        tw = time.time() - t0                     # ms QPs of the W half step
        ns = min(n, 256)
        o2 = oracle.NMFALSOracle(V[:, :ns], num_bases=k)
        o2.W, o2.H = o.W.copy(), H[:, :ns].copy()
        t0 = time.time()
        o2.update_h()
        th = time.time() - t0                     # ns QPs of the H half step (Hessian of the sample's W)
        per_iter = tw * m / ms + th * n / ns
        out.update(value=1.0 / per_iter,
                   sample="oracle NMFALS (exact active-set QP per row / column, float64, one Python-level solve per "
                          "sub-problem like the reference's cvxopt calls): %d of %d W rows in %.2f s, %d of %d H columns "
                          "in %.2f s, scaled to %d + %d QPs of dimension %d per iteration; %s"
                          % (ms, m, tw, ns, n, th, m, n, k, host))
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

        def step():
            o.update_w()
            o.update_h()
        it, dt = _time_loop(step, budget_s)
        out.update(value=(it / dt) * ms / float(m),
                   sample="oracle SNMF (snmf.py:67-91) on V.toarray() of the first %d of %d rows (the reference has no "
                          "sparse path), n=%d k=%d: %d iterations in %.2f s, scaled by %d/%d (cost is linear in rows); %s"
                          % (ms, m, n, k, it, dt, ms, m, host))
    return out


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
                    help="plumbing check on a 1-GPU box: every rank uses device 0 and its own 1-rank RCCL "
                         "communicator (rows still sharded, barriers still used); NOT a measurement")
    args = ap.parse_args()

    from pymf_amd import _lib, dist
    w = dist.init_from_env()
    if w.size != args.gpus and w.rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, w.size),
              file=sys.stderr)
    algo_name, m, n, k, cfg_index, cfg_desc = CONFIGS[args.config]
    m, n, k = args.m or m, args.n or n, args.k or k
    algo = getattr(_lib, "ALGO_" + algo_name)
    lo, hi = w.row_range(m)
    if args.debug_share_gpu:
        ctx = _lib.Context(algo, hi - lo, n, k, device=0, rank=0, nranks=1, nccl_id=_lib.nccl_unique_id())
    else:
        ctx = _lib.Context(algo, hi - lo, n, k, device=w.local_rank, rank=w.rank,
                           nranks=w.size, nccl_id=w.nccl_id)
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

    # ---- disclosed pre-conditioning (untimed, reported as "preroll_ms" / "preroll_iters") ----
    preroll_iters, preroll_ms = 0, 0.0
    if args.preroll_ms > 0:
        ctx.factorize(2, compute_err=False)
        # every iteration carries a collective: ALL ranks must run the same count (the slowest rank's estimate)
        per = dist.allreduce_max(max(ctx.last_loop_ms() / 2.0, 1e-3))
        preroll_iters = int(min(max(args.preroll_ms / per, 1), 20000))
        ctx.factorize(preroll_iters, compute_err=False)
        preroll_ms = ctx.last_loop_ms()
        preroll_iters += 2

    # ---- warm-up (untimed) ----
    if args.warmup > 0:
        _, done, _ = ctx.factorize(args.warmup, compute_err=False)
        assert done == args.warmup

    # ---- timed region: exactly K steps, barrier + device sync on both sides ----
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
    all_ms = ctx.kernel_launch_ms()
    launch_ms = np.sort(all_ms)
    first_ms = [round(float(x), 4) for x in all_ms[:5]]
    ctx.profile_enable(False)

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
        ctx.set_w(host.pop("W0"))
        ctx.set_h(host.pop("H0"))
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
            mdl.factorize(niter=2, compute_err=False)
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
    host.clear()

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
        if algo_name == "SNMF":
            collective = "ncclAllReduce(V^T V), once per factorize()" if w.size > 1 else "none"
        else:
            collective = "ncclAllReduce(W^T V | W^T W), %d B/iter" % (4 * k * (n + k)) if w.size > 1 else "none"
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
            "rank_ms_per_step": [x / args.steps * 1e3 for x in rank_dt],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if algo_name == "NMFALS" else "f32",
            "data": data,
            "config": {"workload": "%s (BASELINE.json configs[%d]): %s %dx%d, k=%d, rows sharded over %d GPU(s), "
                                   "compute_err=False" % (args.config, cfg_index, cfg_desc, m, n, k, w.size),
                       "m": m, "n": n, "k": k, "algo": algo_name, "path": ctx.path_name,
                       "collective": collective,
                       "compute_err_true_iters_per_sec": rate_err,
                       "class_factorize": class_rate,
                       "from_random_start": rate_from_start},
            "roofline": {"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                         "traffic": traffic,
                         "traffic_source": ("profiles/traffic.json: rocprofv3 PMC passes of this kernel on this shape "
                                            "(2 x FETCH_SIZE + WRITE_SIZE per launch), collected by tools/pmc_configs.sh -- "
                                            "a recorded constant, not counted in this run") if traffic is not None
                                           else "none recorded for this kernel / shape / rank count",
                         "kernel": stats["name"], "launches": stats["launches"],
                         "mean_kernel_ms": stats["mean_ms"],
                         "min_kernel_ms": float(launch_ms[0]) if len(launch_ms) else None,
                         "median_kernel_ms": float(np.median(launch_ms)) if len(launch_ms) else None,
                         "max_kernel_ms": float(launch_ms[-1]) if len(launch_ms) else None,
                         "first_launches_ms": first_ms,
                         "flops_per_launch": fl,
                         "executed_flops_per_launch": ex,
                         "executed_TFLOPs": ex / mean_s / 1e12 if mean_s > 0 else 0.0,
                         "algorithmic_bytes_per_launch": by,
                         "achieved_hbm_GBs": by / mean_s / 1e9 if mean_s > 0 else 0.0,
                         "kernel_share_of_timed_region": stats["launches"] * mean_s / dt if dt > 0 else 0.0},
        }
        if algo_name == "NMFALS":
            # The QP kernel runs float64 VALU code: neither HBM nor MFMA bounds it.  Against the 78.6 TFLOP/s float64
            # vector peak: the FMA flop of k_nnqp_quad per problem in this bench's state -- B f once (2 * 64 * 64), then per
            # pass the correction over N and the product with the solution (2 * 64 * ns each), the LDL^T of inv(HA)[N,N]
            # (2 ns^3 / 3) and the two triangular solves (2 * 2 ns^2) -- with the RECORDED sizes of that state: ns = 8
            # unknowns, 1.9 passes (tools/quad_counts.py on a counting build, profiles/r03_experiments.md) and the recorded
            # VALU instruction count of a W-step launch (rocprofv3 SQ_INSTS_VALU, profiles/r03_pmc_cfg3_nnqp_quad.txt:
            # 1.16e8 per launch = 442 per problem).
            qps = float(hi - lo)
            quad = "quad" in stats["name"]
            if quad and k == 64:
                ns, passes = 8.0, 1.9
                fma_flop = (2.0 * 64.0 * 64.0 + passes * (4.0 * 64.0 * ns + 2.0 * ns ** 3 / 3.0 + 4.0 * ns * ns)) * qps
                valu_instr = 1.16e8 * qps / 262144.0
                src = "profiles/r03_pmc_cfg3_nnqp_quad.txt (SQ_INSTS_VALU of k_nnqp_quad in this bench's state)"
            else:                                 # lane-per-variable kernel: DESIGN.md 3.4, profiles/r02_pmc_summary.csv
                fma_flop = 2.0 * 2.24e5 * qps * (k / 64.0) ** 3
                valu_instr = 16.6e3 * qps * (k / 64.0) ** 2
                src = "profiles/r02_pmc_summary.csv (16.6 k VALU instructions per QP at k = 64, k_nnqp)"
            if mean_s > 0:
                out["roofline"].update(bound="valu_f64", achieved=fma_flop / mean_s / 1e12, peak=78.6, unit="TFLOP/s",
                                       frac=fma_flop / mean_s / 1e12 / 78.6)
                out["roofline"]["valu_issue"] = {"wave_instructions_per_launch": valu_instr,
                                                 "achieved_Ginstr_per_s": valu_instr / mean_s / 1e9,
                                                 "peak_Ginstr_per_s": 1024 * 2.4 / 4.0,
                                                 "frac": valu_instr / mean_s / 1e9 / (1024 * 2.4 / 4.0),
                                                 "source": src}
            out["roofline"]["note"] = ("float64 VALU kernel: `achieved` counts the FMA flop of the solves (recorded instruction counts, "
                                       "live launch time of the 16-slot-frame launch and the 32-slot one behind it); at three / two waves per SIMD the kernel is bound by the dependent chains of its "
                                       "factorisations and triangular solves, not by issue slots (valu_issue.frac)")
        if args.config == "cfg5":
            out["config"]["nnz_local"] = nnz_local
            out["config"]["gram_space_loop_w_once_per_factorize"] = gram_loop
            out["config"]["one_pass_csr_kernel_iters_per_sec"] = rate_pass_per_iter
            out["config"]["loop"] = ("k x n sized Gram-space iteration (P = M^T (V^T V), S = P M) + W = V M written in EVERY "
                                     "iteration, as the reference's update_w does" if stats["name"].startswith("k_csr_w")
                                     else "one pass over the CSR rows per iteration")
        if w.size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, m, n, k)
        print(json.dumps(out), flush=True)
    ctx.close()
    dist.shutdown()


if __name__ == "__main__":
    main()
