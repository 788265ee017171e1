#!/usr/bin/env python3
"""bench.py -- factorize() iterations/sec of pymf's NMF hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric, configs[3] = "cfg4"): NMF multiplicative update on
a dense float32 V of 1,048,576 x 256 at k = 64; the rows of V and W are sharded over
the N ranks (fixed total problem => "strong" scaling), H is replicated, and one RCCL
all-reduce of (W^T V | W^T W) -- 80 KiB -- runs per iteration.  A "step" is one
factorize() iteration: update_w + update_h (compute_err=False, the pure update
path; the compute_err=True rate is reported alongside).  Inputs follow
BASELINE.md section 3: V = RandomState(1234).random_sample((m, n)).astype(float32),
np.random.seed(42), W0 then H0 from np.random.random.  All inputs are resident in
HBM before the timed region starts.

Prints ONE JSON line on rank 0 (contract in the task description), including
  "roofline":     live HIP-event timing of the dominant kernel vs the fp32 MFMA peak
  "cpu_baseline": the NumPy oracle (oracle/, the reference's op order) timed on
                  this host's cores on a bounded row sample of the same workload.
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

M_FULL, N_FULL, K_FULL = 1048576, 256, 64
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0
PREROLL_MS_DEFAULT = 300.0   # measured (gpurun_out/r2a, profiles/r02_ramp.md): a fresh box needs ~30 launches to reach its clock


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


def _time_oracle(V, W, H, budget_s):
    from oracle import nmf_update_w, nmf_update_h
    nmf_update_w(V, W, H)
    nmf_update_h(V, W, H)                     # warm-up iteration
    t0 = time.time()
    it = 0
    while True:
        nmf_update_w(V, W, H)
        nmf_update_h(V, W, H)
        it += 1
        dt = time.time() - t0
        if (it >= 3 and dt > budget_s * 0.5) or dt > budget_s or it >= 50:
            break
    return it, dt


def cpu_baseline(m, n, k, budget_s=20.0):
    """The oracle (NumPy restatement of pymf/nmf.py:122-132) on a bounded row sample, in the two
    variants BASELINE.md section 3 asks for: reference-default (float64 W/H, float32 V) = `value`,
    and all-float32."""
    ms = min(m, 32768)
    V = np.random.RandomState(1234).random_sample((ms, n)).astype(np.float32)
    np.random.seed(42)
    W = np.random.random((ms, k))             # float64, the reference's default init
    H = np.random.random((k, n))
    it, dt = _time_oracle(V, W.copy(), H.copy(), budget_s * 0.7)
    it32, dt32 = _time_oracle(V, W.astype(np.float32), H.astype(np.float32), budget_s * 0.3)
    rate_sample = it / dt
    threads = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info
        for info in threadpool_info():
            if info.get("user_api") == "blas":
                threads = int(info.get("num_threads", threads))
    except Exception:
        pass
    return {
        "value": rate_sample * ms / float(m),
        "unit": "iter/s",
        "cores": threads,
        "kind": "port",
        "value_all_fp32": (it32 / dt32) * ms / float(m),
        "sample": "oracle NMF (reference op order, float64 W/H, float32 V) on the first %d of %d rows, "
                  "n=%d k=%d: %d iterations in %.2f s = %.3f iter/s on the sample, scaled by %d/%d "
                  "(cost is linear in rows); all-float32 variant: %d iterations in %.2f s; numpy %s, "
                  "host cpu_count=%d" %
                  (ms, m, n, k, it, dt, rate_sample, ms, m, it32, dt32, np.__version__, os.cpu_count() or 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", dest="m", type=int, default=M_FULL)
    ap.add_argument("--cols", dest="n", type=int, default=N_FULL)
    ap.add_argument("--bases", dest="k", type=int, default=K_FULL)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preroll-ms", type=float, default=PREROLL_MS_DEFAULT,
                    help="disclosed device pre-conditioning: untimed iterations of the same loop for about this "
                         "many ms of device time BEFORE the W counted warm-up steps (the chip's clock/power "
                         "state needs more than 5 launches to settle on a fresh box); 0 disables")
    ap.add_argument("--fill", choices=["numpy", "device"], default="numpy",
                    help="numpy: BASELINE.md protocol; device: counter-based fill (fast start-up)")
    ap.add_argument("--debug-share-gpu", action="store_true",
                    help="plumbing check on a 1-GPU box: every rank uses device 0 and its own 1-rank RCCL "
                         "communicator (rows still sharded, gloo barriers still used); NOT a measurement")
    args = ap.parse_args()

    from pymf_amd import _lib, dist
    w = dist.init_from_env()
    if w.size != args.gpus and w.rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, w.size),
              file=sys.stderr)
    m, n, k = args.m, args.n, args.k
    lo, hi = w.row_range(m)
    if args.debug_share_gpu:
        ctx = _lib.Context(_lib.ALGO_NMF, hi - lo, n, k, device=0, rank=0, nranks=1,
                           nccl_id=_lib.nccl_unique_id())
    else:
        ctx = _lib.Context(_lib.ALGO_NMF, hi - lo, n, k, device=w.local_rank, rank=w.rank,
                           nranks=w.size, nccl_id=w.nccl_id)
    if args.fill == "numpy":
        V = gen_rows(np.random.RandomState(1234), m, n, lo, hi)
        ctx.set_v_dense(V)
        del V
        np.random.seed(42)
        rs = np.random.mtrand._rand            # the global legacy stream, as np.random.random uses
        ctx.set_w(gen_rows(rs, m, k, lo, hi))
        ctx.set_h(np.random.random((k, n)))
    else:
        ctx.fill_v_uniform(1234, lo)
        ctx.fill_w_uniform(42, lo)
        ctx.fill_h_uniform(43)

    # ---- disclosed pre-conditioning (untimed, reported as "preroll_ms" / "preroll_iters") ----
    preroll_iters, preroll_ms = 0, 0.0
    if args.preroll_ms > 0:
        ctx.factorize(2, compute_err=False)
        per = max(ctx.last_loop_ms() / 2.0, 1e-3)
        preroll_iters = int(min(max(args.preroll_ms / per, 1), 20000))
        ctx.factorize(preroll_iters, compute_err=False)
        preroll_ms = ctx.last_loop_ms()
        preroll_iters += 2

    # ---- warm-up (untimed) ----
    _, done, _ = ctx.factorize(args.warmup, compute_err=False)
    assert done == args.warmup

    # ---- timed region: exactly K steps, barrier + device sync on both sides ----
    ctx.profile_enable(True)
    ctx.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    _, done, conv = ctx.factorize(args.steps, compute_err=False)
    ctx.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    assert done == args.steps and conv < 0, "timed run was shortened (%d of %d)" % (done, args.steps)
    dt = dist.allreduce_max(dt)
    stats = ctx.kernel_stats()
    launch_ms = np.sort(ctx.kernel_launch_ms())
    first_ms = [round(float(x), 4) for x in ctx.kernel_launch_ms()[:5]]
    ctx.profile_enable(False)

    # secondary: the API-default compute_err=True rate (not the headline value)
    ne = max(2, min(args.steps, 20))
    ctx.synchronize()
    dist.barrier()
    t1 = time.perf_counter()
    _, done_e, conv_e = ctx.factorize(ne, compute_err=True)
    ctx.synchronize()
    dist.barrier()
    dte = dist.allreduce_max(time.perf_counter() - t1)

    if w.rank == 0:
        ach = (stats["flops_per_launch"] / (stats["mean_ms"] * 1e-3)) / 1e12 if stats["mean_ms"] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "%s@%dx%dx%d/%d" % (stats["name"], m, n, k, w.size)
                traffic = tj.get(key)
            except Exception:
                traffic = None
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
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cfg4: pymf.NMF multiplicative update, dense fp32 V %dx%d, k=%d, rows "
                                   "sharded over %d GPU(s), compute_err=False" % (m, n, k, w.size),
                       "m": m, "n": n, "k": k, "algo": "NMF", "path": ctx.path_name,
                       "collective": "ncclAllReduce(W^T V | W^T W), %d B/iter" % (4 * k * (n + k))
                       if w.size > 1 else "none",
                       "compute_err_true_iters_per_sec": done_e / dte},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                         "kernel": stats["name"], "launches": stats["launches"],
                         "mean_kernel_ms": stats["mean_ms"],
                         "min_kernel_ms": float(launch_ms[0]) if len(launch_ms) else None,
                         "median_kernel_ms": float(np.median(launch_ms)) if len(launch_ms) else None,
                         "max_kernel_ms": float(launch_ms[-1]) if len(launch_ms) else None,
                         "first_launches_ms": first_ms,
                         "flops_per_launch": stats["flops_per_launch"],
                         "algorithmic_bytes_per_launch": stats["bytes_per_launch"],
                         "achieved_hbm_GBs": stats["bytes_per_launch"] / (stats["mean_ms"] * 1e-3) / 1e9
                         if stats["mean_ms"] > 0 else 0.0},
        }
        if w.size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(m, n, k)
        print(json.dumps(out), flush=True)
    ctx.close()
    dist.shutdown()


if __name__ == "__main__":
    main()
