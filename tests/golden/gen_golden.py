#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (nils-werner/pymf).

Runs only in the build container, where /root/reference is mounted.  The
reference is Python 2; it is imported UNMODIFIED through a shim:
  * builtins.xrange = range                       (nmf.py:182)
  * a synthetic empty package 'pymf' whose __path__ points at the reference so
    pymf/__init__.py (which imports cvxopt via nmfals.py:19) is never executed
  * for nmfnnls: module-level eager `map` (Py2 semantics, nmfnnls.py:73,80)
Only inputs (or their seeds) and outputs are written: tests/golden/*.npz.
"""
import builtins
import importlib
import os
import sys
import types

import numpy as np

REF = os.environ.get("PYMF_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    builtins.xrange = range
    pkg = types.ModuleType("pymf")
    pkg.__path__ = [os.path.join(REF, "pymf")]
    sys.modules["pymf"] = pkg
    mods = {}
    for name in ("nmf", "snmf", "nmfnnls", "bnmf", "rnmf", "nndsvd"):
        mods[name] = importlib.import_module("pymf." + name)
    mods["nmfnnls"].map = lambda f, *a: list(builtins.map(f, *a))
    return mods


def load_reference_nmfals():
    """pymf/nmfals.py ITSELF (nmfals.py:70-97), imported unmodified next to a stand-in for its one
    third-party dependency: `cvxopt` is absent from this container (and un-pinned in the reference's
    setup.py:12-16), so `cvxopt.base.matrix` is an ndarray wrapper and `cvxopt.solvers.qp(P, q, G, h)`
    returns the EXACT minimiser of the strictly convex QP the reference poses (G = -I, h = 0: x >= 0)
    through scipy.optimize.nnls on a square-root factor of P.  What these goldens pin is nmfals.py's own
    data flow -- HA / FA construction with its signs and float64 casts, the per-column scatter :75 and
    per-row scatter :90, the eager map -- NOT cvxopt's interior-point digits (still "parity unpinned",
    SURVEY 8(c)): IPM returns ~1e-9 positives where the exact minimiser has zeros."""
    import scipy.optimize

    class _Matrix(np.ndarray):
        pass

    def matrix(x, size=None):
        if size is not None:
            a = np.full(size, float(x), dtype=np.float64)
        else:
            a = np.array(x, dtype=np.float64)
            if a.ndim == 1:
                a = a.reshape(-1, 1)
        return a.view(_Matrix)

    def qp(P, q, G=None, h=None):
        P = np.asarray(P, dtype=np.float64)
        q = np.asarray(q, dtype=np.float64).reshape(-1)
        k = P.shape[0]
        assert np.array_equal(np.asarray(G), -np.eye(k)) and not np.any(np.asarray(h)), "only x >= 0 is posed (nmfals.py:79-80)"
        lam, Q = np.linalg.eigh((P + P.T) / 2.0)
        keep = lam > lam.max() * 1e-13
        A = (np.sqrt(lam[keep])[:, None]) * Q[:, keep].T          # A^T A = P on its range
        b = -(Q[:, keep].T.dot(q)) / np.sqrt(lam[keep])           # 1/2|Ax - b|^2 = 1/2 x'Px + q'x + const
        x, _ = scipy.optimize.nnls(A, b, maxiter=50 * k)
        return {"x": matrix(x), "status": "optimal"}

    cv = types.ModuleType("cvxopt")
    cv.base = types.ModuleType("cvxopt.base")
    cv.base.matrix = matrix
    cv.solvers = types.ModuleType("cvxopt.solvers")
    cv.solvers.qp = qp
    cv.solvers.options = {}
    sys.modules.update({"cvxopt": cv, "cvxopt.base": cv.base, "cvxopt.solvers": cv.solvers})
    mod = importlib.import_module("pymf.nmfals")
    mod.map = lambda f, *a: list(builtins.map(f, *a))             # Py2 eager map (nmfals.py:82,97)
    return mod


def run_case(cls, V, k, niter, seed, cast32, flags=None, w0=None, h0=None):
    """Run reference class; returns dict of arrays."""
    np.random.seed(seed)
    m, n = V.shape
    W0 = np.random.random((m, k)) if w0 is None else w0
    H0 = np.random.random((k, n)) if h0 is None else h0
    if cast32:
        W0 = W0.astype(np.float32)
        H0 = H0.astype(np.float32)
    mdl = cls(V, num_bases=k)
    mdl.W = W0.copy()
    mdl.H = H0.copy()
    flags = flags or {}
    mdl.factorize(niter=niter, **flags)
    out = dict(W0=W0, H0=H0, W=np.asarray(mdl.W), H=np.asarray(mdl.H))
    if hasattr(mdl, "ferr"):
        out["ferr"] = np.asarray(mdl.ferr, dtype=np.float64)
    return out


def main():
    mods = load_reference()
    NMF, SNMF, NNLS = mods["nmf"].NMF, mods["snmf"].SNMF, mods["nmfnnls"].NMFNNLS
    BNMF = mods["bnmf"].BNMF
    cases = {}
    only = sys.argv[1:]          # e.g. `gen_golden.py nndsvd` rewrites only the nndsvd_* fixtures

    def wanted(name):
        return not only or any(name.startswith(o) for o in only)

    def add(name, cls, V, vdesc, k, niter, seed, cast32, **kw):
        if not wanted(name):
            return
        r = run_case(cls, V, k, niter, seed, cast32, **kw)
        r.update(vdesc)
        r["k"] = np.int64(k)
        r["niter"] = np.int64(niter)
        r["seed"] = np.int64(seed)
        cases[name] = r

    # (i) BASELINE cfg1: 100x50, k=4, niter=50; float64-default init and fp32 init
    V1 = np.random.RandomState(20260101).random_sample((100, 50)).astype(np.float32)
    d1 = dict(V=V1)
    for cname, cls in (("nmf", NMF), ("snmf", SNMF)):
        add(cname + "_cfg1_f64", cls, V1, d1, 4, 50, 42, False)
        add(cname + "_cfg1_f32", cls, V1, d1, 4, 50, 42, True)
    # (ii) 512x128, k=16, 10 iterations, fp32
    V2 = np.random.RandomState(7).random_sample((512, 128)).astype(np.float32)
    for cname, cls in (("nmf", NMF), ("snmf", SNMF)):
        add(cname + "_512x128_k16", cls, V2, dict(V=V2), 16, 10, 42, True)
    # (iii) shrunken analogues of cfg2/cfg4 (V stored by seed only)
    for (m, n, k, it, tag) in ((2048, 256, 64, 8, "cfg4s"), (1024, 512, 32, 8, "cfg2s"),
                               (384, 128, 128, 6, "cfg5s_dense")):
        Vs = np.random.RandomState(1234).random_sample((m, n)).astype(np.float32)
        desc = dict(V_seed=np.int64(1234), V_shape=np.array([m, n], dtype=np.int64))
        add("nmf_" + tag, NMF, Vs, desc, k, it, 42, True)
        add("snmf_" + tag, SNMF, Vs, desc, k, it, 42, True)
    # sparse-like data densified (cfg5 analogue: ~1% nnz)
    rs = np.random.RandomState(99)
    Vsp = (rs.random_sample((768, 128)) < 0.01) * rs.random_sample((768, 128))
    Vsp = Vsp.astype(np.float32)
    add("snmf_sparse1pct", SNMF, Vsp, dict(V=Vsp), 16, 6, 42, True)
    # ragged / odd shapes (padding paths)
    V3 = np.random.RandomState(5).random_sample((37, 29)).astype(np.float32)
    add("nmf_37x29_k5", NMF, V3, dict(V=V3), 5, 12, 3, False)
    add("snmf_37x29_k5", SNMF, V3, dict(V=V3), 5, 12, 3, False)
    # (vi) the reference test's own data: tests/test_pymf.py:32-33,69,77
    np.random.seed(400401)
    A = np.random.random((3, 50)) + 2.0
    add("nmf_reftest", NMF, A, dict(V=A), 4, 20, 11, False)
    add("snmf_reftest", SNMF, A, dict(V=A), 4, 20, 11, False)
    # (iv) flag variants + repeated calls (tests/test_pymf.py:92-95)
    np.random.seed(21)
    mdl = NMF(V1, num_bases=4)
    mdl.factorize(niter=5)
    seq = dict(V=V1, W0=None)
    w_a, h_a, f_a = mdl.W.copy(), mdl.H.copy(), mdl.ferr.copy()
    mdl.factorize(niter=5, compute_h=False)
    w_b, h_b, f_b = mdl.W.copy(), mdl.H.copy(), mdl.ferr.copy()
    mdl.factorize(niter=5, compute_w=False)
    w_c, h_c, f_c = mdl.W.copy(), mdl.H.copy(), mdl.ferr.copy()
    mdl.factorize(niter=5, compute_err=False)
    w_d, h_d, f_d = mdl.W.copy(), mdl.H.copy(), mdl.ferr.copy()
    cases["nmf_flagseq"] = dict(V=V1, seed=np.int64(21), k=np.int64(4),
                                W_a=w_a, H_a=h_a, ferr_a=f_a, W_b=w_b, H_b=h_b, ferr_b=f_b,
                                W_c=w_c, H_c=h_c, ferr_c=f_c, W_d=w_d, H_d=h_d, ferr_d=f_d)
    # (v) early exit: exact data, compute_w=False -> len(ferr)==2 (nmf.py:198-202)
    Vx = np.array([[1.0, 0.0, 2.0, 1.0, 0.5], [0.0, 1.0, 1.0, 3.0, 0.25]])
    mdl = NMF(Vx, num_bases=2)
    mdl.W = np.array([[1.0, 0.0], [0.0, 1.0]])
    np.random.seed(1)
    mdl.H = np.random.random((2, 5))
    H0x = mdl.H.copy()
    mdl.factorize(niter=20, compute_w=False)
    cases["nmf_earlyexit"] = dict(V=Vx, W0=np.eye(2), H0=H0x, W=mdl.W, H=mdl.H,
                                  ferr=mdl.ferr, k=np.int64(2), niter=np.int64(20))
    # NMFALS proxy: the reference's NNLS sibling minimises the same objective
    V4 = np.random.RandomState(8).random_sample((24, 18)).astype(np.float64)
    add("nnls_24x18_k4", NNLS, V4, dict(V=V4), 4, 5, 9, False)
    np.random.seed(400401)
    A = np.random.random((3, 50)) + 2.0
    add("nnls_reftest", NNLS, A, dict(V=A), 4, 10, 11, False)

    # NMFALS through nmfals.py itself (exact-QP stand-in for cvxopt, see load_reference_nmfals)
    ALS = load_reference_nmfals().NMFALS
    add("nmfals_24x18_k4", ALS, V4, dict(V=V4), 4, 5, 9, False)
    add("nmfals_reftest", ALS, A, dict(V=A), 4, 10, 11, False)
    V5 = np.random.RandomState(12).random_sample((130, 90)).astype(np.float32)
    add("nmfals_130x90_k33", ALS, V5, dict(V=V5), 33, 3, 5, False)
    Vc3 = np.random.RandomState(1234).random_sample((2048, 1024)).astype(np.float32)      # cfg3's width and k
    add("nmfals_cfg3s", ALS, Vc3, dict(V_seed=np.int64(1234), V_shape=np.array([2048, 1024], dtype=np.int64)),
        64, 2, 42, False)

    # Round 4: the SAME inputs through pymf/nmfnnls.py -- the reference's own NNLS class, REAL scipy.optimize.nnls, no
    # stand-in anywhere -- which minimises the identical objective row by row / column by column (nmfnnls.py:69-80):
    # these pin NMFALS' NUMBERS at the shapes that matter (cfg3's width and num_bases among them); the nmfals_* fixtures
    # above (stub cvxopt) pin nmfals.py's data flow only.
    add("nnls_130x90_k33", NNLS, V5, dict(V=V5), 33, 3, 5, False)
    add("nnls_cfg3s", NNLS, Vc3, dict(V_seed=np.int64(1234), V_shape=np.array([2048, 1024], dtype=np.int64)), 64, 2, 42, False)
    Vq_, dq_ = (np.random.RandomState(78).random_sample((300, 200))).astype(np.float32), \
        dict(V_seed=np.int64(78), V_shape=np.array([300, 200], dtype=np.int64), V_shift=np.float64(0.0))
    add("nnls_300x200_k72", NNLS, Vq_, dq_, 72, 2, 42, False)
    Vq_, dq_ = (np.random.RandomState(79).random_sample((260, 300))).astype(np.float32), \
        dict(V_seed=np.int64(79), V_shape=np.array([260, 300], dtype=np.int64), V_shift=np.float64(0.0))
    add("nnls_260x300_k130", NNLS, Vq_, dq_, 130, 2, 42, False)

    # cfg5's shape class (k = n = 128): H H^T of a square uniform H has cond ~ 1e7.  Two iterations (no
    # convergence test can fire before i > 1, so both runs execute exactly two) of reference SNMF
    # (snmf.py:67-91) with float64-default operands = the golden; the SAME run with all-float32 operands
    # is kept beside it (W32, H32, ferr32): the distance between the two is what the reference's own
    # float32 path loses at this conditioning.  Dense U[0,1) data, and cfg5's sparse pattern (1 % nnz)
    # densified -- the reference cannot take scipy.sparse input, semantics = SNMF on V.toarray().
    import scipy.sparse as sp
    Vs5 = np.random.RandomState(1234).random_sample((384, 128)).astype(np.float32)
    Vcsr = sp.random(2048, 128, density=0.01, format="csr", dtype=np.float32, random_state=np.random.RandomState(1234))
    Vd = np.asarray(Vcsr.toarray(), dtype=np.float32)
    for tag, Vk, desc in (("snmf_cfg5s_dense_f64", Vs5, dict(V_seed=np.int64(1234), V_shape=np.array([384, 128], dtype=np.int64))),
                          ("snmf_csr_k128_f64", Vd, dict(V=Vd))):
        add(tag, SNMF, Vk, desc, 128, 2, 42, False)
        if wanted(tag):
            r32 = run_case(SNMF, Vk, 128, 2, 42, True)
            cases[tag].update(W32=r32["W"], H32=r32["H"], ferr32=r32["ferr"])

    # BNMF ("next" row 1): binary data, the reference test's own input (tests/test_pymf.py:80)
    Vb = (np.random.RandomState(17).random_sample((96, 64)) < 0.3).astype(np.float32)
    add("bnmf_96x64_k8", BNMF, Vb, dict(V=Vb), 8, 15, 42, False)
    add("bnmf_96x64_k8_f32", BNMF, Vb, dict(V=Vb), 8, 15, 42, True)
    np.random.seed(400401)
    A = np.random.random((3, 50)) + 2.0
    add("bnmf_reftest", BNMF, np.round(A - 2.0), dict(V=np.round(A - 2.0)), 4, 20, 11, False)
    Vb2 = (np.random.RandomState(5).random_sample((1024, 256)) < 0.2).astype(np.float32)
    add("bnmf_1024x256_k64", BNMF, Vb2, dict(V_seed=np.int64(5), V_shape=np.array([1024, 256], dtype=np.int64)),
        64, 8, 42, True)

    # RNMF ("next" row 3): lazy init is part of the behaviour (rnmf.py:81-94), so the goldens are
    # generated the way users call it: seed, construct, factorize; W/H/S come out of init_w/init_h
    RNMF = mods["rnmf"].RNMF
    for tag, (m_, n_, k_, lamb_, it_) in {"rnmf_60x40_k4": (60, 40, 4, 0.3, 12),
                                          "rnmf_300x256_k32": (300, 256, 32, 2.0, 10),
                                          "rnmf_300x256_k8": (300, 256, 8, 1.0, 10)}.items():
        rs_ = np.random.RandomState(31)
        Vr = rs_.random_sample((m_, n_))
        out_idx = rs_.randint(0, m_ * n_, size=max(3, m_ * n_ // 200))
        Vr.flat[out_idx] += 4.0 * rs_.random_sample(out_idx.shape[0]) + 1.0      # sparse outliers
        Vr = Vr.astype(np.float32)
        np.random.seed(7)
        mdl = RNMF(Vr, num_bases=k_, lamb=lamb_)
        mdl.factorize(niter=it_)
        cases[tag] = dict(V=Vr, W=mdl.W, H=mdl.H, S=mdl.S, ferr=np.asarray(mdl.ferr, dtype=np.float64),
                          k=np.int64(k_), niter=np.int64(it_), seed=np.int64(7), lamb=np.float64(lamb_))

    # NNDSVD ("next" row 4): deterministic (no RNG); tall, wide, ragged, the docstring example
    # (nndsvd.py:50-53), and the documented use -- NMF started from the NNDSVD factors (:59-64)
    NNDSVD = mods["nndsvd"].NNDSVD
    nd_inputs = {
        "nndsvd_300x40_k6": (np.random.RandomState(3).random_sample((300, 40)).astype(np.float32), 6),
        "nndsvd_40x300_k6": (np.random.RandomState(3).random_sample((40, 300)).astype(np.float32), 6),
        "nndsvd_cfg1_k4": (V1, 4),
        "nndsvd_37x29_k5": (V3, 5),
        "nndsvd_doc_k2": (np.array([[1.0, 0.0, 2.0], [0.0, 1.0, 1.0]]), 2),
        "nndsvd_1024x256_k64": (np.random.RandomState(1234).random_sample((1024, 256)).astype(np.float32), 64),
    }
    for tag, (Vn, k_) in nd_inputs.items():
        mdl = NNDSVD(Vn, num_bases=k_)
        mdl.factorize()
        d = dict(W=mdl.W, H=mdl.H, ferr=np.asarray(mdl.ferr, dtype=np.float64), k=np.int64(k_))
        if tag == "nndsvd_1024x256_k64":
            d.update(V_seed=np.int64(1234), V_shape=np.array([1024, 256], dtype=np.int64))
        else:
            d["V"] = Vn
        if tag == "nndsvd_300x40_k6":
            nmf_mdl = NMF(Vn, num_bases=k_)
            nmf_mdl.W = mdl.W.copy()
            nmf_mdl.H = mdl.H.copy()
            nmf_mdl.factorize(niter=10)
            d.update(W_nmf10=nmf_mdl.W, H_nmf10=nmf_mdl.H, ferr_nmf10=np.asarray(nmf_mdl.ferr))
        cases[tag] = d

    # num_bases beyond the register-resident kernels (SNMF/RNMF/NNDSVD > 128, NMFALS > 64): the reference has
    # no limit (snmf.py:69-70, nmfals.py:78-80, rnmf.py:109-115).  V by seed; mixed-sign data for SNMF.
    def seeded(m_, n_, seed_, shift=0.0):
        Vq = (np.random.RandomState(seed_).random_sample((m_, n_)) - shift).astype(np.float32)
        return Vq, dict(V_seed=np.int64(seed_), V_shape=np.array([m_, n_], dtype=np.int64), V_shift=np.float64(shift))
    Vq, dq = seeded(512, 320, 77, 0.3)
    add("bigk_snmf_512x320_k160", SNMF, Vq, dq, 160, 4, 42, False)
    Vq, dq = seeded(300, 200, 78)
    add("bigk_nmfals_300x200_k72", ALS, Vq, dq, 72, 2, 42, False)
    Vq, dq = seeded(260, 300, 79)
    add("bigk_nmfals_260x300_k130", ALS, Vq, dq, 130, 2, 42, False)
    Vq, dq = seeded(500, 300, 80)
    mdl = NNDSVD(Vq, num_bases=150)
    mdl.factorize()
    cases["bigk_nndsvd_500x300_k150"] = dict(W=mdl.W, H=mdl.H, ferr=np.asarray(mdl.ferr, dtype=np.float64), k=np.int64(150), **dq)
    rs_ = np.random.RandomState(81)
    Vr = rs_.random_sample((300, 256))
    out_idx = rs_.randint(0, Vr.size, size=Vr.size // 200)
    Vr.flat[out_idx] += 4.0 * rs_.random_sample(out_idx.shape[0]) + 1.0
    Vr = Vr.astype(np.float32)
    np.random.seed(7)
    mdl = RNMF(Vr, num_bases=140, lamb=1.0)
    mdl.factorize(niter=6)
    cases["bigk_rnmf_300x256_k140"] = dict(V=Vr, W=mdl.W, H=mdl.H, S=mdl.S, ferr=np.asarray(mdl.ferr, dtype=np.float64),
                                           k=np.int64(140), niter=np.int64(6), seed=np.int64(7), lamb=np.float64(1.0))

    for name, d in cases.items():
        if only and not any(name.startswith(o) for o in only):
            continue
        d = {k: v for k, v in d.items() if v is not None}
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        f = d.get("ferr")
        print("%-24s ferr[-1]=%s len=%s" % (name, None if f is None else "%.9f" % f[-1],
                                             None if f is None else len(f)))


if __name__ == "__main__":
    main()
