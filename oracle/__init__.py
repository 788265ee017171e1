"""CPU oracle for the pymf factorize() hot path -- TEST INFRASTRUCTURE ONLY.

NumPy restatement of the reference's update rules (nils-werner/pymf:
pymf/nmf.py, pymf/snmf.py, pymf/nmfals.py), written from scratch in the
reference's operation order.  Nothing in the shipped package (pymf_amd/)
imports this; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg do, and only as the checker / reported CPU baseline.

Parity pin: NMF and SNMF are checked against golden vectors produced by
importing the real reference in the build container
(tests/golden/gen_golden.py -> tests/golden/*.npz, tests/test_oracle_golden.py).
NMFALS: the arithmetic lives in cvxopt (un-pinned in the reference's setup.py,
absent here) -> "parity unpinned" for the cvxopt interior-point digits; the
oracle solves the same strictly convex QP exactly (active set, float64) and is
pinned against the reference's own NNLS sibling (pymf/nmfnnls.py), which
minimises the identical objective.
"""
from .nmf_oracle import NMFOracle, nmf_update_w, nmf_update_h, frobenius_norm  # noqa: F401
from .snmf_oracle import SNMFOracle, snmf_update_w, snmf_update_h  # noqa: F401
from .bnmf_oracle import BNMFOracle  # noqa: F401
from .rnmf_oracle import RNMFOracle, soft_thresholding  # noqa: F401
from .nndsvd_oracle import NNDSVDOracle, nndsvd_closed_form, dense_svd  # noqa: F401
from .nmfals_oracle import NMFALSOracle, nnqp_solve, als_update_w, als_update_h  # noqa: F401
