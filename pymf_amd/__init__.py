"""pymf_amd -- MI355X-native factorize() hot path of nils-werner/pymf.

Drop-in classes for the reference's `pymf.NMF`, `pymf.NMFALS`, `pymf.SNMF`
(pymf/__init__.py:16-43 star-exports them the same way).
"""
from .nmf import NMF          # noqa: F401
from .nmfals import NMFALS    # noqa: F401
from .snmf import SNMF        # noqa: F401
from .nmfnnls import NMFNNLS  # noqa: F401  (SURVEY 8(f) 'next' row 2)
from .bnmf import BNMF        # noqa: F401  (SURVEY 8(f) 'next' row 1)
from .nndsvd import NNDSVD    # noqa: F401  (SURVEY 8(f) 'next' row 4)
from . import dist            # noqa: F401

__all__ = ["NMF", "NMFALS", "SNMF", "NMFNNLS", "BNMF", "NNDSVD", "dist"]
__version__ = "0.1.0"
