"""pymf_amd -- MI355X-native factorize() hot path of nils-werner/pymf.

Drop-in classes for the reference's `pymf.NMF`, `pymf.NMFALS`, `pymf.SNMF`
(pymf/__init__.py:16-43 star-exports them the same way).
"""
from .nmf import NMF          # noqa: F401
from .nmfals import NMFALS    # noqa: F401
from .snmf import SNMF        # noqa: F401
from . import dist            # noqa: F401

__all__ = ["NMF", "NMFALS", "SNMF", "dist"]
__version__ = "0.1.0"
