"""tntorch_amd -- MI355X-native TT orthogonalisation / rounding behind tntorch's API.

``import tntorch_amd as tn`` exposes the names of the reference (``tntorch/__init__.py:1-14``)
that sit on the TT decomposition / rounding hot path: ``tn.Tensor``, ``tn.round_tt``,
``tn.round``, ``tn.truncated_svd``, the unfoldings, plus the small helpers the reference's
tests use around them (``rand``/``randn``, ``dot``/``norm``/``relative_error``).
"""

from .tools import *  # noqa: F401,F403
from .round import *  # noqa: F401,F403
from .tensor import *  # noqa: F401,F403
from .create import *  # noqa: F401,F403
from .metrics import *  # noqa: F401,F403
from .matrix import *  # noqa: F401,F403
from . import dist_batch  # noqa: F401
from ._patch import patch  # noqa: F401

__version__ = "0.1.0"


def hip_available() -> bool:
    """True when the HIP kernel library has been built in-tree."""
    from . import _hip

    return _hip.available()
