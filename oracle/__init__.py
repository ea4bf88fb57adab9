"""TEST INFRASTRUCTURE ONLY -- CPU restatement of tntorch's TT rounding path.

Nothing under ``tntorch_amd/`` may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
only as the checker / the timed CPU baseline -- never as the product path.
"""

from .tt_oracle import *  # noqa: F401,F403
