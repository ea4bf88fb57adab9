import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _restore_library_knobs(request):
    """GPU tests that turn the library's diagnostic knobs (ttr_debug_set_knob) must not leak their setting into the tests that
    run after them in the same process: the packing / rank-skip knobs are put back to their defaults after every GPU test."""
    yield
    if "gpu" not in request.keywords:
        return
    import torch

    if not torch.cuda.is_available():
        return
    from tntorch_amd import _hip, _hipops

    if _hip._lib is None:   # nothing loaded, nothing to restore
        return
    _hip.set_knob(_hip.KNOB_QR_RANK_SKIP, 8)
    _hip.set_knob(_hip.KNOB_QR_PACK, 0 if _hipops._FUSE_APPLY_GRAM else 3)
    for knob, default in ((_hip.KNOB_RANK_NOISE_FLOOR, 0 if os.environ.get("TTR_STRICT_RANKS", "1") == "0" else 1), (_hip.KNOB_ORTH_ROUNDS, 4), (_hip.KNOB_QR_PACK_PRE, 1), (_hip.KNOB_EIGH_BIG_OCC, 0), (_hip.KNOB_QR_STAGGER, 0), (_hip.KNOB_JACOBI_LIVE_WAVE, 0),
                          (_hip.KNOB_ORTH_V2, 2), (_hip.KNOB_EIGH_SMALL, 2), (_hip.KNOB_QR_INTERLEAVE, 1), (_hip.KNOB_SWEEP_STAGGER, 1), (_hip.KNOB_ORTH_SPLIT, int(os.environ.get("TTR_ORTH_SPLIT", "2048") or 2048))):
        _hip.set_knob(knob, default)
    _hipops.SWEEP_C_ENABLED = os.environ.get("TTR_SWEEP_C", "1") != "0"
