"""``tntorch_amd.patch(tntorch)``: install the MI355X sweeps under the reference's own ``Tensor`` class.

The reference has no plugin registry: its boundary is the public Python API, so "dropping in" means
rebinding the hot-path methods of ``tntorch.Tensor`` (tensor.py:1771-2098) and the free function
``tntorch.truncated_svd`` (round.py:52-187).  A patched method wraps the object's cores / factors (shared, not
copied) into a ``tntorch_amd.Tensor``, runs this package's implementation (device cores -> HIP kernels, CPU cores ->
host mirror, or the reference's own code with ``cpu=False``) and rebinds ``self.cores`` / ``self.Us`` -- the same
list-entry rebinding the reference performs, so holders of the old core tensors see no change.
``tntorch.round_tt`` / ``round`` / ``round_tucker`` (round.py:7-49) clone and call the methods, so they follow.
"""

from typing import Callable

from .round import truncated_svd as _truncated_svd
from .tensor import Tensor

_METHODS = ("round_tt", "round_tucker", "round", "orthogonalize", "left_orthogonalize", "right_orthogonalize",
            "factor_orthogonalize")


def _wrap(ref_cls, name: str, cpu: bool):
    original = getattr(ref_cls, name)

    def method(self, *args, **kwargs):
        if not cpu and self.cores[0].device.type == "cpu":
            return original(self, *args, **kwargs)
        ours = Tensor(list(self.cores), Us=list(self.Us), idxs=self.idxs, batch=getattr(self, "batch", False))
        out = getattr(ours, name)(*args, **kwargs)
        self.cores, self.Us = ours.cores, ours.Us
        return out

    method.__name__ = name
    method.__doc__ = getattr(Tensor, name).__doc__
    method._tntorch_amd_original = original
    return method


_ACTIVE = {}  # id(module) -> (module, undo): a module is patched at most once


def patch(tn_module, cpu: bool = True) -> Callable[[], None]:
    """Rebind the orthogonalisation / rounding methods of ``tn_module.Tensor`` and ``tn_module.truncated_svd`` to this
    package.  ``cpu=False`` leaves CPU tensors to the reference's own code.  Returns a function that undoes the patch.

    Idempotent: patching an already patched module returns the SAME undo function (the first patch stays in force, its
    ``cpu`` setting included), so nested ``patch`` / ``undo`` pairs cannot restore half of the originals.
    Rebound names: the methods in ``_METHODS`` on ``tn_module.Tensor``, ``tn_module.truncated_svd`` and
    ``tn_module.round.truncated_svd``.  Modules that bound the function at import time with ``from .round import
    truncated_svd`` (the reference has none besides its ``__init__`` star import, which is the first of the two names)
    keep the original."""
    key = id(tn_module)
    if key in _ACTIVE and _ACTIVE[key][0] is tn_module:
        return _ACTIVE[key][1]
    ref_cls = tn_module.Tensor
    saved = {}
    for name in _METHODS:
        if hasattr(ref_cls, name):
            saved[name] = getattr(ref_cls, name)
            if hasattr(saved[name], "_tntorch_amd_original"):  # patched through another module object: keep the true original
                saved[name] = saved[name]._tntorch_amd_original
            setattr(ref_cls, name, _wrap(ref_cls, name, cpu))
    targets = [tn_module] + [m for m in (getattr(tn_module, "round", None),) if hasattr(m, "truncated_svd")]
    saved_fn = [(m, m.truncated_svd) for m in targets if hasattr(m, "truncated_svd")]
    for m, _ in saved_fn:
        m.truncated_svd = _truncated_svd

    def unpatch():
        if _ACTIVE.pop(key, None) is None:
            return  # already undone
        for name, fn in saved.items():
            setattr(ref_cls, name, fn)
        for m, fn in saved_fn:
            m.truncated_svd = fn

    _ACTIVE[key] = (tn_module, unpatch)
    return unpatch
