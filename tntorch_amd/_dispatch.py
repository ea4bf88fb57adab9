"""Backend selection: CPU tensors -> host mirror, device tensors -> HIP kernels (no fallback)."""

import torch

from . import _hostops


def ops_for(t: torch.Tensor):
    """Return the module implementing the sweeps for tensor ``t``.

    A CUDA/HIP tensor ALWAYS maps to the hand-written kernels; if the library is missing
    or the dtype is unsupported this raises -- it never silently computes on the CPU or
    through torch's own GPU linear algebra.
    """
    if t.device.type == "cpu":
        return _hostops
    if t.device.type != "cuda":
        raise RuntimeError(f"tntorch_amd: unsupported device {t.device}")
    if getattr(t, "requires_grad", False):
        raise NotImplementedError("tntorch_amd: the HIP kernels are not differentiable; detach the tensor or use CPU tensors")
    from . import _hip, _hipops

    _hip.lib()  # raises RuntimeError if libttround_hip.so has not been built
    _hip.dtype_code(t.dtype)  # raises TypeError for dtypes the kernels do not implement
    # the sub-batch streams are created on first contact with a device: HIP assigns hardware queues round-robin at
    # stream creation (4 by default), so early streams are less likely to end up sharing one
    _hipops._side_streams(t.device, 2)
    return _hipops
