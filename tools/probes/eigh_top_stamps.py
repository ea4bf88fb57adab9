"""Cycle stamps inside the top-r first pass (ttr_eigh_top; library built with -DTTR_EIGH_STAMPS:
    bash tools/build_variant.sh stamps ttr_eigh.hip -DTTR_EIGH_STAMPS;  TTR_LIB_PATH=tntorch_amd/libttround_stamps.so python tools/probes/eigh_top_stamps.py
Matrix 0 of a launch: wave 0 (tridiagonalisation, eigenvalues, twisted factorisations) and wave 1 (Q formation, eigenvalues,
back-transformation + Newton-Schulz on the matrix cores), for a full 64 x 64 problem and for one that shrinks to 32 x 32."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip

L = _hip.lib()
L.ttr_debug_set_eigh_stamps.argtypes = [ctypes.c_void_p]
L.ttr_debug_set_eigh_stamps.restype = None
torch.manual_seed(0)
for B in (1, 2048):
    for kind in ("full 64", "zero tail (32)"):
        M = torch.randn(B, 64, 2048, device="cuda")
        if kind != "full 64":
            M[:, 32:] = 0
        G = _hip.rowgram(M)
        buf = torch.zeros(64, dtype=torch.int64, device="cuda")
        _hip.eigh_top(G, 32, 0.125)
        torch.cuda.synchronize()
        L.ttr_debug_set_eigh_stamps(buf.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        V, sg, info, flat = _hip.eigh_top(G, 32, 0.125)
        e1.record()
        torch.cuda.synchronize()
        L.ttr_debug_set_eigh_stamps(None)
        st = buf.cpu().tolist()
        w0 = [x for x in st[:32] if x]
        w1 = [x for x in st[32:] if x]
        t0 = w0[0]
        print(f"B={B} {kind}: launch {e0.elapsed_time(e1) * 1e3:.0f} us, G parts {G.shape[1] if G.dim() == 4 else 1}, flat {int(flat[0])}")
        print("   wave 0 [start, tridiag end, (dup), eigenvalues + factorisation, norms, barrier B1, vectors written, barrier B2, end]:",
              [x - t0 for x in w0])
        print("   wave 1 [Q formed, eigenvalues + factorisation, vectors written, barrier B2, Z^T Q^T, S + Newton-Schulz, stored]:",
              [x - t0 for x in w1])
