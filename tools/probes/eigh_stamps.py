"""Cycle stamps inside the tridiagonal eigensolver (library built with -DTTR_EIGH_STAMPS, TTR_LIB_PATH): phases of matrix 0
and the split of the QL phase into recurrence / rotation application.   python tools/probes/eigh_stamps.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tntorch_amd import _hip
L = _hip.lib()
torch.manual_seed(0)
for B in (1, 2048):
    for kind in ("rank32", "full"):
        M = torch.randn(B, 64, 2048, device="cuda")
        if kind == "rank32":
            M[:, 32:] = M[:, :32] + 1e-7 * torch.randn(B, 32, 2048, device="cuda")
        G = _hip.gemm(M, M, transB=True)
        n = 64
        V = torch.empty((B, n, n), device="cuda"); sig = torch.empty((B, n), device="cuda"); info = torch.empty((B,), dtype=torch.int32, device="cuda")
        ws = torch.zeros(64, dtype=torch.int64, device="cuda")
        for _ in range(2):
            code = L.ttr_eigh_trunc(_hip.dtype_code(G.dtype), n, B, G.data_ptr(), n, n * n, 1, 0, V.data_ptr(), n, n * n, sig.data_ptr(), n,
                                    info.data_ptr(), _hip.EIG_RAW, 0, 0.0, None, 64, _hip.SOLVER_TRIDIAG, None, None, None, 0, ws.data_ptr(), 64 * 8, _hip._stream())
            assert code == 0
        torch.cuda.synchronize()
        st = ws.cpu().tolist()
        t = st[:4]
        print(f"B={B} {kind}: tridiag {t[1]-t[0]} cyc, Q formation {t[2]-t[1]}, QL {t[3]-t[2]} (recurrence {st[6]}, apply {st[7]}; "
              f"{st[4]} iterations, {st[5]} rotations -> {st[6]/max(st[5],1):.0f} + {st[7]/max(st[5],1):.0f} cycles per rotation)")
