"""Where does a config-scale dense TT-SVD spend its time?  Wall time of every right-to-left step (one truncate() call each)
and the library's per-kernel-kind device time.  python tools/probes/dense_steps_probe.py 16 64 64 64 64 64 [--alg eig] [--batch B] [--rank r]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tntorch_amd as tn
from tntorch_amd import _hip, _hipops

alg = "svd"
batch = 0
rk = 16
args = sys.argv[1:]
if "--batch" in args:
    i = args.index("--batch"); batch = int(args[i + 1]); del args[i:i + 2]
if "--rank" in args:
    i = args.index("--rank"); rk = int(args[i + 1]); del args[i:i + 2]
if "--alg" in args:
    i = args.index("--alg"); alg = args[i + 1]; del args[i:i + 2]
shape = [int(a) for a in args] or [16, 64, 64, 64, 64]
X = torch.randn(([batch] if batch else []) + shape, device="cuda", dtype=torch.float32)
orig = _hipops.truncate
def timed(M, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _hip.prof_enable(True)
    out = orig(M, *a, **k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    prof = _hip.prof_collect(); _hip.prof_enable(False)
    kinds = ", ".join(f"{k2}: {v['ms']:.1f} ms/{v['launches']}" for k2, v in prof.items() if v["launches"])
    print(f"  truncate M={tuple(M.shape)} -> rank {out.rank}: {dt*1e3:.1f} ms wall | {kinds}", flush=True)
    return out
_hipops.truncate = timed
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = tn.Tensor(X, ranks_tt=rk, algorithm=alg, batch=bool(batch))
    torch.cuda.synchronize()
    print(f"rep {rep}: {time.perf_counter()-t0:.3f} s, ranks {t.ranks_tt.tolist()}, reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB", flush=True)
