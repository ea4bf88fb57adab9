"""Where the dense configs spend their time (GPU): per-kind kernel time of one C3 share / one C1-class step for a few
settings of the block-Jacobi inner sweep count and with / without the 128 x 128-tile GEMM; raw GEMM rates of the C1 shapes."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tntorch_amd as tn
from tntorch_amd import _hip as h

dev = torch.device("cuda")
def timed(fn, reps=2):
    fn(); torch.cuda.synchronize()
    h.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    prof = h.prof_collect(); h.prof_enable(False)
    return el, {k: (round(v["ms"] / reps, 2), v["launches"] // reps) for k, v in prof.items() if v["launches"]}, r

which = sys.argv[1:] or ["gemm", "c3", "c1"]
if "gemm" in which:
    for (rows, n) in ((1 << 20, 1024), (1 << 21, 256)):
        M = torch.randn(1, rows, n, device=dev)
        V = torch.randn(1, n, n, device=dev)
        for big in (1, 0):
            h.set_knob(h.KNOB_GEMM_BIG, big)
            for name, fn, fl in (("gram A^T A", lambda: h.gemm(M, M, transA=True), 2.0 * rows * n * n),
                                 ("rotate M V", lambda: h.gemm(M, V), 2.0 * rows * n * n)):
                fn(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3): fn()
                torch.cuda.synchronize()
                el = (time.perf_counter() - t0) / 3
                print(json.dumps({"gemm": name, "rows": rows, "n": n, "big": big, "ms": round(el * 1e3, 2), "TFLOPs_full": round(fl / el / 1e12, 1)}), flush=True)
        h.set_knob(h.KNOB_GEMM_BIG, 1)
        del M, V
if "c3" in which:
    X = torch.randn(64, 32, 32, 32, 32, 32, device=dev)
    for inner in (2, 1, 3, 0):
        h.set_knob(h.KNOB_BJ_INNER_SWEEPS, inner)
        for alg in ("svd", "eig"):
            el, kinds, r = timed(lambda: tn.Tensor(X, ranks_tt=8, batch=True, algorithm=alg))
            err = float(((r.torch()[0] - X[0]).norm() / X[0].norm()).item())
            print(json.dumps({"c3_share": 64, "alg": alg, "inner_sweeps": inner, "ms": round(el * 1e3, 1), "relerr_item0": err, "kinds": kinds}), flush=True)
    h.set_knob(h.KNOB_BJ_INNER_SWEEPS, 2)
    del X
if "c1" in which:
    torch.cuda.empty_cache()
    shape = [32] + [64] * 4 if "small" in which else [48] + [64] * 5
    X = torch.randn(shape, device=dev)
    for inner in (2, 0):
        h.set_knob(h.KNOB_BJ_INNER_SWEEPS, inner)
        el, kinds, r = timed(lambda: tn.Tensor(X, ranks_tt=16), reps=1)
        print(json.dumps({"c1_shape": shape, "inner_sweeps": inner, "ms": round(el * 1e3, 1), "ranks": r.ranks_tt.tolist(), "kinds": kinds}), flush=True)
    h.set_knob(h.KNOB_BJ_INNER_SWEEPS, 2)
