"""`bench.py --config c1|c3`: BASELINE's dense configs through the public API (one JSON line, same format as bench.py).

C3: 512 independent dense 32^5 fp32 tensors -> TT, rmax = 8 (the per-GPU share is 64 tensors; `--gpus 1` runs the
whole config on one GPU in resident sub-batches).  C1: dense 64^k fp32 -> TT rank 16 for the largest k whose tensor
(plus the streaming workspace) fits the GPU -- 64^6 (256 GiB) does not fit 288 GB together with its carry.
"""
import json
import math
import os
import time

import torch


def _c3(args, tn, dev):
    total, sub = 512, 64
    shape = [32] * 5
    gen = torch.Generator(device=dev).manual_seed(99)
    X = torch.randn([sub] + shape, generator=gen, device=dev, dtype=torch.float32)

    def step():
        out = None
        for _ in range(total // sub):  # the same resident sub-batch stands in for every share (synthetic data)
            out = tn.Tensor(X, ranks_tt=8, batch=True, algorithm=args.algorithm)
        return out

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert out.ranks_tt.tolist() == [1, 8, 8, 8, 8, 1]
    flop, byts = 7.71e9, 3.72e8  # SURVEY 8d, per tensor
    tensors = total * args.steps
    return {
        "metric": "TT-SVD of dense 32^5 fp32 tensors to rmax 8 (BASELINE config C3: 512 tensors), tensors/s",
        "value": tensors / el, "unit": "tensors/s", "ms_per_step": el / args.steps * 1e3,
        "config": {"workload": "512 x dense 32^5 fp32 -> TT rmax 8, sub-batches of 64 resident tensors", "algorithm": args.algorithm},
        "gflops": flop * tensors / el / 1e9,
        "roofline": {"bound": "hbm", "achieved": byts * tensors / el / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": byts * tensors / el / 1e9 / 8000.0, "traffic": None},
    }


def _c1(args, tn, dev):
    free, _ = torch.cuda.mem_get_info()
    # the largest C1-class shape that fits: the input + its first carry (1/4 of it) + Gram workspaces.  64^6 (256 GiB)
    # does not; the leading mode is shortened before a whole mode is dropped ([32] + [64]*5 = 128 GiB is resident).
    cands = [[64] * 6, [48] + [64] * 5, [32] + [64] * 5, [16] + [64] * 5, [64] * 5, [64] * 4, [64] * 3]
    shape = next(sh for sh in cands if math.prod(sh) * 4 * 1.35 <= free)
    k = len(shape)
    gen = torch.Generator(device=dev).manual_seed(7)
    X = torch.randn(shape, generator=gen, device=dev, dtype=torch.float32)

    def step():
        return tn.Tensor(X, ranks_tt=16, algorithm=args.algorithm)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # step with rows = 64^j, n = 64 r: gram 2 rows n^2 + project 2 rows n r; bytes 4 (2 rows n + rows r)
    flop = byts = 0.0
    r_next = 1
    for j in range(k - 1, 0, -1):
        rows, n = float(math.prod(shape[:j])), 64.0 * r_next
        r = min(16.0, rows, n)
        flop += 2 * rows * n * n + 9 * min(rows, n) ** 3 + 2 * rows * n * r
        byts += 4 * (2 * rows * n + rows * r + r * n)
        r_next = r
    return {
        "metric": "TT-SVD of a dense " + "x".join(map(str, shape)) + " fp32 tensor to ranks_tt=16 (BASELINE config C1 class; 64^6 = 256 GiB does not fit), s/tensor",
        "value": el / args.steps, "unit": "s", "higher_is_better": False, "ms_per_step": el / args.steps * 1e3,
        "config": {"workload": "dense " + "x".join(map(str, shape)) + f" fp32 ({math.prod(shape) * 4 / 2 ** 30:.1f} GiB) -> TT ranks 16", "algorithm": args.algorithm,
                   "ranks": out.ranks_tt.tolist()},
        "gflops": flop * args.steps / el / 1e9,
        "roofline": {"bound": "hbm", "achieved": byts * args.steps / el / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": byts * args.steps / el / 1e9 / 8000.0, "traffic": None},
    }


def main(args):
    import tntorch_amd as tn
    from tntorch_amd import _hip

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    _hip.lib()
    res = (_c3 if args.config == "c3" else _c1)(args, tn, dev)
    base = {"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
    base.update(res)
    print(json.dumps(base))
