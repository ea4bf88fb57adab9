"""BASELINE.json's non-headline configs (C1 .. C4) measured through the public API on ONE GPU, for `bench.py`:

    python bench.py                      # default run: the metric line carries `configs` = one entry per config below
    python bench.py --config c2          # (c1 | c2 | c3 | c4) one config alone, one JSON line

Every entry has: `workload`, `ms` (median of the timed repetitions), the ALGORITHMIC flops / bytes of SURVEY.md section
8(d) for the shape that was run, `roofline` {bound, achieved, peak, unit, frac} against the roof that binds it,
`cpu_baseline` (the oracle = the reference's operator sequence on the host cores, timed on the config itself where the
reference's algorithm can run it, else on the largest feasible proxy -- labelled) and `oracle_check` (a post-timing
comparison with the oracle where the oracle can run).  Peaks: HBM 8 TB/s, fp32 MFMA 157.3 TF, fp64 MFMA 78.6 TF
(vendor figure; `tools/microbench.hip d` measures it on the box, see profiles/).

C1  dense 64^6 fp32 -> ranks_tt = 16.  64^6 is 256 GiB and does not fit 288 GB with its carry: the largest member of
    the family that does (48 x 64^5 = 192 GiB with ~250 GiB free) is run and named in `workload`.
C2  round_tt(eps = 1e-4) of a rank-64 TT, 10 cores x mode 128, fp64: ONE tensor (eps mode: data-dependent ranks) and a
    resident batch of 256 (batch mode ignores eps, tensor.py:2036-2037: the same rounding with rmax = 32).
C3  512 dense 32^5 fp32 tensors -> rmax 8, the per-GPU share of the 8-GPU config (64 tensors = 8.6 GB) and the whole
    config on one GPU (8 shares back to back).
C4  CP-ALS R = 32 on a dense 256^4 fp32 tensor (17.2 GB resident): HOSVD init and one ALS sweep.
"""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_GBS = 8000.0
MFMA_F32_TF = 157.3
MFMA_F64_TF = 78.6


def _timeit(fn, reps, warmup=1):
    for _ in range(warmup):
        r = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts, r


def _kinds(fn):
    """Per-kind kernel time of ONE extra (untimed) repetition: HIP events around every launch inside the library."""
    from tntorch_amd import _hip

    torch.cuda.synchronize()
    _hip.prof_enable(True)
    fn()
    torch.cuda.synchronize()
    prof = _hip.prof_collect()
    _hip.prof_enable(False)
    return {k: {"ms": round(v["ms"], 3), "launches": int(v["launches"])} for k, v in prof.items() if v["launches"]}


def _cpu_time(fn, threads=(8,), reps=2, budget_s=12.0):
    """Median seconds of `fn` on the host cores, best over a few MKL thread counts, bounded by `budget_s`."""
    saved = torch.get_num_threads()
    ncpu = os.cpu_count() or 8
    best, used, t_start = None, None, time.perf_counter()
    for nt in [t for t in threads if t <= ncpu] or [min(8, ncpu)]:
        torch.set_num_threads(nt)
        ts = []
        for _ in range(reps + 1):  # first run = warm-up (MKL init) unless the budget is already gone
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget_s:
                break
        sec = sorted(ts[1:] or ts)[len(ts[1:] or ts) // 2]
        if best is None or sec < best:
            best, used = sec, nt
        if time.perf_counter() - t_start > budget_s:
            break
    torch.set_num_threads(saved)
    return best, used


def _roof(bound, flops, byts, sec):
    if bound == "hbm":
        ach, peak, unit = byts / sec / 1e9, HBM_GBS, "GB/s"
    elif bound == "mfma_f64":
        ach, peak, unit = flops / sec / 1e12, MFMA_F64_TF, "TFLOP/s"
    else:
        ach, peak, unit = flops / sec / 1e12, MFMA_F32_TF, "TFLOP/s"
    return {"bound": "hbm" if bound == "hbm" else "mfma", "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
            "traffic": None, "algorithmic_flops": flops, "algorithmic_bytes": byts,
            "hbm_frac": byts / sec / 1e9 / HBM_GBS}


def _paths(fn):
    """Which truncation path every big bond (both dimensions above 64) of ONE extra run took (`_hipops.PATH_TRACE`):
    'subspace' = certified range finder + fused small truncation, 'topk_one_pass' = selected eigenpairs of the Gram matrix
    (flat kept spectrum), 'full_one_pass' / 'full_two_pass' = the n x n eigen-decomposition(s)."""
    from tntorch_amd import _hipops

    _hipops.PATH_TRACE = []
    try:
        fn()
        torch.cuda.synchronize()
        tr = list(_hipops.PATH_TRACE)
    finally:
        _hipops.PATH_TRACE = None
    return [{"path": p, "m": m, "n": n, "cap": r} for p, m, n, r in tr]


def _lowrank_plus_noise(shape, r, dev, gen, batch=None, noise=1e-3, split=None, out=None):
    """SURVEY 8d's primary dense input: a TT-rank-r tensor of unit RMS + noise * randn, built on the device without a second
    tensor of its size: the product of the (modes < split) and (modes >= split) halves of a random rank-r train, written in
    row chunks, the noise added chunk by chunk.  `batch`: leading batch dimension (independent items)."""
    N = len(shape)
    split = split if split is not None else (N + 1) // 2
    lead = () if batch is None else (batch,)
    rr = [1] + [r] * (N - 1) + [1]
    cores = [torch.randn(lead + (rr[k], shape[k], rr[k + 1]), generator=gen, device=dev) for k in range(N)]

    def chain(cs):
        acc = cs[0].reshape(lead + (-1, cs[0].shape[-1]))
        for c in cs[1:]:
            acc = (acc @ c.reshape(lead + (c.shape[-3], -1))).reshape(lead + (-1, c.shape[-1]))
        return acc

    L = chain(cores[:split])                                   # [.., rows, r]  (rows = prod(shape[:split]))
    Rt = cores[split].reshape(lead + (r, -1))
    for c in cores[split + 1:]:
        Rt = (Rt.reshape(lead + (-1, c.shape[-3])) @ c.reshape(lead + (c.shape[-3], -1))).reshape(lead + (r, -1))
    # unit RMS: ||L Rt||^2 = trace((L^T L)(Rt Rt^T))
    nrm2 = ((L.transpose(-1, -2) @ L) * (Rt @ Rt.transpose(-1, -2))).sum(dim=(-1, -2))
    L = L * (math.sqrt(math.prod(shape)) / nrm2.sqrt()).reshape(lead + (1, 1))
    rows, cols = L.shape[-2], Rt.shape[-1]
    X = torch.empty(lead + (rows, cols), device=dev, dtype=torch.float32) if out is None else out.view(lead + (rows, cols))
    step = max(1, (1 << 28) // cols)
    for r0 in range(0, rows, step):
        blk = X[..., r0:r0 + step, :]
        torch.matmul(L[..., r0:r0 + step, :], Rt, out=blk) if batch is None else blk.copy_(L[..., r0:r0 + step, :] @ Rt)
        blk.add_(torch.randn(blk.shape, generator=gen, device=dev), alpha=noise)
    return X.reshape(lead + tuple(shape))


def _dense_tt_check(X, t, expect_err=None):
    """Check of a TIMED dense -> TT result at config scale, on the device, slice by slice along the first mode (no second tensor
    of the input's size): T = the projection of X onto the right-orthonormal cores 1.. iff <X, T> = ||T||^2 and the cores
    are right-orthonormal; then ||X - T||^2 = ||X||^2 - ||T||^2.  Reports the relative approximation error, the projection
    defect |<X,T> - ||T||^2| / ||T||^2, the energy identity and the orthonormality defect; `expect_err`: the error a
    low-rank + noise input must come out at."""
    from tntorch_amd import _hip, _hipops

    cores = [c if c.dim() == 4 else c[None] for c in t.cores]          # [1, r, I, r']
    xx = tt = xt = d2 = 0.0
    # pieces of <= 1 GiB: one index of the first mode x a group of indices of the second (the check has to fit next to an
    # input that fills the device)
    per = X[0].numel()
    grp = X.shape[1] if per <= (1 << 28) else max(1, X.shape[1] // (per >> 28))
    for i in range(X.shape[0]):
      for j0 in range(0, X.shape[1], grp):
        xi = X[i, j0:j0 + grp]
        Ti = _hipops.decompress([cores[0][:, :, i:i + 1, :].contiguous(), cores[1][:, :, j0:j0 + grp, :].contiguous()]
                                + cores[2:]).reshape(xi.shape)
        xx += float(_hip.norm(xi.reshape(1, -1))[0].item()) ** 2
        tt += float(_hip.norm(Ti.reshape(1, -1))[0].item()) ** 2
        # <x, t> as 1024 partial dot products (batched 1 x n x 1 GEMMs), summed in double
        xt += float(_hip.gemm(xi.reshape(1024, 1, -1), Ti.reshape(1024, -1, 1)).double().sum().item())
        d2 += float(_hipops.dense_dist(xi, Ti).item()) ** 2
        del Ti
    orth = 0.0
    for c in cores[1:]:
        Rm = c.reshape(1, c.shape[1], -1)
        G = _hip.gemm(Rm, Rm, transB=True)[0]
        orth = max(orth, float((G - torch.eye(G.shape[0], device=G.device, dtype=G.dtype)).abs().max().item()))
    err = math.sqrt(d2 / xx)
    out = {"approx_err": err, "kept_energy_fraction": tt / xx, "projection_defect": abs(xt - tt) / max(tt, 1e-300),
           "energy_identity_defect": abs(d2 - (xx - tt)) / xx, "right_orthonormality_defect": orth,
           "bounds": {"projection_defect": 1e-3, "energy_identity_defect": 1e-4, "right_orthonormality_defect": 5e-5}}
    ok = out["projection_defect"] <= 1e-3 and out["energy_identity_defect"] <= 1e-4 and orth <= 5e-5
    if expect_err is not None:
        out["expected_err"] = expect_err
        ok = ok and abs(err - expect_err) <= 2e-5
    out["ok"] = bool(ok)
    return out


def _tt_rel_err(a, b):
    import oracle
    a = [c.double() for c in a]
    b = [c.double() for c in b]
    aa, bb, ab = oracle.tt_dot(a, a), oracle.tt_dot(b, b), oracle.tt_dot(a, b)
    return math.sqrt(max((aa + bb - 2 * ab).item(), 0.0) / bb.item())


# ---------------------------------------------------------------------------------------------------------------- C2
def c2(tn, dev, algorithm="svd", cpu=True):
    import oracle

    N, I, r = 10, 128, 32
    flop, byts = 2.31e9, 1.43e8  # SURVEY 8d, per tensor
    torch.manual_seed(0)
    g = oracle.tt_randn([I] * N, r, dtype=torch.float64)
    inp = oracle.tt_add(g, g)
    t_in = tn.Tensor([c.to(dev) for c in inp])
    sec1, all1, out1 = _timeit(lambda: tn.round_tt(t_in, eps=1e-4, algorithm=algorithm), reps=5, warmup=2)
    res = {
        "workload": "round_tt(eps=1e-4) of a rank-64 TT (g+g, g randn rank 32), 10 cores x mode 128, fp64",
        "dtype": "f64", "algorithm": algorithm,
        "single_tensor": {"ms": sec1 * 1e3, "ms_all": [round(x * 1e3, 3) for x in all1], "ranks": out1.ranks_tt.tolist(),
                          "cores_per_s": N / sec1, "roofline": _roof("mfma_f64", flop, byts, sec1)},
    }
    # oracle check (the reference's 'eig' path finishes in 0.2 s; its default 'svd' takes 10 s on this input)
    ref = oracle.round_tt([c.clone() for c in inp], eps=1e-4, algorithm="eig")
    ours = [c.cpu() for c in out1.cores]
    err = _tt_rel_err(ours, ref)
    res["oracle_check"] = {"ranks_identical": oracle.tt_ranks(ours) == oracle.tt_ranks(ref), "rel_err_vs_oracle_eig": err,
                           "bound": 1e-7, "ok": bool(oracle.tt_ranks(ours) == oracle.tt_ranks(ref) and err <= 1e-7)}
    # resident batch
    B = 256
    gen = torch.Generator(device=dev).manual_seed(5)
    rr = [1] + [r] * (N - 1) + [1]
    cores = []
    for k in range(N):
        gk = torch.randn((B, rr[k], I, rr[k + 1]), generator=gen, device=dev, dtype=torch.float64)
        if k == 0:
            c = torch.cat([gk, gk], dim=-1)
        elif k == N - 1:
            c = torch.cat([gk, gk], dim=-3)
        else:
            z = torch.zeros_like(gk)
            c = torch.cat([torch.cat([gk, z], dim=-1), torch.cat([z, gk], dim=-1)], dim=-3)
        cores.append(c.contiguous())
    tb = tn.Tensor(cores, batch=True)
    secB, allB, outB = _timeit(lambda: tn.round_tt(tb, rmax=r, algorithm=algorithm), reps=3, warmup=1)
    res["batch_256"] = {"ms": secB * 1e3, "ms_all": [round(x * 1e3, 2) for x in allB], "tensors": B,
                        "ranks": outB.ranks_tt.tolist(), "cores_per_s": B * N / secB,
                        "roofline": _roof("mfma_f64", flop * B, byts * B, secB),
                        "kernel_ms": _kinds(lambda: tn.round_tt(tb, rmax=r, algorithm=algorithm))}
    del tb, outB, cores
    # (`ms` and `roofline` describe the SAME measurement -- the resident batch of 256; the single train of BASELINE's wording
    # is latency-bound, 0.008 of the fp64 MFMA peak, and stands beside it as `single_tensor`)
    res["ms"] = secB * 1e3
    res["ms_is"] = "batch_256"
    res["roofline"] = res["batch_256"]["roofline"]
    if cpu:
        sec, nt = _cpu_time(lambda: oracle.round_tt([c.clone() for c in inp], eps=1e-4, algorithm="eig"), threads=(8, 16), reps=2)
        res["cpu_baseline"] = {"value": N / sec, "unit": "cores/s", "cores": nt, "kind": "port",
                               "sample": "oracle.round_tt(eps=1e-4, algorithm='eig') of ONE C2 tensor (the config itself; the "
                                         "reference's default 'svd' needs 10.8 s per tensor on it, SURVEY section 6)",
                               "sec_per_tensor": sec}
        res["single_tensor"]["speedup_vs_cpu"] = sec / sec1
        res["batch_256"]["speedup_vs_cpu"] = sec / (secB / B)
    return res


# ---------------------------------------------------------------------------------------------------------------- C3
def c3(tn, dev, algorithm="svd", cpu=True, variant="randn"):
    import oracle

    share, total = 64, 512
    shape = [32] * 5
    flop, byts = 7.71e9, 3.72e8  # SURVEY 8d, per tensor
    gen = torch.Generator(device=dev).manual_seed(99)
    if variant == "randn":
        X = torch.randn([share] + shape, generator=gen, device=dev, dtype=torch.float32)
    else:  # "plus low-rank+noise variant" (TT rank 8 = the cap, unit RMS, + 1e-3 randn)
        X = _lowrank_plus_noise(shape, 8, dev, gen, batch=share, split=3)
    sec, allt, out = _timeit(lambda: tn.Tensor(X, ranks_tt=8, batch=True, algorithm=algorithm), reps=3, warmup=1)
    assert out.ranks_tt.tolist() == [1, 8, 8, 8, 8, 1]

    # the whole config on ONE GPU: 8 shares back to back (the same resident share stands in for every one: synthetic data).
    # One resident batch of all 512 tensors was measured and is NOT faster (582 vs 496 ms: the block-Jacobi launches of the
    # n = 256 bonds are throughput-bound already at 64 items).
    def whole():
        o = None
        for _ in range(total // share):
            o = tn.Tensor(X, ranks_tt=8, batch=True, algorithm=algorithm)
        return o

    secW, allW, _ = _timeit(whole, reps=1, warmup=0)
    how = "8 shares of 64 back to back"
    out = tn.Tensor(X, ranks_tt=8, batch=True, algorithm=algorithm)  # (for the oracle check below)
    kinds = _kinds(lambda: tn.Tensor(X, ranks_tt=8, batch=True, algorithm=algorithm))
    res = {
        "kernel_ms": kinds,
        "big_bond_paths": _paths(lambda: tn.Tensor(X, ranks_tt=8, batch=True, algorithm=algorithm)),
        "input": variant if variant == "randn" else "TT rank 8 of unit RMS + 1e-3 randn",
        "workload": "TT-SVD of dense 32^5 fp32 tensors, rmax 8: the 64-tensor per-GPU share of the 8-GPU config (8.6 GB resident)",
        "dtype": "f32", "algorithm": algorithm,
        "ms": sec * 1e3, "ms_all": [round(x * 1e3, 2) for x in allt], "tensors": share, "tensors_per_s": share / sec,
        "roofline": _roof("hbm", flop * share, byts * share, sec),
        "whole_config_on_one_gpu": {"tensors": total, "how": how, "ms": secW * 1e3, "tensors_per_s": total / secW,
                                    "roofline": _roof("hbm", flop * total, byts * total, secW)},
    }
    x0 = X[0].cpu()
    rec0 = out.torch()[0].cpu()
    ref = oracle.dense_to_tt(x0, 8, algorithm=algorithm)
    e_o = ((rec0.double() - x0.double()).norm() / x0.double().norm()).item()
    e_r = ((oracle.tt_to_dense([c.double() for c in ref]) - x0.double()).norm() / x0.double().norm()).item()
    res["oracle_check"] = {"approx_err_ours": e_o, "approx_err_oracle": e_r, "bound_abs_diff": 1e-5,
                           "ok": bool(abs(e_o - e_r) <= 1e-5 and oracle.tt_ranks(ref) == [1, 8, 8, 8, 8, 1])}
    if cpu and variant == "randn":
        secc, nt = _cpu_time(lambda: oracle.dense_to_tt(x0, 8, algorithm="eig"), threads=(8,), reps=1, budget_s=5.0)
        res["cpu_baseline"] = {"value": 1.0 / secc, "unit": "tensors/s", "cores": nt, "kind": "port",
                               "sample": "oracle.dense_to_tt(rmax=8, algorithm='eig') of ONE dense 32^5 tensor of the config "
                                         "(`_full_rank_tt` + `round_tt`, tensor.py:10-104, 401-408)", "sec_per_tensor": secc}
        res["speedup_vs_cpu"] = secc / (sec / share)
    del X, out
    return res


# ---------------------------------------------------------------------------------------------------------------- C1
def _c1_model(shape):
    """SURVEY 8d, dense right-to-left TT-SVD: step with rows = prod(shape[:j]), n = I r: gram 2 rows n^2 + eig 9 n^3 + project
    2 rows n r; bytes 4 (2 rows n + rows r + r n)."""
    flop = byts = gram = 0.0
    r_next = 1
    for j in range(len(shape) - 1, 0, -1):
        rows, n = float(math.prod(shape[:j])), float(shape[j]) * r_next
        r = min(16.0, rows, n)
        flop += 2 * rows * n * n + 9 * min(rows, n) ** 3 + 2 * rows * n * r
        gram += 2 * rows * n * n
        byts += 4 * (2 * rows * n + rows * r + r * n)
        r_next = r
    return flop, byts, gram


def c1(tn, dev, algorithm="svd", cpu=True, variant="randn", shape=None):
    import oracle

    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    cands = [[64] * 6, [48] + [64] * 5, [32] + [64] * 5, [16] + [64] * 5, [64] * 5, [64] * 4]
    nbytes = lambda sh: math.prod(sh) * 4
    # a shape fits next to its carry (1.35 x), or -- round 4 -- alone, with the first carry written IN PLACE over the consumed
    # front of the input (`Tensor.from_dense_consuming`: the input is destroyed, so every repetition regenerates it, untimed)
    fits = lambda sh: nbytes(sh) * 1.35 <= free or nbytes(sh) + (5 << 30) <= free   # (in place: the input + < 4 GiB of carries and workspaces)
    shape = shape or next(sh for sh in cands if fits(sh))
    free_gib = free / 2 ** 30
    consume = not nbytes(shape) * 1.35 <= free

    def make(out=None):
        gen = torch.Generator(device=dev).manual_seed(7)
        if variant == "randn":
            if out is None:
                return torch.randn(shape, generator=gen, device=dev, dtype=torch.float32)
            flat = out.view(-1)
            for c0 in range(0, flat.numel(), 1 << 30):    # (chunks: identical values for a fresh and a refilled buffer)
                flat[c0:c0 + (1 << 30)].normal_(generator=gen)
            return out
        # SURVEY 8d's primary C1 input: low-rank (TT rank 16, unit RMS) + 1e-3 randn
        return _lowrank_plus_noise(shape, 16, dev, gen, out=out).view(shape)

    X = make(torch.empty(shape, device=dev, dtype=torch.float32))

    def run():
        if consume:
            return tn.Tensor.from_dense_consuming(X, 16, algorithm=algorithm)
        return tn.Tensor(X, ranks_tt=16, algorithm=algorithm)

    ts, out = [], None
    for rep in range(3):                                   # one warm-up + two timed repetitions
        if consume and rep > 0:
            out = None
            make(X)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    allt = sorted(ts[1:])
    sec = allt[len(allt) // 2]
    ranks_out = out.ranks_tt.tolist()
    flop, byts, gram = _c1_model(shape)
    name = "x".join(map(str, shape))
    if consume:
        make(X)                                            # the same values again: the check needs the input the result came from
    check = _dense_tt_check(X, out, expect_err=None if variant == "randn" else 1e-3)
    del out

    def again():
        if consume:
            make(X)
        return run()

    res = {
        "input": variant if variant == "randn" else "TT rank 16 of unit RMS + 1e-3 randn",
        "oracle_check": dict(check, what="the TIMED result checked on the device: T is the orthogonal projection of X onto its "
                                        "right-orthonormal cores (<X,T> = ||T||^2, ||X-T||^2 = ||X||^2 - ||T||^2); the choice of the "
                                        "subspace against the oracle: tests/test_gpu_parity.py at 64^4"),
        "in_place_first_carry": bool(consume), "free_GiB_before": round(free_gib, 1),
        "big_bond_paths": _paths(again),
        "kernel_ms": _kinds(again),
        "workload": f"TT-SVD of a dense {name} fp32 tensor ({math.prod(shape) * 4 / 2 ** 30:.0f} GiB resident) to ranks_tt=16: the largest "
                    "member of the C1 family that fits" + (" -- with the first carry written in place over the consumed input "
                                                            "(Tensor.from_dense_consuming)" if consume else ""),
        "dtype": "f32", "algorithm": algorithm, "shape": shape, "ranks": ranks_out,
        "ms": sec * 1e3, "ms_all": [round(x * 1e3, 1) for x in allt],
        # AI = flop / bytes ~ 60 > ridge 19.7: the n = 1024 Gram matrix of the second step binds (MFMA >= 301 ms, HBM >= 121 ms at 64^6)
        "roofline": _roof("mfma", flop, byts, sec),
    }
    del X
    torch.cuda.empty_cache()
    if cpu and variant == "randn":
        torch.manual_seed(3)
        xp = torch.randn(64, 64, 64, 64)
        secc, nt = _cpu_time(lambda: oracle.dense_to_tt(xp, 16, algorithm="eig"), threads=(8,), reps=1, budget_s=5.0)
        per_elem = secc / xp.numel()
        res["cpu_baseline"] = {"value": 1.0 / (per_elem * math.prod(shape)), "unit": "tensors/s (extrapolated per element)", "cores": nt,
                               "kind": "port", "sec_proxy": secc,
                               "sample": "oracle.dense_to_tt(ranks 16, 'eig') of a dense 64^4 fp32 tensor -- the largest C1 proxy the "
                                         "reference's algorithm can run (`_full_rank_tt` materialises eye(64^3) = 256 GiB at 64^6, SURVEY "
                                         "appendix B); extrapolated per element, which flatters the CPU (its QR cost grows faster)"}
        res["speedup_vs_cpu_extrapolated"] = per_elem * math.prod(shape) / sec
    return res


# ---------------------------------------------------------------------------------------------------------------- C4
def c4(tn, dev, cpu=True, I=256):
    import oracle
    from tntorch_amd import _hip as h
    from tntorch_amd import _hipops

    R, N = 32, 4
    gen = torch.Generator(device=dev).manual_seed(0)
    fac = [torch.randn(I, R, generator=gen, device=dev) for _ in range(N)]
    T = fac[0]
    for f in fac[1:-1]:
        T = (T[:, None, :] * f[None, :, :]).reshape(-1, R)
    X = h.gemm(T[None], fac[-1][None], transB=True)[0].reshape([I] * N)  # rank-32 CP ...
    del T
    X.add_(torch.randn(X.shape, generator=gen, device=dev), alpha=0.01 * float(X.std()))  # ... + 1 % noise
    # the init: first call (cold: includes whatever its kernels' first launch on this queue costs the runtime -- scratch sizing,
    # code loading: 80 ms to 1.2 s in round 4's full-line runs, 94 - 145 ms alone) and the median of three warm repetitions, like
    # every other timing of this file
    t_inits = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _hipops.cp_hosvd_init(X, R)
        torch.cuda.synchronize()
        t_inits.append(time.perf_counter() - t0)
    t_init_cold, t_init = t_inits[0], sorted(t_inits[1:])[1]

    def run(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, errs = _hipops.cp_als(X, R, max_iter=iters, tol=-1.0)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, errs

    t1, _ = run(1)
    t6, errs = run(6)
    sweep = (t6 - t1) / 5
    elems = float(I) ** N
    own_bytes = 2 * elems * 4                     # this formulation: X read twice per sweep
    # ... by two GEMMs X x_n A_n (2 I^N R flops each); the Khatri-Rao folds work on I^(N-1) x R intermediates (1 / I of that)
    own_flops = 2 * 2.0 * elems * R * (1.0 + 1.0 / I)
    survey_bytes = 9.66e10 * elems / 256.0 ** 4   # N reads of X + Khatri-Rao matrices + reconstruction, per iteration
    survey_flops = 1.37e12 * elems / 256.0 ** 4
    res = {
        "workload": f"CP-ALS R=32 on a dense {I}^4 fp32 tensor ({elems * 4 / 1e9:.1f} GB resident; rank-32 CP + 1 % noise): one ALS sweep "
                    "(all 4 modes + the error), HOSVD init reported separately",
        "dtype": "f32", "ms": sweep * 1e3, "init_ms": t_init * 1e3, "init_cold_ms": t_init_cold * 1e3, "errors": [round(float(e), 6) for e in errs],
        # the roof that binds THIS formulation: AI = own flops / own bytes = 16 flop/B, below the fp32 ridge (157.3 TF / 8 TB/s =
        # 19.7): HBM.  (Round 4's line divided the REFERENCE formulation's 1.37e12 flops by this path's bytes -- AI 40, "mfma" --
        # a mixed model; the reference's figures now only appear under `survey_model`.)
        "roofline": dict(_roof("hbm", own_flops, own_bytes, sweep), arithmetic_intensity=own_flops / own_bytes,
                         ridge_flop_per_byte=MFMA_F32_TF * 1e12 / (HBM_GBS * 1e9),
                         mfma_frac=own_flops / sweep / 1e12 / MFMA_F32_TF),
        "survey_model": {"bytes_per_iteration": survey_bytes, "flops_per_iteration": survey_flops,
                         "note": "SURVEY 8d prices the REFERENCE's formulation (X read once per mode + Khatri-Rao matrices + a dense "
                                 "reconstruction for the error: >= 15.4 ms at 8 TB/s for 256^4); this path reads X twice per sweep "
                                 "(roofline.algorithmic_bytes), so `frac` is against its own traffic, not the reference's",
                         "reference_formulation_floor_ms": survey_bytes / (HBM_GBS * 1e9) * 1e3,
                         "frac_of_reference_formulation_floor": survey_bytes / (HBM_GBS * 1e9) / sweep},
    }
    del X
    torch.cuda.empty_cache()
    # oracle check + CPU baseline on the largest proxy the oracle finishes in seconds (64^4, two sweeps)
    Ip = 64
    torch.manual_seed(1)
    fp = [torch.randn(Ip, R, dtype=torch.float64) for _ in range(N)]
    Xp = oracle.cp_to_dense(fp)
    Xp = (Xp / Xp.norm() * math.sqrt(Xp.numel()) + 1e-2 * torch.randn(Xp.shape, dtype=torch.float64)).float()
    tp = tn.Tensor(Xp.to(dev), ranks_cp=R, max_iter=2, tol=-1.0)
    t0 = time.perf_counter()
    _, ref_err = oracle.cp_als(Xp.double(), R, max_iter=2, tol=-1.0)   # fp64 yardstick (tests/test_gpu_parity.py explains why)
    t_or = time.perf_counter() - t0
    d = max(abs(float(a) - float(b)) for a, b in zip(tp.cp_errors, ref_err))
    res["oracle_check"] = {"proxy": "64^4 fp32, R = 32, 2 sweeps", "errors_ours": [float(e) for e in tp.cp_errors],
                           "errors_oracle": [float(e) for e in ref_err], "max_abs_diff": d, "bound": 3e-4, "ok": bool(d <= 3e-4)}
    if cpu:
        init = oracle.cp_hosvd_init(Xp, R)
        secc, nt = _cpu_time(lambda: oracle.cp_als(Xp, R, max_iter=1, tol=-1.0, init=init), threads=(8,), reps=1, budget_s=5.0)
        per_elem = secc / Xp.numel()
        res["cpu_baseline"] = {"value": 1.0 / (per_elem * elems), "unit": "sweeps/s (extrapolated per element)", "cores": nt, "kind": "port",
                               "sec_proxy_sweep": secc,
                               "sample": "one sweep of oracle.cp_als (tensor.py:295-381 restated) on a dense 64^4 fp32 tensor, R = 32; the "
                                         "reference at 256^4 needs ~65 GB of host RAM (SURVEY section 6); extrapolated per element"}
        res["speedup_vs_cpu_extrapolated"] = per_elem * elems / sweep
    return res


CONFIGS = {"c2": c2, "c3": c3, "c4": c4, "c1": c1}   # run order: the 192 GiB config last, after everything else was freed


def config_extras(tn, dev, budget_s=170.0, algorithm="svd", cpu=True):
    """One entry per config; a config that fails or would overrun the time budget reports why instead of a number."""
    out, t_start = {}, time.perf_counter()
    for name, fn in CONFIGS.items():
        if time.perf_counter() - t_start > budget_s:
            out[name] = {"skipped": f"time budget of {budget_s:.0f} s for the config extras used up"}
            continue
        t0 = time.perf_counter()
        try:
            kw = {"cpu": cpu} if name == "c4" else {"cpu": cpu, "algorithm": algorithm}
            out[name] = fn(tn, dev, **kw)
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": repr(e)[:400]}
            if name == "c1":  # (a device that cannot hold 64^6 alone after all: the member that fits next to its carry)
                torch.cuda.empty_cache()
                try:
                    out[name] = fn(tn, dev, shape=[32] + [64] * 5, **kw)
                    out[name]["note"] = "64^6 failed (" + repr(e)[:120] + "): ran the largest member that fits next to its carry"
                except Exception as e2:  # noqa: BLE001
                    out[name] = {"error": repr(e2)[:400]}
        out[name]["wall_s"] = round(time.perf_counter() - t0, 1)
        torch.cuda.empty_cache()
        if name in ("c1", "c3") and "error" not in out[name]:
            # SURVEY 8d's other input variant (low rank + 1e-3 noise: the kept spectrum is NOT flat), same shape, same checks
            t0 = time.perf_counter()
            try:
                kw = {"cpu": False, "algorithm": algorithm, "variant": "lowrank"}
                if name == "c1":
                    kw["shape"] = out[name]["shape"]
                v = fn(tn, dev, **kw)
                v["ratio_to_randn"] = v["ms"] / out[name]["ms"]
            except Exception as e:  # noqa: BLE001
                v = {"error": repr(e)[:400]}
            v["wall_s"] = round(time.perf_counter() - t0, 1)
            out[name]["lowrank_variant"] = v
            torch.cuda.empty_cache()
    return out


def main(args):
    import tntorch_amd as tn
    from tntorch_amd import _hip

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    _hip.lib()
    cpu = not getattr(args, "no_cpu_baseline", False)
    kw = {"cpu": cpu} if args.config == "c4" else {"cpu": cpu, "algorithm": args.algorithm}
    if args.config in ("c1", "c3") and getattr(args, "variant", "randn") != "randn":
        kw["variant"] = args.variant
    res = CONFIGS[args.config](tn, dev, **kw)
    line = {"metric": f"BASELINE config {args.config.upper()}: " + res["workload"], "value": res["ms"], "unit": "ms",
            "n_gpus": 1, "steps": None, "warmup": None, "ms_per_step": res["ms"], "higher_is_better": False, "scaling": "weak",
            "vs_baseline": None, "dtype": res.get("dtype", "f32"), "data": "synthetic", "config": {"workload": res["workload"]},
            "roofline": res["roofline"]}
    if "cpu_baseline" in res:
        line["cpu_baseline"] = res["cpu_baseline"]
    line["detail"] = res
    print(json.dumps(line))
