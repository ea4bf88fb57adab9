#!/usr/bin/env python3
"""Generate the golden vectors under ``tests/golden/`` from the UNMODIFIED reference.

TEST INFRASTRUCTURE.  Run in the build container only (the reference lives at
/root/reference and does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python3 oracle/gen_golden.py

Every case seeds torch, builds inputs with the reference's own constructors,
runs the reference's hot path (``tn.round_tt`` / ``tn.Tensor(..., ranks_tt=)`` /
``tn.truncated_svd`` / ``orthogonalize``) and stores inputs + outputs as
``.npz``.  ``tests/test_oracle_golden.py`` replays them against ``oracle/``;
the ``-m gpu`` parity tests replay them against the HIP path.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import tntorch as tn  # noqa: E402  (the reference itself)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
META = {"generator": "oracle/gen_golden.py", "reference": "rballester/tntorch v1.1.1", "torch": torch.__version__, "cases": {}}


def npl(cores):
    return {f"{i}": c.detach().cpu().numpy() for i, c in enumerate(cores)}


def save(name, meta, **groups):
    flat = {}
    for g, d in groups.items():
        if isinstance(d, dict):
            for k, v in d.items():
                flat[f"{g}/{k}"] = v
        else:
            flat[g] = np.asarray(d)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **flat)
    META["cases"][name] = meta


def case_round_eps_f64():
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(1)
    g = tn.rand([6] * 8, ranks_tt=4)
    t = g + g
    outs = {}
    for alg in ("svd", "eig"):
        r = tn.round_tt(t, eps=1e-8, algorithm=alg)
        outs[alg] = npl(r.cores)
    save(
        "round_eps_f64",
        {"what": "t=g+g, g=tn.rand([6]*8, ranks_tt=4), tn.round_tt(t, eps=1e-8)", "ref": "tensor.py:2008-2083", "dtype": "float64", "eps": 1e-8},
        inp=npl(t.cores), svd=outs["svd"], eig=outs["eig"],
    )


def case_round_rmax_f32():
    torch.set_default_dtype(torch.float32)
    torch.manual_seed(2)
    g = tn.randn([8] * 5, ranks_tt=6)
    outs = {}
    for alg in ("svd", "eig"):
        r = tn.round_tt(g, rmax=3, algorithm=alg)
        outs[alg] = npl(r.cores)
    save(
        "round_rmax_f32",
        {"what": "g=tn.randn([8]*5, ranks_tt=6), tn.round_tt(g, rmax=3)", "ref": "tensor.py:2008-2083", "dtype": "float32", "rmax": 3},
        inp=npl(g.cores), svd=outs["svd"], eig=outs["eig"],
    )


def case_round_batch_f64():
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(3)
    g = tn.randn([3, 5, 5, 5, 5], ranks_tt=5, batch=True)
    outs = {}
    for alg in ("svd", "eig"):
        r = tn.round_tt(g, rmax=2, algorithm=alg)
        outs[alg] = npl(r.cores)
    save(
        "round_batch_f64",
        {"what": "g=tn.randn([3,5,5,5,5], ranks_tt=5, batch=True), tn.round_tt(g, rmax=2)", "ref": "tensor.py:2036-2037,2065-2076", "dtype": "float64", "rmax": 2},
        inp=npl(g.cores), svd=outs["svd"], eig=outs["eig"],
    )


def case_dense_f64():
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(4)
    low = tn.randn([8, 7, 6, 9, 5], ranks_tt=4).torch()
    X = low / low.std() + 1e-3 * torch.randn(8, 7, 6, 9, 5)
    outs = {}
    for alg in ("svd", "eig"):
        outs[alg] = npl(tn.Tensor(X, ranks_tt=4, algorithm=alg).cores)
    save(
        "dense_f64",
        {"what": "X=lowrank(8x7x6x9x5, r=4, unit std)+1e-3*randn; tn.Tensor(X, ranks_tt=4)", "ref": "tensor.py:10-104,401-408", "dtype": "float64", "ranks_tt": 4},
        X=X.numpy(), svd=outs["svd"], eig=outs["eig"],
    )


def case_dense_batch_f32():
    torch.set_default_dtype(torch.float32)
    torch.manual_seed(5)
    X = torch.rand(3, 5, 5, 5, 5)
    outs = {}
    for alg in ("svd", "eig"):
        outs[alg] = npl(tn.Tensor(X, ranks_tt=3, batch=True, algorithm=alg).cores)
    save(
        "dense_batch_f32",
        {"what": "X=torch.rand(3,5,5,5,5); tn.Tensor(X, ranks_tt=3, batch=True) (tests/test_tensor.py:28-49)", "ref": "tensor.py:10-104,401-408", "dtype": "float32", "ranks_tt": 3},
        X=X.numpy(), svd=outs["svd"], eig=outs["eig"],
    )


def case_c0():
    torch.set_default_dtype(torch.float32)
    torch.manual_seed(0)
    X = torch.randn(16, 16, 16, 16)
    outs = {}
    for alg in ("svd", "eig"):
        t = tn.Tensor(X)
        t.round_tt(rmax=4, algorithm=alg)
        outs[alg] = npl(t.cores)
    save(
        "c0_16x4_rmax4_f32",
        {
            "what": "BASELINE config C0: torch.manual_seed(0); X=torch.randn(16,16,16,16); t=tn.Tensor(X); t.round_tt(rmax=4)",
            "ref": "tensor.py:10-104,2008-2083",
            "dtype": "float32",
            "input": "regenerated from the seed at test time; checksum below",
            "x_sum": float(X.double().sum()),
            "x_sumsq": float((X.double() ** 2).sum()),
        },
        svd=outs["svd"], eig=outs["eig"],
    )


def case_truncated_svd():
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(6)
    groups = {}
    meta = {"what": "tn.truncated_svd variants (tests/test_round.py:21-38 + rectangular/eps/rmax/left_ortho)", "ref": "round.py:52-187", "dtype": "float64", "calls": []}
    Mb = torch.rand(2, 32, 32)
    groups["Mb"] = Mb.numpy()
    for alg in ("svd", "eig"):
        u, v = tn.truncated_svd(Mb, batch=True, algorithm=alg)
        groups[f"batch_{alg}_left"] = u.numpy()
        groups[f"batch_{alg}_right"] = v.numpy()
    mats = {"wide": torch.randn(6, 20), "tall": torch.randn(20, 6)}
    # give them a decaying spectrum so eps truncation is well defined
    for k, M in mats.items():
        U, s, Vh = torch.linalg.svd(M, full_matrices=False)
        s = torch.tensor([2.0 ** (-j) for j in range(len(s))])
        mats[k] = (U * s) @ Vh
        groups[f"M_{k}"] = mats[k].numpy()
    idx = 0
    for k, M in mats.items():
        for alg in ("svd", "eig"):
            for lo in (True, False):
                for kw in ({"eps": 0.05}, {"rmax": 3}, {"delta": 0.2, "rmax": 5}):
                    u, v = tn.truncated_svd(M, left_ortho=lo, algorithm=alg, **kw)
                    groups[f"call{idx}_left"] = u.numpy()
                    groups[f"call{idx}_right"] = v.numpy()
                    meta["calls"].append({"i": idx, "M": k, "algorithm": alg, "left_ortho": lo, **kw})
                    idx += 1
    save("truncated_svd_f64", meta, **groups)


def case_orthogonalize():
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(7)
    g = tn.rand([4, 5, 6, 3, 4], ranks_tt=3)
    a = g.clone(); a.left_orthogonalize(0)
    b = g.clone(); b.right_orthogonalize(4)
    c = g.clone(); c.orthogonalize(2)
    d = g.clone(); d.orthogonalize(4)
    save(
        "orthogonalize_f64",
        {"what": "g=tn.rand([4,5,6,3,4], ranks_tt=3); left_orthogonalize(0) / right_orthogonalize(4) / orthogonalize(2) / orthogonalize(4)", "ref": "tensor.py:1800-1909", "dtype": "float64"},
        inp=npl(g.cores), left0=npl(a.cores), right4=npl(b.cores), orth2=npl(c.cores), orth4=npl(d.cores),
    )


def _tucker_input(dtype, batch=None, seed=11):
    """TT whose modes are Tucker-compressible: cores with mode sizes S absorbed with random factors I x S."""
    torch.manual_seed(seed)
    S, I, r = [6, 5, 7, 6], [12, 10, 14, 11], 3
    if batch is None:
        g = tn.rand(S, ranks_tt=r)
        cores = [c.to(dtype) for c in g.cores]
        Us = [torch.randn(i, s_, dtype=dtype) for i, s_ in zip(I, S)]
        absorbed = [torch.einsum("iak,ja->ijk", c, U) for c, U in zip(cores, Us)]
    else:
        ranks = [1, r, r, r, 1]
        cores = [torch.rand(batch, ranks[k], S[k], ranks[k + 1], dtype=dtype) for k in range(4)]
        Us = [torch.randn(batch, i, s_, dtype=dtype) for i, s_ in zip(I, S)]
        absorbed = [torch.einsum("biak,bja->bijk", c, U) for c, U in zip(cores, Us)]
    return absorbed


def case_round_tucker():
    torch.set_default_dtype(torch.float64)
    inp = _tucker_input(torch.float64)
    groups = {"inp": npl(inp)}
    for alg in ("svd", "eig"):
        t = tn.Tensor([c.clone() for c in inp])
        t.round_tucker(eps=1e-8, algorithm=alg)
        groups[f"{alg}_cores"] = npl(t.cores)
        groups[f"{alg}_Us"] = npl(t.Us)
    save("round_tucker_eps_f64",
         {"what": "TT 12x10x14x11 (ranks 3) with Tucker ranks 6,5,7,6; t.round_tucker(eps=1e-8)", "ref": "tensor.py:1911-2006",
          "dtype": "float64", "eps": 1e-8}, **groups)

    torch.set_default_dtype(torch.float32)
    inp = _tucker_input(torch.float32, seed=12)
    groups = {"inp": npl(inp)}
    for alg in ("svd", "eig"):
        t = tn.Tensor([c.clone() for c in inp])
        t.round_tucker(rmax=3, algorithm=alg)
        groups[f"{alg}_cores"] = npl(t.cores)
        groups[f"{alg}_Us"] = npl(t.Us)
    save("round_tucker_rmax_f32",
         {"what": "same structure, float32; t.round_tucker(rmax=3)", "ref": "tensor.py:1911-2006", "dtype": "float32", "rmax": 3}, **groups)

    # batch mode: the reference's truncated_svd crashes for tall factors with left_ortho=True (round.py:176,
    # torch.diag on a [B, r] tensor), i.e. whenever I_k > R_k * R_{k+1}; only small modes work.
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(13)
    ranks = [1, 4, 4, 4, 1]
    I = [3, 4, 3, 3]
    inp = [torch.rand(3, ranks[k], I[k], ranks[k + 1]) for k in range(4)]
    groups = {"inp": npl(inp)}
    for alg in ("svd", "eig"):
        t = tn.Tensor([c.clone() for c in inp], batch=True)
        t.round_tucker(rmax=2, algorithm=alg)
        groups[f"{alg}_cores"] = npl(t.cores)
        groups[f"{alg}_Us"] = npl(t.Us)
    save("round_tucker_batch_f64",
         {"what": "batch of 3 TTs 3x4x3x3 rank 4; t.round_tucker(rmax=2) (eps ignored in batch mode)", "ref": "tensor.py:1911-2006",
          "dtype": "float64", "rmax": 2, "batch": True}, **groups)


def case_ctor_tucker():
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(14)
    low = tn.rand([12, 10, 9, 11], ranks_tt=3, ranks_tucker=4).torch()
    X = low / low.norm() + 1e-3 * torch.randn(12, 10, 9, 11) / np.sqrt(low.numel())
    groups = {"inp": X.numpy()}
    for alg in ("svd", "eig"):
        t = tn.Tensor(X, ranks_tucker=4, ranks_tt=3, algorithm=alg)
        groups[f"{alg}_cores"] = npl(t.cores)
        groups[f"{alg}_Us"] = npl(t.Us)
    save("ctor_tucker_f64",
         {"what": "tn.Tensor(X, ranks_tucker=4, ranks_tt=3), X = TT-Tucker(3; 4) + 1e-3 noise, 12x10x9x11", "ref": "tensor.py:401-408",
          "dtype": "float64"}, **groups)


def case_round_general():
    torch.set_default_dtype(torch.float64)
    inp = _tucker_input(torch.float64, seed=15)
    t0 = tn.Tensor([c.clone() for c in inp])
    t0 = t0 + t0  # redundant TT ranks as well
    groups = {"inp": npl(t0.cores)}
    for alg in ("svd", "eig"):
        t = tn.round(t0, eps=1e-6, algorithm=alg)
        groups[f"{alg}_cores"] = npl(t.cores)
        groups[f"{alg}_Us"] = npl([U for U in t.Us])
    save("round_general_f64",
         {"what": "t0 = t+t (TT ranks 6, Tucker ranks 6,5,7,6 inside 12x10x14x11); tn.round(t0, eps=1e-6)", "ref": "tensor.py:2085-2098",
          "dtype": "float64", "eps": 1e-6}, **groups)


def case_cp_als():
    """tn.Tensor(X, ranks_cp=R) -- tensor.py:210-400 (config C4's algorithm at a size that fits a fixture)."""
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(16)
    fac = [torch.randn(i, 3) for i in (12, 10, 9, 11)]
    low = torch.einsum("ar,br,cr,dr->abcd", *fac)
    X = low / low.norm() + 1e-2 * torch.randn(12, 10, 9, 11) / np.sqrt(low.numel())
    groups = {"inp": X.numpy()}
    meta = {"what": "tn.Tensor(X, ranks_cp=R, max_iter=K), X = rank-3 CP + 1e-2 noise, 12x10x9x11", "ref": "tensor.py:210-400",
            "dtype": "float64", "runs": {}}
    for name, R, K in (("r3_it1", 3, 1), ("r3_it25", 3, 25), ("r5_it4", 5, 4)):
        t = tn.Tensor(X, ranks_cp=R, max_iter=K)
        groups[name] = npl(t.cores)
        meta["runs"][name] = {"R": R, "max_iter": K, "relerr": tn.relative_error(X, t).item()}
    torch.manual_seed(17)
    Y = torch.randn(8, 7, 6, dtype=torch.float32)
    t = tn.Tensor(Y, ranks_cp=4, max_iter=6)
    groups["f32_inp"] = Y.numpy()
    groups["f32_r4_it6"] = npl(t.cores)
    meta["runs"]["f32_r4_it6"] = {"R": 4, "max_iter": 6, "relerr": tn.relative_error(Y, t).item()}
    save("cp_als", meta, **groups)


def case_cp_variants():
    """Batched CP-ALS and CP on a Tucker core (tensor.py:214-300, the batch / ranks_tucker branches)."""
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(19)
    Xb = torch.randn(3, 8, 7, 6)
    tb = tn.Tensor(Xb, ranks_cp=4, batch=True, max_iter=6, tol=-1.0)
    groups = {"batch_inp": Xb.numpy(), "batch_r4_it6": npl(tb.cores), "batch_dense": tb.torch().numpy()}
    fac = [torch.randn(i, 3) for i in (9, 8, 7)]
    low = torch.einsum("ar,br,cr->abc", *fac)
    X = low / low.norm() + 1e-2 * torch.randn(9, 8, 7) / np.sqrt(low.numel())
    torch.manual_seed(21)
    tt = tn.Tensor(X, ranks_cp=3, ranks_tucker=4, max_iter=5, tol=-1.0)
    torch.manual_seed(21)  # the ALS start the reference drew: randn(S_n, R) per mode, nothing else consumes the RNG before
    init = [torch.randn(4, 3) for _ in range(3)]
    groups.update({"tucker_inp": X.numpy(), "tucker_init": npl(init), "tucker_cores": npl(tt.cores), "tucker_Us": npl(tt.Us),
                   "tucker_dense": tt.torch().numpy()})
    save("cp_variants_f64", {"what": "tn.Tensor(randn(3,8,7,6), ranks_cp=4, batch=True, max_iter=6, tol=-1); "
                             "tn.Tensor(X 9x8x7 rank-3 + noise, ranks_cp=3, ranks_tucker=4, max_iter=5, tol=-1) with its recorded randn start",
                             "ref": "tensor.py:214-400", "dtype": "float64",
                             "relerr_tucker": tn.relative_error(X, tt).item()}, **groups)


def case_producers():
    """TT x TT product and the rounding tree (tensor.py:687-773, tools.py:460-512)."""
    import operator
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(18)
    a = tn.rand([7, 6, 8, 5], ranks_tt=3)
    b = tn.rand([7, 6, 8, 5], ranks_tt=2)
    prod = a * b
    ts = [tn.rand([7, 6, 8, 5], ranks_tt=2) for _ in range(5)]
    red = tn.reduce(ts, operator.add, eps=1e-6)  # (at eps <= 1e-8 the TT-vs-TT error estimate of round() is cancellation noise)
    red3 = tn.reduce(ts, operator.add, rmax=3)
    groups = {"a": npl(a.cores), "b": npl(b.cores), "prod": npl(prod.cores), "red_dense": red.torch().numpy(),
              "red_ranks_tt": red.ranks_tt.numpy(), "red_ranks_tucker": red.ranks_tucker.numpy(), "red3_cores": npl(red3.cores)}
    for i, t in enumerate(ts):
        groups[f"t{i}"] = npl(t.cores)
    assert all(U is None for U in red3.Us)
    save("producers_f64", {"what": "a*b (ranks 3 x 2 -> 6); tn.reduce(5 rank-2 TTs, operator.add, eps=1e-6) and (rmax=3)",
                           "ref": "tensor.py:687-773, 2309-2320; tools.py:460-512", "dtype": "float64"}, **groups)


def case_consumers():
    """tools.shift_mode (tools.py:650-697) and matrix.TTMatrix construction / torch() / trace() (matrix.py:23-175)."""
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(23)
    g = tn.rand([5, 6, 7, 4, 8], ranks_tt=[3, 4, 5, 2])
    groups = {"g": npl(g.cores)}
    specs = [(1, 2, 1e-3), (3, -2, 1e-6), (0, 4, "same"), (4, -4, 1e-2), (2, 1, 0.3)]
    for k, (n, sh, eps) in enumerate(specs):
        t = tn.Tensor([c.clone() for c in g.cores])
        tn.shift_mode(t, n, sh, eps=eps)  # in place (returns None for shift != 0)
        groups[f"shift{k}"] = npl(t.cores)
    m = torch.rand(11 * 3 * 4, 23 * 2 * 3)
    ttm = tn.TTMatrix(m, input_dims=[11, 3, 4], output_dims=[23, 2, 3], ranks=[20, 7])
    sq = torch.rand(6 * 5, 6 * 5)
    tsq = tn.TTMatrix(sq, input_dims=[6, 5], output_dims=[6, 5], ranks=[36])
    groups.update({"m": m.numpy(), "ttm_cores": npl(ttm.cores), "ttm_dense": ttm.torch().numpy(), "sq": sq.numpy(),
                   "tsq_cores": npl(tsq.cores), "tsq_trace": np.asarray(tsq.trace().item())})
    save("consumers_f64", {"what": "tn.shift_mode on a rank-(3,4,5,2) TT for (n, shift, eps) in " + repr(specs)
                           + "; tn.TTMatrix(rand(132, 138), input_dims=[11,3,4], output_dims=[23,2,3], ranks=[20,7]) cores / "
                           "torch(); TTMatrix(rand(30, 30), [6,5], [6,5], ranks=[36]).trace()",
                           "ref": "tools.py:650-697; matrix.py:23-175", "dtype": "float64", "shift_specs": [list(map(str, sp)) for sp in specs]},
         **groups)


def case_known_answers():
    """docs/tutorials/decompositions.ipynb cells 1, 3, 18 (analytic 128^3 function)."""
    torch.set_default_dtype(torch.float64)
    X, Y, Z = np.meshgrid(range(128), range(128), range(128))
    full = torch.Tensor(np.sqrt(np.sqrt(X) * (Y + Z) + Y * Z**2) * (X + np.sin(Y) * np.cos(Z)))
    t3 = tn.Tensor(full, ranks_tt=3)
    e3 = tn.relative_error(full, t3).item()
    out = {"ranks_tt3": t3.ranks_tt.tolist(), "relerr_tt3": e3}
    for alg in ("svd", "eig"):
        t = tn.Tensor(full)
        t.round_tt(eps=1e-5, algorithm=alg)
        out[f"ranks_eps1e-5_{alg}"] = t.ranks_tt.tolist()
        out[f"relerr_eps1e-5_{alg}"] = tn.relative_error(full, t).item()
    ones = tn.ones([32] * 4)
    out["arith_round_ranks"] = tn.round((ones + ones) * (ones - 2)).ranks_tt.tolist()
    te = tn.Tensor(full, eps=1e-5)
    out["eps_ctor_ranks_tt"] = te.ranks_tt.tolist()
    out["eps_ctor_ranks_tucker"] = te.ranks_tucker.tolist()
    out["eps_ctor_relerr"] = tn.relative_error(full, te).item()
    META["known_answers"] = {
        "source": "docs/tutorials/decompositions.ipynb cell 1 (function), cell 3 (ranks_tt=3: ranks [1,3,3,1], rel. error 0.0005), cell 18 (round_tt(eps=1e-5): ranks [1,4,6,1], rel. error 8.3358e-06), cell 14 (tn.Tensor(full, eps=1e-5): TT ranks [1,4,6,1], Tucker ranks 4,5,6, rel. error 8.3402e-06)",
        "measured_with_reference_here": out,
    }


CASES = [case_round_eps_f64, case_round_rmax_f32, case_round_batch_f64, case_dense_f64, case_dense_batch_f32, case_c0,
         case_truncated_svd, case_orthogonalize, case_round_tucker, case_ctor_tucker, case_round_general, case_cp_als, case_cp_variants,
         case_producers, case_consumers, case_known_answers]

if __name__ == "__main__":
    # no arguments: regenerate everything; otherwise only the named cases (the CP-ALS case is reproducible only
    # to ~1e-12: multi-threaded LAPACK in the reference), merging into the existing golden_meta.json
    only = set(sys.argv[1:])
    meta_path = os.path.join(OUT, "golden_meta.json")
    if only and os.path.exists(meta_path):
        with open(meta_path) as f:
            old = json.load(f)
        META["cases"].update(old.get("cases", {}))
        if "known_answers" in old:
            META["known_answers"] = old["known_answers"]
    for fn in CASES:
        if not only or fn.__name__ in only:
            fn()
    with open(meta_path, "w") as f:
        json.dump(META, f, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(OUT)))
