"""One-off randomised stress: the seeded parity tests of tests/test_gpu_parity.py over many more seeds (GPU only).
Found the colliding-completion bug of the blocked QR (exactly dependent column blocks) in round 1."""
import sys, subprocess
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import pytest, types
import test_gpu_parity as t
fails = 0
for seed in range(12, 150):
    try:
        t.test_random_trains_vs_oracle(seed)
    except Exception as e:
        fails += 1; print("TRAIN seed", seed, "FAILED:", str(e)[:300].replace("\n", " "))
for seed in range(6, 60):
    try:
        t.test_random_dense_and_tucker_vs_oracle(seed)
    except Exception as e:
        fails += 1; print("DENSE seed", seed, "FAILED:", str(e)[:300].replace("\n", " "))
print("stress done, failures:", fails)


# ---------------------------------------------------------------------------------------------- more paths
import math
import numpy as np, torch
import oracle
import tntorch_amd as tn
from parity import dense, ranks, rel_diff, to_list


def gpu(cores, batch=False):
    return tn.Tensor([c.cuda() for c in cores], batch=batch)


def stress_fp32_batch(seed):
    rng = np.random.RandomState(5000 + seed)
    N = int(rng.randint(2, 6))
    shape = [int(rng.randint(2, 9)) for _ in range(N)]
    r = int(rng.randint(1, 7))
    B = int(rng.choice([1, 3, 17, 130, 257]))
    torch.manual_seed(seed)
    g = oracle.tt_randn(shape, r, dtype=torch.float32, batch_size=B)
    inp = oracle.tt_add(g, g, batch=True) if seed % 2 else g
    rmax = int(rng.randint(1, 7))
    for alg in ("svd", "eig"):
        t = gpu(inp, batch=True)
        t.round_tt(rmax=rmax, algorithm=alg)
        ref = oracle.round_tt([c.clone() for c in inp], rmax=rmax, algorithm=alg, batch=True)
        assert ranks([c[0] for c in to_list(t.cores)]) == ranks([c[0] for c in ref]), (seed, alg, "ranks")
        X = oracle.tt_to_dense([c.double() for c in inp], batch=True)
        for b in range(0, B, max(1, B // 5)):
            e_o = rel_diff(oracle.tt_to_dense([c[b].cpu().double() for c in t.cores]), X[b])
            e_r = rel_diff(oracle.tt_to_dense([c[b].double() for c in ref]), X[b])
            if not math.isnan(e_r):  # (the reference divides by an exactly-zero sigma in batch mode: NaN; ours: finite)
                assert abs(e_o - e_r) <= 3e-5, (seed, alg, B, b, e_o, e_r)


def stress_product(seed):
    rng = np.random.RandomState(6000 + seed)
    N = int(rng.randint(2, 5))
    shape = [int(rng.randint(2, 8)) for _ in range(N)]
    ra, rb = int(rng.randint(2, 13)), int(rng.randint(2, 10))
    torch.manual_seed(seed)
    a = oracle.tt_randn(shape, ra, dtype=torch.float64); b = oracle.tt_randn(shape, rb, dtype=torch.float64)
    p = gpu(a) * gpu(b)
    want = oracle.tt_to_dense(a) * oracle.tt_to_dense(b)
    assert rel_diff(p.torch().cpu(), want) <= 1e-12, (seed, "product")
    eps = float(10.0 ** rng.uniform(-8, -2))
    p.round_tt(eps=eps)
    assert rel_diff(p.torch().cpu(), want) <= eps * (1 + 1e-6) + 1e-10, (seed, "round", eps)
    ref = oracle.round_tt(oracle.tt_mul(a, b), eps=eps)
    assert sum(p.ranks_tt.tolist()) <= sum(ranks(ref)), (seed, "ranks", p.ranks_tt.tolist(), ranks(ref))


def stress_truncated_svd(seed):
    rng = np.random.RandomState(7000 + seed)
    m, n = int(rng.randint(1, 200)), int(rng.randint(1, 300))
    dt = torch.float64 if seed % 2 else torch.float32
    torch.manual_seed(seed)
    k = int(rng.randint(1, min(m, n) + 1))
    M = (torch.randn(m, k, dtype=torch.float64) @ torch.randn(k, n, dtype=torch.float64)).to(dt)
    for alg in ("svd", "eig"):
        for lo in (True, False):
            eps = float(10.0 ** rng.uniform(-6, -1))
            L, R = tn.truncated_svd(M.cuda(), eps=eps, left_ortho=lo, algorithm=alg)
            Lr, Rr = oracle.truncated_svd(M, eps=eps, left_ortho=lo, algorithm=alg)
            err = rel_diff((L @ R).cpu(), M)
            floor = (3e-4 if alg == "eig" else 5e-6) if dt == torch.float32 else (1e-7 if alg == "eig" else 1e-10)
            assert err <= eps * (1 + 1e-4) + floor, (seed, alg, lo, m, n, k, eps, err)
            if dt == torch.float64 and alg == "svd":  # (noise-level ranks are only well defined there)
                assert L.shape[1] <= Lr.shape[1], (seed, alg, lo, m, n, L.shape, Lr.shape)
            orth = L if lo else R.T
            if alg == "svd" and k >= L.shape[1] and dt == torch.float64:
                sig_min_rel = 1e-3
                G = (orth.T @ orth).cpu().double()
                assert (G - torch.eye(G.shape[0], dtype=torch.float64)).abs().max() < (1e-3 if dt == torch.float32 else 1e-8), (seed, alg, lo, m, n)


def stress_cp(seed):
    rng = np.random.RandomState(8000 + seed)
    N = int(rng.randint(2, 6))
    shape = [int(rng.randint(3, 12)) for _ in range(N)]
    R = int(rng.randint(1, 6))
    torch.manual_seed(seed)
    fac = [torch.randn(i, R, dtype=torch.float64) for i in shape]
    X = oracle.cp_to_dense(fac)
    X = X / X.norm() + 10.0 ** rng.uniform(-4, -1) * torch.randn(shape, dtype=torch.float64) / math.sqrt(X.numel())
    K = int(rng.randint(1, 8))
    Rfit = max(1, min(R + int(rng.randint(-1, 2)), min(shape)))   # (R > I_n would draw random completion columns)
    t = tn.Tensor(X, ranks_cp=Rfit, max_iter=K, tol=-1.0, device="cuda")
    ref, errs = oracle.cp_als(X, Rfit, max_iter=K, tol=-1.0)
    e_o = rel_diff(oracle.cp_to_dense([c.cpu() for c in t.cores]), X)
    e_r = rel_diff(oracle.cp_to_dense(ref), X)
    assert abs(e_o - e_r) <= 1e-6 + 1e-3 * e_r, (seed, shape, R, Rfit, K, e_o, e_r)
    assert abs(t.cp_errors[-1] - e_o) <= 1e-6, (seed, "error estimate", t.cp_errors[-1], e_o)


def stress_dense_batch(seed):
    rng = np.random.RandomState(9000 + seed)
    N = int(rng.randint(3, 6))
    shape = [int(rng.randint(2, 10)) for _ in range(N)]
    B = int(rng.choice([1, 2, 5, 130]))
    dt = torch.float32 if seed % 2 else torch.float64
    torch.manual_seed(seed)
    X = torch.randn([B] + shape, dtype=torch.float64).to(dt)
    r = int(rng.randint(1, 6))
    t = tn.Tensor(X, ranks_tt=r, batch=True, device="cuda")
    ref = oracle.dense_to_tt(X, r, batch=True)
    assert ranks([c[0] for c in to_list(t.cores)]) == ranks([c[0] for c in ref]), (seed, "ranks")
    for b in range(0, B, max(1, B // 4)):
        e_o = rel_diff(oracle.tt_to_dense([c[b].cpu().double() for c in t.cores]), X[b])
        e_r = rel_diff(oracle.tt_to_dense([c[b].double() for c in ref]), X[b])
        assert abs(e_o - e_r) <= (3e-5 if dt == torch.float32 else 1e-9), (seed, B, b, e_o, e_r)
    # Tucker rounding of the batch (rmax mode; small modes: the reference itself only supports I <= R R' here)
    rk = int(rng.randint(1, 4))
    t.round_tucker(rmax=rk)
    assert max(t.ranks_tucker.tolist()) <= rk and torch.isfinite(t.torch()).all()


def stress_orthogonalize(seed):
    rng = np.random.RandomState(10000 + seed)
    N = int(rng.randint(2, 7))
    shape = [int(rng.randint(1, 12)) for _ in range(N)]
    rk = [1] + [int(rng.randint(1, 10)) for _ in range(N - 1)] + [1]
    if N > 2 and seed % 4 == 0:
        rk[int(rng.randint(1, N))] = int(rng.randint(65, 100))
    dt = torch.float32 if seed % 2 else torch.float64
    torch.manual_seed(seed)
    cores = [torch.randn(rk[k], shape[k], rk[k + 1], dtype=torch.float64).to(dt) for k in range(N)]
    X = dense(cores)
    mu = int(rng.randint(0, N))
    t = gpu(cores)
    t.orthogonalize(mu)
    tol_o = 1e-4 if dt == torch.float32 else 1e-10
    assert rel_diff(t.torch().cpu(), X) <= (2e-5 if dt == torch.float32 else 1e-11), (seed, "tensor changed")
    for k in range(N):
        c = t.cores[k].cpu().double()
        if k < mu:
            M = c.reshape(-1, c.shape[-1]); G = M.T @ M
        elif k > mu:
            M = c.reshape(c.shape[0], -1); G = M @ M.T
        else:
            continue
        assert (G - torch.eye(G.shape[0], dtype=torch.float64)).abs().max() < tol_o, (seed, N, shape, rk, mu, k)
    # batched truncated_svd, both orientations
    B, m, n = int(rng.randint(1, 6)), int(rng.randint(1, 90)), int(rng.randint(1, 120))
    M = torch.randn(B, m, n, dtype=torch.float64).to(dt)
    r = int(rng.randint(1, min(m, n) + 1))
    for lo in (True, False):
        L, R = tn.truncated_svd(M.cuda(), rmax=r, left_ortho=lo, batch=True)
        Lr, Rr = oracle.truncated_svd(M, rmax=r, left_ortho=lo, batch=True)
        e_o = rel_diff((L @ R).cpu(), M); e_r = rel_diff(Lr @ Rr, M)
        assert L.shape == Lr.shape and abs(e_o - e_r) <= (3e-5 if dt == torch.float32 else 1e-10), (seed, lo, B, m, n, r, e_o, e_r)


fails = 0
for name, fn, seeds in (("fp32 batch", stress_fp32_batch, range(40)), ("product", stress_product, range(40)),
                        ("truncated_svd", stress_truncated_svd, range(60)),
                        ("cp_als", stress_cp, range(40)), ("dense batch", stress_dense_batch, range(40)),
                        ("orthogonalize / batched truncated_svd", stress_orthogonalize, range(60))):
    for seed in seeds:
        try:
            fn(seed)
        except Exception as e:
            fails += 1; print(name, "seed", seed, "FAILED:", type(e).__name__, str(e)[:300].replace("\n", " "))
print("stress 2 done, failures:", fails)
