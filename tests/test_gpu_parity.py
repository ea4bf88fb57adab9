"""Parity of the HIP path (through the drop-in API -> C ABI) against the CPU oracle and the
golden vectors recorded from the reference, on a real MI355X.

Tolerances (stated per SURVEY 8c, measured margins in DESIGN.md):
  float64: ranks identical; bond singular values <= 1e-11 rel. to sigma_max; dense
           reconstruction <= 1e-10; sign-gauged cores <= 1e-8 (separated spectra).
  float32: ranks identical in rmax mode; approximation error agrees to <= 1e-5 absolute;
           bond singular values <= 2e-5 rel. to sigma_max; reconstruction <= 2e-5 when the
           spectrum is separated (flat `randn` spectra make the truncated subspace itself
           ill-conditioned: the reference's own f32-vs-f64 results differ by up to 7e-4 there).
"""
import math

import numpy as np
import pytest
import torch

import oracle
import tntorch_amd as tn
from parity import assert_tt_close, dense, load_case, load_meta, ranks, rel_diff, to_list

pytestmark = pytest.mark.gpu


def gpu_tensor(cores, batch=False):
    return tn.Tensor([c.cuda() for c in cores], batch=batch)


def tt_rel_err(a, b):
    """||a-b||/||b|| of two CPU trains through float64 inner products (valid down to ~1e-8)."""
    a = [c.double() for c in a]
    b = [c.double() for c in b]
    aa, bb, ab = oracle.tt_dot(a, a), oracle.tt_dot(b, b), oracle.tt_dot(a, b)
    return math.sqrt(max((aa + bb - 2 * ab).item(), 0.0) / bb.item())


# ------------------------------------------------------------------ golden vectors
def test_golden_round_eps_f64_svd():
    g = load_case("round_eps_f64")
    t = gpu_tensor(g["inp"])
    t.round_tt(eps=1e-8, algorithm="svd")
    assert_tt_close(to_list(t.cores), g["svd"], tol_dense=1e-10, tol_sv=1e-11, tol_cores=1e-8, what="round_eps_f64/svd")


def test_golden_round_eps_f64_eig():
    """'eig' at eps=1e-8 in float64 sits exactly on the Gram noise floor: a null eigenvalue that comes out
    slightly negative is clamped to 1e-8 (sigma = 1e-4, round.py:118-119) and survives delta, one that comes
    out slightly positive does not.  The reference's own test does not assert ranks for 'eig'
    (tests/test_round.py:52-59); neither do we: the represented tensor must match."""
    g = load_case("round_eps_f64")
    t = gpu_tensor(g["inp"])
    t.round_tt(eps=1e-8, algorithm="eig")
    ours, ref = to_list(t.cores), g["eig"]
    assert rel_diff(dense(ours), dense(ref)) <= 1e-10
    assert all(4 <= r <= 8 for r in ranks(ours)[1:-1])


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_golden_round_rmax_f32(alg):
    g = load_case("round_rmax_f32")
    t = gpu_tensor(g["inp"])
    t.round_tt(rmax=3, algorithm=alg)
    ours, ref = to_list(t.cores), g[alg]
    assert ranks(ours) == ranks(ref) == [1, 3, 3, 3, 3, 1]
    X = dense(g["inp"])
    e_ours, e_ref = rel_diff(dense(ours), X), rel_diff(dense(ref), X)
    assert abs(e_ours - e_ref) <= 1e-5, (e_ours, e_ref)
    for a, b in zip(oracle.bond_singular_values(ours), oracle.bond_singular_values(ref)):
        assert ((a - b).abs().max() / b.max()).item() <= 2e-5


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_golden_round_batch_f64(alg):
    g = load_case("round_batch_f64")
    t = gpu_tensor(g["inp"], batch=True)
    t.round_tt(rmax=2, algorithm=alg)
    assert t.batch and list(t.ranks_tt) == [1, 2, 2, 2, 1]
    for i in range(3):
        assert_tt_close(to_list(t.cores, i), [c[i] for c in g[alg]], tol_dense=1e-10, tol_sv=1e-11, what=f"batch item {i}")


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_golden_dense_f64(alg):
    g = load_case("dense_f64")
    t = tn.Tensor(g["X"].cuda(), ranks_tt=4, algorithm=alg)
    assert_tt_close(to_list(t.cores), g[alg], tol_dense=1e-10, tol_sv=1e-11, tol_cores=1e-8, what=f"dense_f64/{alg}")


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_golden_dense_batch_f32(alg):
    g = load_case("dense_batch_f32")
    X = g["X"]
    t = tn.Tensor(X.cuda(), ranks_tt=3, batch=True, algorithm=alg)
    assert list(t.ranks_tt) == [1, 3, 3, 3, 1]
    ours = t.torch().cpu().double()
    ref = oracle.tt_to_dense([c.double() for c in g[alg]], batch=True)
    for i in range(3):
        e_o = rel_diff(ours[i], X[i]); e_r = rel_diff(ref[i], X[i])
        assert abs(e_o - e_r) <= 1e-5, (i, e_o, e_r)


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_golden_c0_on_device(alg):
    """BASELINE config C0's tensor through the device dense entry (tn.Tensor(X, ranks_tt=4))."""
    g = load_case("c0_16x4_rmax4_f32")
    torch.manual_seed(0)
    X = torch.randn(16, 16, 16, 16)
    t = tn.Tensor(X.cuda(), ranks_tt=4, algorithm=alg)
    ours = to_list(t.cores)
    assert ranks(ours) == [1, 4, 4, 4, 1]
    e_o, e_r = rel_diff(dense(ours), X), rel_diff(dense(g[alg]), X)
    assert abs(e_o - e_r) <= 1e-5, (e_o, e_r)


def test_golden_truncated_svd():
    g = load_case("truncated_svd_f64")
    for c in load_meta()["cases"]["truncated_svd_f64"]["calls"]:
        kw = {k: c[k] for k in ("eps", "rmax", "delta") if k in c}
        M = g["M_" + c["M"]]
        u, v = tn.truncated_svd(M.cuda(), left_ortho=c["left_ortho"], algorithm=c["algorithm"], **kw)
        u, v = u.cpu(), v.cpu()
        ur, vr = g[f"call{c['i']}_left"], g[f"call{c['i']}_right"]
        assert u.shape == ur.shape and v.shape == vr.shape, c
        assert (u @ v - ur @ vr).abs().max() <= 1e-11, c
        r = u.shape[1]
        if c["left_ortho"]:
            assert (u.t() @ u - torch.eye(r, dtype=u.dtype)).abs().max() <= 1e-9, c
        else:
            assert (v @ v.t() - torch.eye(r, dtype=u.dtype)).abs().max() <= 1e-9, c
    for alg in ("svd", "eig"):
        u, v = tn.truncated_svd(g["Mb"].cuda(), batch=True, algorithm=alg)
        assert u.shape == (2, 32, 32) and v.shape == (2, 32, 32)
        assert (u.cpu() @ v.cpu() - g["Mb"]).abs().max() <= 1e-10
        for i in range(2):  # batch == loop over items (tests/test_round.py:21-38)
            u1, v1 = tn.truncated_svd(g["Mb"][i].cuda(), batch=False, algorithm=alg)
            assert (u1.cpu() @ v1.cpu() - u[i].cpu() @ v[i].cpu()).abs().max() <= 1e-10


def test_truncated_svd_errors():
    M = torch.rand(4, 5).cuda()
    with pytest.raises(ValueError):
        tn.truncated_svd(M, delta=0.1, eps=0.1)
    with pytest.raises(AssertionError):
        tn.truncated_svd(M, algorithm="qr")
    with pytest.raises(AssertionError):
        tn.truncated_svd(M, rmax=0)


def test_golden_orthogonalize():
    g = load_case("orthogonalize_f64")
    X = dense(g["inp"])
    for name, fn in [("left0", lambda t: t.left_orthogonalize(0)), ("right4", lambda t: t.right_orthogonalize(4)),
                     ("orth2", lambda t: t.orthogonalize(2)), ("orth4", lambda t: t.orthogonalize(4))]:
        t = gpu_tensor(g["inp"])
        fn(t)
        ours = to_list(t.cores)
        assert ranks(ours) == ranks(g[name]), name
        assert rel_diff(dense(ours), X) <= 1e-12, name
        # |core| agrees with the reference wherever the reference core is orthogonal (sign gauge only)
    t = gpu_tensor(g["inp"]); t.orthogonalize(2)
    c = to_list(t.cores)
    for k in (0, 1):
        L = c[k].reshape(-1, c[k].shape[-1])
        assert (L.t() @ L - torch.eye(L.shape[1], dtype=L.dtype)).abs().max() <= 1e-12
    for k in (3, 4):
        Rm = c[k].reshape(c[k].shape[0], -1)
        assert (Rm @ Rm.t() - torch.eye(Rm.shape[0], dtype=Rm.dtype)).abs().max() <= 1e-12


# ------------------------------------------------------------------ the reference's own property tests, on device
def test_ref_orthogonalization_property():
    """tests/test_round.py:7-18 (fewer trials, float64, device cores)."""
    rng = np.random.RandomState(0)
    torch.manual_seed(0)
    for _ in range(12):
        shape = rng.randint(1, 8, rng.randint(2, 6))
        gt = tn.rand(shape, ranks_tt=int(rng.randint(1, 6)), dtype=torch.float64, device="cuda")
        X = gt.torch()
        t = gt.clone()
        t.left_orthogonalize(0)
        assert tn.relative_error(X, t) <= 1e-7
        t.right_orthogonalize(t.dim() - 1)
        assert tn.relative_error(X, t) <= 1e-7
        t.orthogonalize(int(rng.randint(t.dim())))
        assert tn.relative_error(X, t) <= 1e-7


@pytest.mark.parametrize("alg,tol", [("svd", 1e-4), ("eig", 1e-7)])
def test_ref_round_tt_rank_recovery(alg, tol):
    """tests/test_round.py:41-59: gt+gt rounds back to gt's ranks ('svd' asserts the ranks)."""
    rng = np.random.RandomState(1)
    torch.manual_seed(1)
    for _ in range(8):
        shape = rng.randint(1, 8, rng.randint(8, 10))
        gt = tn.rand(shape, ranks_tt=int(rng.randint(1, 10)), dtype=torch.float64, device="cuda")
        gt.round_tt(1e-8, algorithm=alg)
        t = gt + gt
        t.round_tt(1e-8, algorithm=alg)
        assert tn.relative_error(gt, t / 2) <= tol
        if alg == "svd":
            assert max(gt.ranks_tt) == max(t.ranks_tt)


def test_ref_gpu_tt():
    """tests/test_gpu.py:9-13."""
    torch.manual_seed(0)
    X = torch.randn(16, 16, 16, dtype=torch.float64)
    y1 = oracle.tt_to_dense(oracle.dense_to_tt(X, 3))
    y2 = tn.Tensor(X, ranks_tt=3, device="cuda").torch().cpu()
    assert torch.abs(y1 - y2).max() < 1e-5


@pytest.mark.parametrize("alg", ["svd", "eig"])
@pytest.mark.parametrize("shape,r,dt", [([16] * 4, 8, torch.float64), ([32] * 4, 16, torch.float32)])
def test_dense_large_bond(shape, r, dt, alg):
    """Dense -> TT with I*r = 128 / 512: the bond eigenproblems exceed one workgroup and run through the
    block-Jacobi driver (C1's structure at a size the oracle finishes in seconds)."""
    torch.manual_seed(5)
    low = oracle.tt_to_dense(oracle.tt_randn(shape, r, dtype=torch.float64))   # TT rank r + noise below it
    X = (low / low.norm() + 1e-3 * torch.randn(shape, dtype=torch.float64) / math.sqrt(low.numel())).to(dt)
    ref = oracle.dense_to_tt(X, r, algorithm=alg)
    t = tn.Tensor(X, ranks_tt=r, device="cuda", algorithm=alg)
    assert list(t.ranks_tt) == ranks(ref)
    e_o = rel_diff(t.torch().cpu().double(), X.double())
    e_r = rel_diff(oracle.tt_to_dense([c.double() for c in ref]), X.double())
    # truncation error within 1 % (fp32) / 1e-6 (fp64) of the reference algorithm's
    assert e_o <= e_r * (1 + (1e-2 if dt == torch.float32 else 1e-6)) + (2e-6 if dt == torch.float32 else 1e-12), (e_o, e_r)
    for k, c in enumerate(t.cores[1:], 1):   # right-orthonormal cores (sweep ends at core 0)
        M = c.reshape(c.shape[0], -1).double().cpu()
        assert (M @ M.T - torch.eye(M.shape[0], dtype=torch.float64)).abs().max() < (3e-5 if dt == torch.float32 else 1e-11)


@pytest.mark.parametrize("alg", ["svd", "eig"])
@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_round_rank_above_64(dt, alg):
    """TT ranks above the 64-column TSQR panel (here 96: a rank-48 train added to itself): blocked QR with
    explicit Q in the L2R sweep, 96 x 96 bond eigenproblems in the R2L sweep."""
    torch.manual_seed(7)
    g = oracle.tt_randn([12, 10, 12, 10, 12], 48, dtype=torch.float64)
    inp = [c.to(dt) for c in oracle.tt_add(g, g)]
    assert max(ranks(inp)) == 96
    ref = oracle.round_tt([c.clone() for c in inp], rmax=48, algorithm=alg)
    t = gpu_tensor(inp)
    t.round_tt(rmax=48, algorithm=alg)
    ours = to_list(t.cores)
    assert ranks(ours) == ranks(ref)
    tolr = 3e-5 if dt == torch.float32 else 1e-10
    X = dense(inp)
    assert rel_diff(dense(ours), dense(ref)) <= tolr
    assert rel_diff(dense(ours), X) <= tolr          # the redundant input is reproduced
    # eps mode on the same train
    t2 = gpu_tensor(inp)
    t2.round_tt(eps=1e-3 if dt == torch.float32 else 1e-8, algorithm=alg)
    assert max(t2.ranks_tt) <= 48 and rel_diff(dense(to_list(t2.cores)), X) <= (2e-3 if dt == torch.float32 else 2e-8)
    # orthogonalize() with wide-rank cores: all cores left of mu orthonormal, tensor unchanged
    t3 = gpu_tensor(inp)
    t3.orthogonalize(3)
    assert rel_diff(dense(to_list(t3.cores)), X) <= tolr
    for k in range(3):
        c = t3.cores[k]
        M = c.reshape(-1, c.shape[-1]).double().cpu()
        assert (M.T @ M - torch.eye(M.shape[1], dtype=torch.float64)).abs().max() < (5e-5 if dt == torch.float32 else 1e-11)


def test_quirks():
    # default eps=1e-14 drops exactly-zero tails even with rmax (SURVEY A-1)
    t = tn.Tensor(torch.ones(4, 4, 4, dtype=torch.float64).cuda(), ranks_tt=3)
    assert list(t.ranks_tt) == [1, 1, 1, 1]
    assert torch.allclose(t.torch().cpu(), torch.ones(4, 4, 4, dtype=torch.float64))
    # all-zero train rounds to rank 1 zeros (SURVEY A-5)
    z = tn.Tensor([torch.zeros(1, 5, 3).cuda(), torch.zeros(3, 5, 3).cuda(), torch.zeros(3, 5, 1).cuda()])
    z.round_tt()
    assert list(z.ranks_tt) == [1, 1, 1, 1] and z.torch().abs().max() == 0
    # rounding rebinds list entries: tensors held by the caller are not modified
    g = oracle.tt_randn([6] * 4, 3, dtype=torch.float32)
    held = [c.cuda() for c in g]
    copies = [c.clone() for c in held]
    t = tn.Tensor(held); t.round_tt(rmax=2)
    for a, b in zip(held, copies):
        assert torch.equal(a, b)
    # rmax list length is checked (tensor.py:2029)
    with pytest.raises(AssertionError):
        tn.Tensor(held).round_tt(rmax=[2, 2])
    # the kernels are not differentiable: refuse instead of silently cutting the graph
    with pytest.raises(NotImplementedError):
        tn.Tensor([c.clone().requires_grad_() for c in held]).round_tt(rmax=2)


# ------------------------------------------------------------------ consumers on the device
@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_device_dot_norm_decompress(dt):
    """metrics.dot / norm / dist / relative_error and Tensor.torch() on device tensors (ttr_gemm chains)
    against the oracle's float64 contraction."""
    torch.manual_seed(21)
    a = oracle.tt_randn([9, 7, 8, 6, 5], 5, dtype=torch.float64)
    b = oracle.tt_randn([9, 7, 8, 6, 5], 3, dtype=torch.float64)
    ta, tb = gpu_tensor([c.to(dt) for c in a]), gpu_tensor([c.to(dt) for c in b])
    rt = 2e-5 if dt == torch.float32 else 1e-12
    A, Bd = oracle.tt_to_dense(a), oracle.tt_to_dense(b)
    assert rel_diff(ta.torch().cpu().double(), A) <= rt
    ref = oracle.tt_dot(a, b).item()
    scale = math.sqrt(oracle.tt_dot(a, a).item() * oracle.tt_dot(b, b).item())
    assert abs(tn.dot(ta, tb).item() - ref) <= rt * scale
    assert abs(tn.norm(ta).item() - A.norm().item()) <= rt * A.norm().item()
    assert abs(tn.dist(ta, tb).item() - (A - Bd).norm().item()) <= 50 * rt * A.norm().item()
    # dense-vs-TT branches (metrics.py:135-151)
    Xd = A.to(dt).cuda()
    assert tn.relative_error(Xd, ta).item() <= 10 * rt
    assert abs(tn.relative_error(Xd, tb).item() - ((A - Bd).norm() / A.norm()).item()) <= 10 * rt
    assert abs(tn.dot(Xd, tb).item() - (A * Bd).sum().item()) <= rt * scale
    # boundary ranks > 1 are summed away by torch() (tensor.py:1639-1687)
    cs = [torch.randn(3, 4, 2, dtype=torch.float64), torch.randn(2, 5, 4, dtype=torch.float64)]
    full = torch.einsum("aib,bjc->ij", cs[0], cs[1])
    assert rel_diff(gpu_tensor([c.to(dt) for c in cs]).torch().cpu().double(), full) <= rt
    # batch decompression
    gb = oracle.tt_randn([6, 5, 4], 3, dtype=torch.float64, batch_size=3)
    tb3 = gpu_tensor([c.to(dt) for c in gb], batch=True)
    assert rel_diff(tb3.torch().cpu().double(), oracle.tt_to_dense(gb, batch=True)) <= rt


def test_device_norm_metric_size():
    """||t|| of a 64^8 rank-32 fp32 train on the device vs the oracle's float64 inner product."""
    torch.manual_seed(3)
    g = oracle.tt_randn([64] * 8, 32, dtype=torch.float32)
    t = gpu_tensor(g)
    ref = math.sqrt(oracle.tt_dot([c.double() for c in g], [c.double() for c in g]).item())
    assert abs(tn.norm(t).item() - ref) <= 1e-5 * ref


# ------------------------------------------------------------------ Tucker rounding / round() on the device
def _tk(cores):
    return [c.shape[-2] for c in cores]


def _tucker_dense(t):
    return oracle.tucker_to_dense([c.cpu().double() for c in t.cores], [None if U is None else U.cpu().double() for U in t.Us],
                                  batch=t.batch)


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_golden_round_tucker(alg):
    g = load_case("round_tucker_eps_f64")
    t = gpu_tensor(g["inp"])
    t.round_tucker(eps=1e-8, algorithm=alg)
    assert _tk(t.cores) == _tk(g[f"{alg}_cores"]) and tuple(t.shape) == (12, 10, 14, 11)
    ref = oracle.tucker_to_dense(g[f"{alg}_cores"], g[f"{alg}_Us"])
    assert rel_diff(_tucker_dense(t), ref) <= 1e-10
    assert rel_diff(t.torch().cpu(), ref) <= 1e-10          # device decompression with factors
    for U in t.Us:                                           # orthonormal factors (left_ortho=True)
        Uc = U.cpu()
        assert (Uc.T @ Uc - torch.eye(Uc.shape[1], dtype=Uc.dtype)).abs().max() < 1e-10
    # float32, rank cap
    g = load_case("round_tucker_rmax_f32")
    t = gpu_tensor(g["inp"])
    t.round_tucker(rmax=3, algorithm=alg)
    assert _tk(t.cores) == [3, 3, 3, 3]
    X = dense(g["inp"])
    e_o = rel_diff(_tucker_dense(t), X)
    e_r = rel_diff(oracle.tucker_to_dense([c.double() for c in g[f"{alg}_cores"]], [U.double() for U in g[f"{alg}_Us"]]), X)
    assert abs(e_o - e_r) <= 1e-5
    # batch
    g = load_case("round_tucker_batch_f64")
    t = gpu_tensor(g["inp"], batch=True)
    t.round_tucker(rmax=2, algorithm=alg)
    ref = oracle.tucker_to_dense(g[f"{alg}_cores"], g[f"{alg}_Us"], batch=True)
    assert rel_diff(_tucker_dense(t), ref) <= 1e-10


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_golden_ctor_tucker_and_round(alg):
    g = load_case("ctor_tucker_f64")
    t = tn.Tensor(g["inp"], ranks_tucker=4, ranks_tt=3, algorithm=alg, device="cuda")   # dense ST-HOSVD + TT-SVD
    assert t.ranks_tt.tolist() == [1, 3, 3, 3, 1] and t.ranks_tucker.tolist() == [4, 4, 4, 4]
    ref = oracle.tucker_to_dense(g[f"{alg}_cores"], g[f"{alg}_Us"])
    assert rel_diff(_tucker_dense(t), ref) <= 1e-9
    t = tn.Tensor(g["inp"], ranks_tucker=4, algorithm=alg, device="cuda")                # Tucker only
    assert t.ranks_tucker.tolist() == [4, 4, 4, 4]
    rt, _ = oracle.dense_to_tucker_tt(g["inp"], ranks_tucker=4, algorithm=alg)
    e_r = rel_diff(oracle.tucker_to_dense(*oracle.dense_to_tucker_tt(g["inp"], ranks_tucker=4, algorithm=alg)), g["inp"])
    assert abs(rel_diff(_tucker_dense(t), g["inp"]) - e_r) <= 1e-10
    g = load_case("round_general_f64")
    t = tn.round(gpu_tensor(g["inp"]), eps=1e-6, algorithm=alg)
    assert ranks(to_list(t.cores)) == ranks(g[f"{alg}_cores"]) and _tk(t.cores) == _tk(g[f"{alg}_cores"])
    ref = oracle.tucker_to_dense(g[f"{alg}_cores"], g[f"{alg}_Us"])
    assert rel_diff(_tucker_dense(t), ref) <= 1e-9
    u = t + t                                              # factors survive + / dot / scalar *
    assert rel_diff(u.torch().cpu(), 2 * ref) <= 1e-9
    assert abs(tn.dot(t, t).item() - (ref * ref).sum().item()) <= 1e-9 * (ref * ref).sum().item()


def test_known_answer_eps_ctor_device():
    """decompositions.ipynb cell 14 on the device: tn.Tensor(full, eps=1e-5) -> TT [1,4,6,1], Tucker 4,5,6, 8.3402e-06."""
    from parity import analytic_128
    full = analytic_128()
    t = tn.Tensor(full, eps=1e-5, device="cuda")
    assert t.ranks_tt.tolist() == [1, 4, 6, 1] and t.ranks_tucker.tolist() == [4, 5, 6]
    assert abs(rel_diff(_tucker_dense(t), full) - 8.340228167320888e-06) < 1e-9


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_round_tucker_wide_modes(dt):
    """Mode size 96 > 64: the factor QR runs through the blocked path, the factor eigenproblem is 96 x 96."""
    torch.manual_seed(17)
    S, I = [7, 9, 8], [96, 20, 96]
    g = oracle.tt_randn(S, 4, dtype=torch.float64)
    Us = [torch.randn(i, s_, dtype=torch.float64) for i, s_ in zip(I, S)]
    inp = [c.to(dt) for c in oracle.tucker_absorb(g, Us)]
    cores_r, Us_r = oracle.round_tucker(inp, None, eps=1e-4 if dt == torch.float32 else 1e-9)
    t = gpu_tensor(inp)
    t.round_tucker(eps=1e-4 if dt == torch.float32 else 1e-9)
    assert _tk(t.cores) == _tk(cores_r) == [4, 9, 4]
    X = dense(inp)
    assert rel_diff(_tucker_dense(t), X) <= (2e-4 if dt == torch.float32 else 1e-9)
    assert rel_diff(_tucker_dense(t), oracle.tucker_to_dense([c.double() for c in cores_r], [U.double() for U in Us_r])) <= (
        2e-4 if dt == torch.float32 else 1e-9)


# ------------------------------------------------------------------ CP-ALS on the device (SURVEY 8f-1, config C4's algorithm)
def test_golden_cp_als():
    """tn.Tensor(X, ranks_cp=R) on the device vs the reference's factors (golden): HOSVD-initialised ALS is
    deterministic up to the signs of the initial eigenvectors, which cancel in the reconstruction."""
    g = load_case("cp_als")
    runs = load_meta()["cases"]["cp_als"]["runs"]
    X = g["inp"]
    for name, tol_rec in (("r3_it1", 1e-9), ("r3_it25", 1e-7), ("r5_it4", 1e-6)):
        t = tn.Tensor(X, ranks_cp=runs[name]["R"], max_iter=runs[name]["max_iter"], device="cuda")
        assert all(c.dim() == 2 and c.is_cuda for c in t.cores) and tuple(t.shape) == (12, 10, 9, 11)
        ours = oracle.cp_to_dense([c.cpu() for c in t.cores])
        ref = oracle.cp_to_dense(g[name])
        assert rel_diff(ours, ref) <= tol_rec, (name, rel_diff(ours, ref))
        assert abs(rel_diff(ours, X) - runs[name]["relerr"]) <= 1e-8
        assert abs(t.cp_errors[-1] - runs[name]["relerr"]) <= 1e-6      # algebraic error estimate of the sweep
        assert rel_diff(t.torch().cpu(), ours) <= 1e-12                   # CP -> TT diagonal cores on the device
    Y = g["f32_inp"]
    t = tn.Tensor(Y, ranks_cp=4, max_iter=6, device="cuda")
    e_o = rel_diff(oracle.cp_to_dense([c.cpu().double() for c in t.cores]), Y)
    assert abs(e_o - runs["f32_r4_it6"]["relerr"]) <= 2e-3


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_cp_als_recovers_low_rank(dt):
    """tests/test_tensor.py:52-62 style: exact rank-R data is recovered; 5-mode and 2-mode tensors exercise
    every branch of the fused MTTKRP (trailing / leading contractions, no contraction)."""
    torch.manual_seed(23)
    for shape, R in (([9, 8, 7, 6, 5], 3), ([40, 30], 4), ([20, 18, 16], 5)):
        fac = [torch.randn(i, R, dtype=torch.float64) for i in shape]
        X = oracle.cp_to_dense(fac).to(dt)
        t = tn.Tensor(X, ranks_cp=R, max_iter=60, tol=1e-9 if dt == torch.float64 else 1e-6, device="cuda")
        err = rel_diff(oracle.cp_to_dense([c.cpu().double() for c in t.cores]), X)
        ref_cores, ref_err = oracle.cp_als(X.double(), R, max_iter=60, tol=1e-9 if dt == torch.float64 else 1e-6)
        e_ref = rel_diff(oracle.cp_to_dense(ref_cores), X)
        assert err <= max(10 * e_ref, 5e-3 if dt == torch.float32 else 1e-5), (shape, err, e_ref)


# ------------------------------------------------------------------ producers on the device
def test_golden_producers():
    import operator
    g = load_case("producers_f64")
    a, b = gpu_tensor(g["a"]), gpu_tensor(g["b"])
    p = a * b                                              # ttr_core_kron
    assert max((x.cpu() - y).abs().max().item() for x, y in zip(p.cores, g["prod"])) < 1e-15
    ts = [gpu_tensor(g[f"t{i}"]) for i in range(5)]
    red = tn.reduce(ts, operator.add, eps=1e-6)            # tree of + and round() on the device
    ref = g["red_dense"]
    assert red.ranks_tt.tolist() == g["red_ranks_tt"].tolist() and red.ranks_tucker.tolist() == g["red_ranks_tucker"].tolist()
    assert rel_diff(red.torch().cpu(), ref) <= 1e-9
    red3 = tn.reduce(ts, operator.add, rmax=3)
    ref3 = oracle.tt_to_dense(g["red3_cores"])
    assert rel_diff(red3.torch().cpu(), ref3) <= 1e-9
    # arithmetics.ipynb cell 1.  The reference (float32, LAPACK) gets exactly-zero trailing singular values on this
    # all-ones structure and eps = 1e-14 then drops them; QL/Jacobi leave them at the 1e-8 round-off level, which
    # 1e-14 keeps -- in float64 (or with any eps above round-off) the ranks collapse to 1 here as well.
    ones = tn.ones([32] * 4, device="cuda", dtype=torch.float64)
    assert tn.round((ones + ones) * (ones - 2)).ranks_tt.tolist() == [1, 1, 1, 1, 1]
    ones = tn.ones([32] * 4, device="cuda")
    assert tn.round((ones + ones) * (ones - 2), eps=1e-6).ranks_tt.tolist() == [1, 1, 1, 1, 1]
    # batch + float32 product against dense
    torch.manual_seed(31)
    x = oracle.tt_randn([5, 6, 4], 3, dtype=torch.float32, batch_size=2)
    y = oracle.tt_randn([5, 6, 4], 2, dtype=torch.float32, batch_size=2)
    pb = gpu_tensor(x, batch=True) * gpu_tensor(y, batch=True)
    want = oracle.tt_to_dense([c.double() for c in x], batch=True) * oracle.tt_to_dense([c.double() for c in y], batch=True)
    assert rel_diff(pb.torch().cpu(), want) <= 2e-6


# ------------------------------------------------------------------ randomised shapes (ragged modes, wide cores, ranks 1..80)
@pytest.mark.parametrize("seed", list(range(12)) + [99, 135])   # 99 / 135: wide cores with 150-160 exactly dependent columns
def test_random_trains_vs_oracle(seed):
    """Seeded random trains: N in 2..6, ragged mode sizes 1..9 (one mode up to 40), ranks 1..12 (one bond up to 80,
    i.e. above the 64-column TSQR panel), float64; eps mode and rmax mode, both algorithms, against the oracle:
    identical ranks in rmax mode, reconstruction within 1e-9, eps bound respected."""
    rng = np.random.RandomState(1000 + seed)
    N = int(rng.randint(2, 7))
    shape = [int(rng.randint(1, 10)) for _ in range(N)]
    shape[int(rng.randint(N))] = int(rng.randint(10, 41))
    rk = [1] + [int(rng.randint(1, 13)) for _ in range(N - 1)] + [1]
    if N > 2 and seed % 3 == 0:
        rk[int(rng.randint(1, N))] = int(rng.randint(65, 81))
    torch.manual_seed(seed)
    cores = [torch.randn(rk[k], shape[k], rk[k + 1], dtype=torch.float64) for k in range(N)]
    inp = oracle.tt_add(cores, oracle.tt_scale(cores, 0.5)) if seed % 2 else cores   # redundant ranks every other seed
    X = dense(inp)
    for alg in ("svd", "eig"):
        rmax = int(rng.randint(1, 9))
        ref = oracle.round_tt([c.clone() for c in inp], rmax=rmax, algorithm=alg)
        t = gpu_tensor(inp)
        t.round_tt(rmax=rmax, algorithm=alg)
        ours = to_list(t.cores)
        # 'eig' on a redundant train decides the rank of exactly-dependent directions on Gram round-off (1e-8): the
        # reference's own test does not assert ranks there (tests/test_round.py:52-59); the error must still agree
        if alg == "svd" or not seed % 2:
            assert ranks(ours) == ranks(ref), (seed, alg, shape, rk, rmax)
        else:
            assert all(a <= b for a, b in zip(ranks(ours), [rmax if 0 < i < N else 1 for i in range(N + 1)]))
        e_o, e_r = rel_diff(dense(ours), X), rel_diff(dense(ref), X)
        assert abs(e_o - e_r) <= 1e-9 + 1e-6 * e_r + (1e-7 if alg == "eig" else 0.0), (seed, alg, shape, rk, rmax, e_o, e_r)
        eps = float(10.0 ** rng.uniform(-8, -1))
        t = gpu_tensor(inp)
        t.round_tt(eps=eps, algorithm=alg)
        ref = oracle.round_tt([c.clone() for c in inp], eps=eps, algorithm=alg)
        e_o = rel_diff(dense(to_list(t.cores)), X)
        assert e_o <= eps * (1 + 1e-6) + 1e-9, (seed, alg, eps, e_o)
        assert sum(ranks(to_list(t.cores))) <= sum(ranks(ref)) + (0 if alg == "svd" else N), (seed, alg, eps)


def test_stream_chunks_identical():
    """Batches >= 128 run as two sub-batches on separate streams writing into shared result tensors: the result
    must be bit-identical to the single-stream sweep (odd split 65 / 66, both algorithms, explicit-Q ranks too)."""
    from tntorch_amd import _hipops
    torch.manual_seed(41)
    for shape, r in (([6, 7, 5, 8], 5), ([4, 6, 4], 70)):
        g = oracle.tt_randn(shape, r, dtype=torch.float32, batch_size=131)
        inp = [c.cuda() for c in oracle.tt_add(g, g, batch=True)]
        for alg in ("svd", "eig"):
            outs = []
            for enabled in (True, False):
                _hipops.STREAM_CHUNKS_ENABLED = enabled
                try:
                    t = tn.Tensor([c.clone() for c in inp], batch=True)
                    t.round_tt(rmax=3, algorithm=alg)
                finally:
                    _hipops.STREAM_CHUNKS_ENABLED = True
                torch.cuda.synchronize()
                outs.append(t)
            for a, b in zip(outs[0].cores, outs[1].cores):
                assert a.shape == b.shape and a.is_contiguous() and torch.equal(a, b)
            # the chunked sweep lays the cores out back to back: the gather's packing step is a view, not a copy
            flat = tn.dist_batch.pack_cores(outs[0].cores)
            assert flat.data_ptr() == outs[0].cores[0].data_ptr() and flat.numel() == sum(c.numel() for c in outs[0].cores)
            assert tn.dist_batch.pack_cores(outs[1].cores).data_ptr() != outs[1].cores[0].data_ptr()
            X = oracle.tt_to_dense([c.cpu().double() for c in inp], batch=True)
            ref = oracle.round_tt([c.cpu() for c in inp], rmax=3, algorithm=alg, batch=True)
            e_o = rel_diff(outs[0].torch().cpu(), X)
            e_r = rel_diff(oracle.tt_to_dense([c.double() for c in ref], batch=True), X)
            assert abs(e_o - e_r) <= 2e-5


@pytest.mark.parametrize("seed", range(6))
def test_random_dense_and_tucker_vs_oracle(seed):
    """Seeded random dense tensors (3..5 modes, ragged sizes 2..14): the TT / Tucker-TT constructors and
    round_tucker on the device against the oracle -- ranks identical, approximation error within 1e-9 (float64)."""
    rng = np.random.RandomState(2000 + seed)
    N = int(rng.randint(3, 6))
    shape = [int(rng.randint(2, 15)) for _ in range(N)]
    torch.manual_seed(100 + seed)
    r = int(rng.randint(2, 5))
    low = oracle.tt_to_dense(oracle.tt_randn(shape, r, dtype=torch.float64))
    X = low / low.norm() + 10.0 ** rng.uniform(-6, -2) * torch.randn(shape, dtype=torch.float64) / math.sqrt(low.numel())
    for alg in ("svd", "eig"):
        rt = int(rng.randint(1, 6))
        ref = oracle.dense_to_tt(X, rt, algorithm=alg)
        t = tn.Tensor(X, ranks_tt=rt, algorithm=alg, device="cuda")
        assert t.ranks_tt.tolist() == ranks(ref), (seed, alg, shape, rt)
        e_o, e_r = rel_diff(t.torch().cpu(), X), rel_diff(dense(ref), X)
        assert abs(e_o - e_r) <= 1e-9 + 1e-6 * e_r, (seed, alg, shape, rt, e_o, e_r)
        rk = int(rng.randint(1, 5))
        cores_r, Us_r = oracle.dense_to_tucker_tt(X, ranks_tucker=rk, ranks_tt=rt, algorithm=alg)
        t2 = tn.Tensor(X, ranks_tucker=rk, ranks_tt=rt, algorithm=alg, device="cuda")
        assert t2.ranks_tucker.tolist() == [c.shape[1] for c in cores_r] and t2.ranks_tt.tolist() == ranks(cores_r)
        e_o, e_r = rel_diff(t2.torch().cpu(), X), rel_diff(oracle.tucker_to_dense(cores_r, Us_r), X)
        assert abs(e_o - e_r) <= 1e-9 + 1e-5 * e_r, (seed, alg, shape, rk, rt, e_o, e_r)
        eps = float(10.0 ** rng.uniform(-7, -2))
        c3, U3 = oracle.round_tucker(ref, None, eps=eps, algorithm=alg)
        t3 = gpu_tensor(ref)
        t3.round_tucker(eps=eps, algorithm=alg)
        assert rel_diff(t3.torch().cpu(), dense(ref)) <= eps * (1 + 1e-6) + 1e-9
        assert sum(t3.ranks_tucker.tolist()) <= sum(c.shape[1] for c in c3) + (0 if alg == "svd" else N)


@pytest.mark.parametrize("scales", [[1e15, 1e-15, 1e12, 1e-12, 1.0], [1.0, 1e-15, 1.0, 1.0, 1.0], [1e-15, 1.0, 1.0, 1.0, 1.0],
                                    [1.0, 1.0, 1.0, 1.0, 1e-15], [1e15, 1.0, 1.0, 1.0, 1.0]])
def test_fp32_badly_scaled_cores(scales):
    """fp32 cores scaled by 1e+-15 (legitimate fp32 data): squares of their entries leave the fp32 range.  The QR
    blocks are factored at the exponent of their largest entry and the sweep carries the R factors' exponents
    separately; the reference (LAPACK) is fine here too."""
    torch.manual_seed(0)
    g = oracle.tt_randn([8, 9, 7, 8, 6], 5, dtype=torch.float32)
    inp = [c * s for c, s in zip(oracle.tt_add(g, g), scales)]
    X = dense(inp)
    t = gpu_tensor(inp)
    t.round_tt(rmax=5)
    assert t.ranks_tt.tolist() == [1, 5, 5, 5, 5, 1]
    e_r = rel_diff(dense(oracle.round_tt([c.clone() for c in inp], rmax=5)), X)
    assert rel_diff(dense(to_list(t.cores)), X) <= max(5e-6, 3 * e_r)
    u = gpu_tensor(inp)
    u.orthogonalize(2)
    assert rel_diff(dense(to_list(u.cores)), X) <= 5e-6


@pytest.mark.parametrize("scale", [1e18, 1e-18])
def test_fp32_out_of_range_inputs(scale):
    """Dense tensors / matrices whose Gram matrices leave the fp32 range: dense -> TT, truncated_svd and
    round_tucker take the binary exponent out first (the reference's LAPACK calls are scale safe)."""
    torch.manual_seed(4)
    low = oracle.tt_to_dense(oracle.tt_randn([10, 9, 8, 7], 3, dtype=torch.float64))
    X = ((low / low.norm() + 1e-4 * torch.randn(low.shape, dtype=torch.float64) / math.sqrt(low.numel())) * scale).float()
    # (references are run on the UNSCALED data: at 1e-18 the reference's absolute zero guard, round.py:137-145,
    # returns zeros, and at 1e18 its float32 'svd' is simply scale safe)
    Xu = (X.double() / scale).float()
    t = tn.Tensor(X, ranks_tt=3, device="cuda")
    ref = oracle.dense_to_tt(Xu, 3)
    e_o, e_r = rel_diff(t.torch().cpu(), X), rel_diff(dense(ref), Xu)
    assert math.isfinite(e_o) and abs(e_o - e_r) <= 1e-5
    M = X.reshape(90, 56)
    for lo in (True, False):
        L, R = tn.truncated_svd(M.cuda(), eps=1e-3, left_ortho=lo)
        Lr, Rr = oracle.truncated_svd(Xu.reshape(90, 56), eps=1e-3, left_ortho=lo)
        assert L.shape == Lr.shape and rel_diff((L @ R).cpu(), M) <= 1e-3 * (1 + 1e-3)
    # (reference on the UNSCALED train: at 1e-18 its absolute zero guard, round.py:137-145, would return zeros)
    cr, Ur = oracle.round_tucker([c / (scale if k == 0 else 1.0) for k, c in enumerate(to_list(t.cores))], None, rmax=3)
    t.round_tucker(rmax=3)
    e_o, e_r = rel_diff(t.torch().cpu(), X), rel_diff(oracle.tucker_to_dense(cr, Ur), X / scale)
    assert math.isfinite(e_o) and abs(e_o - e_r) <= 1e-4


def test_high_order_fp32_no_overflow():
    """A 14-core fp32 train with ||X|| = 6e19: squared norms / Gram entries would overflow fp32 (the reference's LAPACK
    rescales internally and stays finite); the device sweep takes exact powers of two out of the R factors."""
    torch.manual_seed(0)
    g = oracle.tt_randn([32] * 14, 24, dtype=torch.float32)
    inp = oracle.tt_add(g, g)
    t = gpu_tensor(inp)
    t.round_tt(rmax=24)
    assert all(torch.isfinite(c).all() for c in t.cores) and t.ranks_tt.tolist() == [1] + [24] * 13 + [1]
    assert tt_rel_err(to_list(t.cores), inp) <= 6e-5
    tb = gpu_tensor([torch.stack([c, 3 * c]) for c in inp], batch=True)     # per-item exponents in batch mode
    tb.round_tt(rmax=24)
    assert all(torch.isfinite(c).all() for c in tb.cores)
    assert tt_rel_err([c[0] for c in to_list(tb.cores)], inp) <= 6e-5
    assert tt_rel_err([c[1] for c in to_list(tb.cores)], [3 * c for c in inp]) <= 6e-5


# ------------------------------------------------------------------ BASELINE-size configs
def _metric_input(B, seed=0):
    """g+g with g = randn TT, shape [64]*8, rank 32, float32 (the metric's workload)."""
    torch.manual_seed(seed)
    g = oracle.tt_randn([64] * 8, 32, dtype=torch.float32, batch_size=B)
    return oracle.tt_add(g, g, batch=True)


def test_metric_config_vs_oracle():
    """64^8 rank 64 -> 32 fp32, batch of 2: ranks, bond singular values and the train itself vs the oracle."""
    inp = _metric_input(2)
    t = gpu_tensor(inp, batch=True)
    t.round_tt(rmax=32)
    assert list(t.ranks_tt) == [1] + [32] * 7 + [1]
    for i in range(2):
        ref = oracle.round_tt([c[i] for c in inp], rmax=32, algorithm="svd")   # the device default against the oracle default
        ours = to_list(t.cores, i)
        assert ranks(ours) == ranks(ref)
        # 1e-5 (fp32); measured 2.6e-6 .. 4e-6 -- the bound must catch a 4x regression, not hide a 5x one
        assert tt_rel_err(ours, ref) <= 1e-5
        assert tt_rel_err(ours, [c[i] for c in inp]) <= 1e-5  # rank-32 redundant input is reproduced
        so, sr = oracle.bond_singular_values(ours), oracle.bond_singular_values(ref)
        assert all(((a - b).abs().max() / b.max()).item() <= 1e-5 for a, b in zip(so, sr))


def test_mixed_rank_batch_vs_oracle():
    """A batch whose items take DIFFERENT routes through the rank-revealing sweep (decided per item on the device): a rank-inflated
    train (g + g, numerical rank 32 of 64: packed rows, 32 x 32 eigenproblems, flat spectra), a genuine rank-64 train (nothing
    packed, full-size eigenproblems, the cut goes through a flat spectrum) and a train of numerical rank 20 (packed; 12 of the
    32 kept directions are null: orthonormal completion).  Every item: the oracle's ranks, right-orthonormal cores, and an
    approximation error equal to the oracle's (the truncated subspace of a flat spectrum is not unique: the trains themselves
    are only compared where the input is reproduced)."""
    N, I = 6, 64
    torch.manual_seed(41)
    g32 = oracle.tt_randn([I] * N, 32, dtype=torch.float32)
    g20 = oracle.tt_randn([I] * N, 20, dtype=torch.float32)
    pad = [torch.zeros((1 if k == 0 else 12, I, 1 if k == N - 1 else 12), dtype=torch.float32) for k in range(N)]
    full = oracle.tt_randn([I] * N, 64, dtype=torch.float32)
    items = [oracle.tt_add(g32, g32), full, oracle.tt_add(oracle.tt_add(g20, g20), oracle.tt_add(pad, pad))]
    items = [[c / c.abs().max() for c in it] for it in items]          # (comparable core scales inside one batch)
    assert all([tuple(c.shape) for c in it] == [tuple(c.shape) for c in items[0]] for it in items)
    inp = [torch.stack([it[k] for it in items]) for k in range(N)]
    t = gpu_tensor(inp, batch=True)
    t.round_tt(rmax=32)
    assert list(t.ranks_tt) == [1] + [32] * (N - 1) + [1]
    for i, it in enumerate(items):
        ours = to_list(t.cores, i)
        ref = oracle.round_tt([c.clone() for c in it], rmax=32, algorithm="svd")
        assert ranks(ours) == ranks(ref)
        e_ours, e_ref = tt_rel_err(ours, it), tt_rel_err(ref, it)
        assert abs(e_ours - e_ref) <= 2e-5 + 1e-3 * e_ref, (i, e_ours, e_ref)
        if i != 1:
            assert e_ours <= 2e-5 and tt_rel_err(ours, ref) <= 2e-5
        assert _right_orth_err(ours) <= 5e-5, (i, _right_orth_err(ours))
        so, sr = oracle.bond_singular_values(ours), oracle.bond_singular_values(ref)
        assert all(((x - y).abs().max() / y.max()).item() <= 2e-5 for x, y in zip(so, sr)), i


def test_reference_ranks_of_null_directions_are_the_default_and_the_switch_turns_them_off():
    """The eps-mode rank rule sees the null directions of a rank-deficient bond at LAPACK's noise level eps sigma_0
    (TTR_KNOB_RANK_NOISE_FLOOR = 1, the library's default since round 6) instead of the exact zeros of the zero-tail
    eigenproblem, so the NON-batch `round_tt(rmax=48)` (eps = 1e-14, tensor.py:2008-2014) of a numerically rank-32 fp32 train keeps
    the cap like the reference (round.py:147-158 on gesdd's output) and like batch mode; fp64 still cuts them (eps^2 < 1e-28), as
    LAPACK's do.  TTR_STRICT_RANKS=0 (knob = 0) cuts exact zeros: the numerical rank."""
    from tntorch_amd import _hip
    N, I = 5, 64
    torch.manual_seed(7)
    g = oracle.tt_randn([I] * N, 32, dtype=torch.float32)
    it = [c / c.abs().max() for c in oracle.tt_add(g, g)]
    ref = oracle.round_tt([c.clone() for c in it], rmax=48, algorithm="svd")
    t = gpu_tensor(it)
    t.round_tt(rmax=48)
    ours = to_list(t.cores)
    assert ranks(ours) == [1] + [48] * (N - 1) + [1]
    assert all(a == b for a, b in zip(ranks(ours), ranks(ref)) if b in (1, 48))   # (the oracle: 48 wherever LAPACK's noise is nonzero)
    assert all(a >= b for a, b in zip(ranks(ours), ranks(ref)))
    assert tt_rel_err(ours, it) <= 1e-5 and _right_orth_err(ours) <= 5e-5
    t64 = gpu_tensor([c.double() for c in it])
    t64.round_tt(rmax=48)
    r64 = oracle.round_tt([c.double() for c in it], rmax=48, algorithm="svd")
    assert ranks(to_list(t64.cores)) == ranks(r64) == [1] + [32] * (N - 1) + [1]
    # a full-rank train is untouched by the switch
    h = oracle.tt_randn([I] * 4, 40, dtype=torch.float32)
    a = gpu_tensor([c.clone() for c in h]); a.round_tt(eps=1e-3)
    _hip.set_knob(_hip.KNOB_RANK_NOISE_FLOOR, 0)
    try:
        b = gpu_tensor([c.clone() for c in h]); b.round_tt(eps=1e-3)
        # switched off: exact zeros are cut by the same rule at any delta >= 0 -- the numerical rank, the same tensor
        z = gpu_tensor(it)
        z.round_tt(rmax=48)
        assert ranks(to_list(z.cores)) == [1] + [32] * (N - 1) + [1]
        assert tt_rel_err(to_list(z.cores), it) <= 2e-5
    finally:
        _hip.set_knob(_hip.KNOB_RANK_NOISE_FLOOR, 1)
    assert ranks(to_list(a.cores)) == ranks(to_list(b.cores))
    assert all(torch.equal(x, y) for x, y in zip(a.cores, b.cores))


@pytest.mark.parametrize("batch", [True, False])
def test_rank_cap_above_the_numerical_rank_of_a_packed_train(batch):
    """rmax = 48 on a train of numerical rank 32 in 64 (g + g): every bond packs, its Gram matrix is solved as a 32 x 32 problem
    (V = blockdiag(V11, I), sigma[32:] = 0), and 16 of the 48 directions under the cap are null -- batch mode keeps them
    (round.py:149-150), and so does the non-batch call: the rank rule (round.py:147-158) sees them at LAPACK's noise level
    eps sigma_0, as the reference's gesdd returns them.  Batch and non-batch calls agree with each other and with the oracle."""
    N, I = 5, 64
    torch.manual_seed(7)
    g = oracle.tt_randn([I] * N, 32, dtype=torch.float32)
    it = [c / c.abs().max() for c in oracle.tt_add(g, g)]
    if batch:
        t = gpu_tensor([torch.stack([c, c]) for c in it], batch=True)
    else:
        t = gpu_tensor(it)
    t.round_tt(rmax=48)
    ours = to_list(t.cores, 1) if batch else to_list(t.cores)
    ref = oracle.round_tt([c.clone() for c in it], rmax=48, algorithm="svd")
    assert ranks(ours) == [1] + [48] * (N - 1) + [1]
    # the oracle's LAPACK returns 1e-7-level noise for the 16 null directions and keeps them: 48 wherever that noise is nonzero
    assert all(32 <= r <= 48 for r in ranks(ref)[1:-1])
    assert all(a == b for a, b in zip(ranks(ours), ranks(ref)) if b in (1, 48))
    assert tt_rel_err(ours, it) <= 2e-5
    assert _right_orth_err(ours) <= 5e-5


def test_metric_config_properties():
    """Size-independent properties at the metric's full size (batch of 8)."""
    inp = _metric_input(8, seed=1)
    t = gpu_tensor(inp, batch=True)
    t.round_tt(rmax=32)
    c = [x.cpu().double() for x in t.cores]
    for k in range(1, 8):  # cores 1..N-1 right-orthonormal
        Rm = c[k].reshape(8, c[k].shape[1], -1)
        err = (Rm @ Rm.transpose(1, 2) - torch.eye(Rm.shape[1], dtype=torch.float64)).abs().max().item()
        assert err <= 5e-5, (k, err)
    # all the norm sits in core 0: ||t|| = ||core_0||
    for i in range(8):
        n0 = c[0][i].norm().item()
        nt = math.sqrt(oracle.tt_dot([x[i] for x in c], [x[i] for x in c]).item())
        assert abs(n0 - nt) / nt <= 1e-5
    # idempotence: rounding the rounded train again changes nothing
    t2 = t.clone(); t2.round_tt(rmax=32)
    for i in range(8):
        assert tt_rel_err(to_list(t2.cores, i), to_list(t.cores, i)) <= 2e-5


def test_c2_config_vs_oracle():
    """BASELINE config C2: round_tt(eps=1e-4) of a rank-64 TT, 10 cores x mode 128, float64."""
    torch.manual_seed(0)
    g = oracle.tt_randn([128] * 10, 32, dtype=torch.float64)
    inp = oracle.tt_add(g, g)
    ref = oracle.round_tt(inp, eps=1e-4, algorithm="eig")
    for alg in ("svd", "eig"):
        t = gpu_tensor(inp)
        t.round_tt(eps=1e-4, algorithm=alg)
        ours = to_list(t.cores)
        assert ranks(ours) == ranks(ref) == [1] + [32] * 9 + [1]
        assert tt_rel_err(ours, ref) <= 1e-7
        so, sr = oracle.bond_singular_values(ours), oracle.bond_singular_values(ref)
        for a, b in zip(so, sr):
            assert ((a - b).abs().max() / b.max()).item() <= 1e-10


# ------------------------------------------------------------------ round 2: what is timed is what is tested
def test_metric_config_two_stream_path_vs_oracle():
    """The benchmark's path: 64^8 rank 64 -> 32 fp32 at a batch that runs as TWO sub-batches on two HIP streams writing
    into the shared result arena (B >= 128), default algorithm 'svd'; items from both sub-batches against the oracle's
    'svd' (LAPACK gesdd, round.py:96): ranks, bond singular values <= 1e-5 sigma_max, train <= 1e-5."""
    B = 130
    inp = _metric_input(B, seed=7)
    t = gpu_tensor(inp, batch=True)
    t.round_tt(rmax=32)
    assert list(t.ranks_tt) == [1] + [32] * 7 + [1]
    flat = tn.dist_batch.pack_cores(t.cores)
    assert flat.data_ptr() == t.cores[0].data_ptr()          # the arena layout (the gather's wire format) was used
    for i in (0, 64, 65, 129):                               # first / last item of either sub-batch
        ref = oracle.round_tt([c[i] for c in inp], rmax=32, algorithm="svd")
        ours = to_list(t.cores, i)
        assert ranks(ours) == ranks(ref)
        assert tt_rel_err(ours, ref) <= 1e-5
        assert tt_rel_err(ours, [c[i] for c in inp]) <= 1e-5
        so, sr = oracle.bond_singular_values(ours), oracle.bond_singular_values(ref)
        for a, b in zip(so, sr):
            assert ((a - b).abs().max() / b.max()).item() <= 1e-5


@pytest.mark.parametrize("alg", ["svd", "eig"])
@pytest.mark.parametrize("kind", ["randn", "lowrank"])
def test_c3_unit_dense_batch_vs_oracle(alg, kind):
    """BASELINE config C3's unit of work: dense 32^5 fp32 -> TT rmax 8, batched (8 tensors = 1.07 GB on the device),
    against the oracle's `_full_rank_tt` + `round_tt` (tensor.py:10-104, 401-408): ranks identical, approximation error
    agrees to 1e-5 absolute (randn: flat spectrum, error ~0.99; low rank + noise: error ~1e-3)."""
    B = 8 if kind == "randn" else 4
    torch.manual_seed(3)
    if kind == "randn":
        X = torch.randn(B, 32, 32, 32, 32, 32, dtype=torch.float32)
    else:
        X = torch.stack([oracle.tt_to_dense(oracle.tt_randn([32] * 5, 8, dtype=torch.float32)) for _ in range(B)])
        X = X / X.reshape(B, -1).norm(dim=1).reshape(B, 1, 1, 1, 1, 1) * math.sqrt(32 ** 5)
        X = X + 1e-3 * torch.randn(X.shape, dtype=torch.float32)
    t = tn.Tensor(X.cuda(), ranks_tt=8, batch=True, algorithm=alg)
    assert t.ranks_tt.tolist() == [1, 8, 8, 8, 8, 1]
    rec = t.torch().cpu()
    for i in range(0, B, 3):
        ref = oracle.dense_to_tt(X[i], 8, algorithm=alg)
        assert ranks(ref) == [1, 8, 8, 8, 8, 1]
        e_o = rel_diff(rec[i], X[i])
        e_r = rel_diff(dense(ref).float(), X[i])
        assert abs(e_o - e_r) <= 1e-5, (i, e_o, e_r)


def _decaying_tt(shape, R, decay, dtype, seed, batch=None):
    """Train whose bond singular values fall off like 2^(-decay * j) (SURVEY 8d's second input variant): i.i.d. normal
    cores of unit column variance with rank index j of every bond scaled by 2^(-decay * j)."""
    g = torch.Generator().manual_seed(seed)
    N = len(shape)
    r = [1] + [R] * (N - 1) + [1]
    lead = () if batch is None else (batch,)
    cores = []
    for k in range(N):
        c = torch.randn(lead + (r[k], shape[k], r[k + 1]), generator=g, dtype=torch.float64) / math.sqrt(r[k] * shape[k])
        if k < N - 1:
            c = c * (2.0 ** (-decay * torch.arange(r[k + 1], dtype=torch.float64)))
        cores.append(c.to(dtype))
    return cores


def _right_orth_err(cores):
    err = 0.0
    for c in cores[1:]:
        Rm = c.double().reshape(c.shape[0], -1)
        err = max(err, (Rm @ Rm.T - torch.eye(Rm.shape[0], dtype=torch.float64)).abs().max().item())
    return err


# Tolerances held against the float64 LAPACK oracle on decaying spectra (stated in DESIGN.md section 5):
#   bond singular values: absolute error <= 4e-6 sigma_max (fp32) / 1e-13 (fp64) -- the class of LAPACK gesdd;
#   right-orthonormality of the produced cores: 5e-5 (fp32) / 1e-11 (fp64) for EVERY kept direction, including the
#   numerically null ones (sigma below k eps sigma_max), which get an orthonormal completion;
#   approximation error: within 4e-6 + 1e-3 relative (fp32) / 1e-12 (fp64) of the oracle's.
@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("decay", [1.0, 0.25])
@pytest.mark.parametrize("rmax", [12, 32])
def test_decaying_spectrum_rmax(dt, decay, rmax):
    """Cores scaled so that bond sigma_j ~ 2^(-decay j); rmax = 12 keeps live directions only, rmax = 32 (= R, nothing
    is cut) also keeps directions far below the resolution of the input (2^-31 at decay 1)."""
    f32 = dt == torch.float32
    inp = _decaying_tt([12, 16, 16, 16, 12], 32, decay, dt, seed=int(decay * 100) + rmax)
    X = dense(inp)
    ref = oracle.round_tt([c.double() for c in inp], rmax=rmax, algorithm="svd")
    t = gpu_tensor(inp)
    t.round_tt(rmax=rmax)
    ours = to_list(t.cores)
    assert ranks(ours) == ranks(ref)
    e_o, e_r = rel_diff(dense(ours), X), rel_diff(dense(ref), X)
    assert abs(e_o - e_r) <= (4e-6 + 1e-3 * e_r if f32 else 1e-12 + 1e-6 * e_r), (e_o, e_r)
    assert _right_orth_err(ours) <= (5e-5 if f32 else 1e-11), _right_orth_err(ours)
    so, sr = oracle.bond_singular_values(ours), oracle.bond_singular_values(ref)
    for a, b in zip(so, sr):
        assert ((a - b).abs().max() / b.max()).item() <= (4e-6 if f32 else 1e-13)
    # batch mode (per-item exponents, no rank readback): same answers item by item
    inpb = [torch.stack([c, 2 * c, -0.5 * c]) if k == 0 else torch.stack([c, c, c]) for k, c in enumerate(inp)]
    tb = gpu_tensor(inpb, batch=True)
    tb.round_tt(rmax=rmax)
    for i, sc in enumerate((1.0, 2.0, -0.5)):
        ob = to_list(tb.cores, i)
        assert ranks(ob) == ranks(ref)
        assert abs(rel_diff(dense(ob), sc * X) - e_r) <= (4e-6 + 1e-3 * e_r if f32 else 1e-12 + 1e-6 * e_r)
        assert _right_orth_err(ob) <= (5e-5 if f32 else 1e-11)


@pytest.mark.parametrize("dt,eps", [(torch.float32, 1e-3), (torch.float32, 1e-5), (torch.float64, 1e-6), (torch.float64, 1e-11)])
@pytest.mark.parametrize("decay", [1.0, 0.25])
def test_decaying_spectrum_eps(dt, eps, decay):
    """eps mode on decaying spectra: the bound is respected, ranks follow the float64 oracle (a singular value within
    the rounding level of the cut may fall on either side: +-1 per bond), cores stay right-orthonormal."""
    f32 = dt == torch.float32
    inp = _decaying_tt([12, 16, 16, 16, 12], 32, decay, dt, seed=77)
    X = dense(inp)
    ref = oracle.round_tt([c.double() for c in inp], eps=eps, algorithm="svd")
    t = gpu_tensor(inp)
    t.round_tt(eps=eps)
    ours = to_list(t.cores)
    assert all(abs(a - b) <= 1 for a, b in zip(ranks(ours), ranks(ref))), (ranks(ours), ranks(ref))
    assert rel_diff(dense(ours), X) <= eps * (1 + 1e-3) + (2e-6 if f32 else 1e-13)
    assert _right_orth_err(ours) <= (5e-5 if f32 else 1e-11)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("decay", [0.5, 1.0, -1.0])
def test_decaying_spectrum_metric_shape(dt, decay):
    """The decaying-spectrum variant at (a slice of) the metric's shape: cores [64, 64, 64], 64^5, rank 64 -> 32,
    sigma_j ~ 2^(-j/2) (sigma_31 / sigma_0 = 2e-5, sigma_63 / sigma_0 = 3e-10), batch of 2, vs the float64 oracle.
    decay 1 ("sigma ~ 2^-j"; sigma_63 / sigma_0 = 1e-19): the columns of a 64-column unfolding span 63 binary
    orders -- squared norms of the last ones are fp32 denormals, which the hardware sqrt / rcp of the Householder step flush
    (round 4: NaN, then rank-1 zeros by the zero guard; now H = I for sub-columns below 2^-50 of their block), and 15 of the 32
    kept directions of every bond lie below the resolution of the input (orthonormal completion)."""
    f32 = dt == torch.float32
    if decay < 0:
        # decay -1 = decay 1 with ttr_orth_fixup's three-launch rounds (what batches from TTR_KNOB_ORTH_SPLIT = 2048 items per
        # stream take by default), forced on this batch of 2: the same bounds; through ttr_round_tt and through the host loop
        from tntorch_amd import _hip, _hipops
        decay = 1.0
        _hip.set_knob(_hip.KNOB_ORTH_SPLIT, 1)
        inp = _decaying_tt([64] * 5, 64, decay, dt, seed=5, batch=2)
        _hipops.SWEEP_C_ENABLED = False
        th = gpu_tensor(inp, batch=True)
        th.round_tt(rmax=32)
        _hipops.SWEEP_C_ENABLED = True
        t = gpu_tensor(inp, batch=True)
        t.round_tt(rmax=32)
        assert all(torch.equal(a, b) for a, b in zip(t.cores, th.cores))
    else:
        inp = _decaying_tt([64] * 5, 64, decay, dt, seed=5, batch=2)
        t = gpu_tensor(inp, batch=True)
        t.round_tt(rmax=32)
    for i in range(2):
        one = [c[i] for c in inp]
        ref = oracle.round_tt([c.double() for c in one], rmax=32, algorithm="svd")
        ours = to_list(t.cores, i)
        assert ranks(ours) == ranks(ref)
        e_o, e_r = tt_rel_err(ours, one), tt_rel_err(ref, one)
        assert abs(e_o - e_r) <= ((4e-6 if decay == 0.5 else 1e-5) + 1e-2 * e_r if f32 else 1e-8), (e_o, e_r)   # (tt_rel_err itself resolves ~1e-8)
        assert _right_orth_err(ours) <= (5e-5 if f32 else 1e-11)
        so, sr = oracle.bond_singular_values(ours), oracle.bond_singular_values(ref)
        for a, b in zip(so, sr):
            assert ((a - b).abs().max() / b.max()).item() <= (4e-6 if f32 else 1e-13)


def test_orth_fixup_split_rounds_at_their_real_batch_size_vs_oracle(monkeypatch):
    """ttr_orth_fixup's three-launch rounds are what a launch of >= TTR_KNOB_ORTH_SPLIT = 2048 items takes by default (smaller
    launches: the single-workgroup kernel -- the same train may round to different BITS in the directions below the input's
    resolution depending on the batch it travels in).  Here at the real size: 2048 trains of 64^4, rank 64 -> 32,
    bond sigma_j ~ 2^-j (15 of the 32 kept directions of every bond are dead), one launch per kernel, against the float64 oracle
    on the first, a middle and the last item -- the bounds of test_decaying_spectrum_metric_shape."""
    from tntorch_amd import _hipops
    monkeypatch.setattr(_hipops, "STREAM_CHUNKS_ENABLED", False)   # (one sub-batch: 2048 items per launch)
    B, N, I, R = 2048, 4, 64, 64
    gen = torch.Generator(device="cuda").manual_seed(11)
    r = [1] + [R] * (N - 1) + [1]
    inp = []
    for k in range(N):
        c = torch.randn((B, r[k], I, r[k + 1]), generator=gen, device="cuda", dtype=torch.float32) / math.sqrt(r[k] * I)
        if k < N - 1:
            c = c * (2.0 ** (-1.0 * torch.arange(r[k + 1], device="cuda", dtype=torch.float32)))
        inp.append(c)
    t = tn.Tensor(inp, batch=True)
    t.round_tt(rmax=32)
    assert list(t.ranks_tt) == [1, 32, 32, 32, 1]
    for i in (0, B // 2 + 1, B - 1):
        one = [c[i].cpu() for c in inp]
        ref = oracle.round_tt([c.double() for c in one], rmax=32, algorithm="svd")
        ours = to_list(t.cores, i)
        assert ranks(ours) == ranks(ref)
        e_o, e_r = tt_rel_err(ours, one), tt_rel_err(ref, one)
        assert abs(e_o - e_r) <= 1e-5 + 1e-2 * e_r, (i, e_o, e_r)
        assert _right_orth_err(ours) <= 5e-5, (i, _right_orth_err(ours))
        so, sr = oracle.bond_singular_values(ours), oracle.bond_singular_values(ref)
        for a, b in zip(so, sr):
            assert ((a - b).abs().max() / b.max()).item() <= 4e-6


def test_null_directions_kept_are_orthonormal():
    """g + g (exactly rank deficient) rounded with a cap ABOVE the true rank: the reference keeps the null directions
    with noise-level sigma and orthonormal rows (LAPACK); so do we (ttr_orth_fixup), and the train is unchanged."""
    for dt in (torch.float32, torch.float64):
        torch.manual_seed(9)
        g = oracle.tt_randn([10, 12, 12, 10], 6, dtype=dt)
        inp = oracle.tt_add(g, g)
        t = gpu_tensor(inp)
        t.round_tt(rmax=9, eps=0.0)
        ours = to_list(t.cores)
        assert ranks(ours) == [1, 9, 9, 9, 1]
        assert _right_orth_err(ours) <= (5e-5 if dt == torch.float32 else 1e-11)
        assert rel_diff(dense(ours), dense(inp)) <= (5e-6 if dt == torch.float32 else 1e-13)


def test_non_current_device():
    """Tensors on a device that is not the current one are processed on their own device's stream."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    torch.manual_seed(0)
    g = oracle.tt_randn([8, 9, 7, 8], 5, dtype=torch.float32)
    inp = oracle.tt_add(g, g)
    torch.cuda.set_device(0)
    t = tn.Tensor([c.to("cuda:1") for c in inp])
    t.round_tt(rmax=5)
    assert all(c.device.index == 1 for c in t.cores)
    assert rel_diff(dense(to_list(t.cores)), dense(inp)) <= 5e-6


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_fused_add_round_matches_materialised(dt):
    """tools.reduce(operator.add) on device tensors rounds every sum WITHOUT materialising its block-diagonal cores
    (ttr_qr_factor_pushed_sum): same tensor as `round(a + b)` on the padded cores and as the oracle's reduction tree,
    in eps mode (with the Tucker stage of round()), rmax mode and batch mode."""
    import operator
    f32 = dt == torch.float32
    torch.manual_seed(21)
    parts = [oracle.tt_randn([9, 10, 8, 9, 7], 3, dtype=dt) for _ in range(6)]
    ts = [gpu_tensor(p) for p in parts]
    want = sum(dense(p) for p in parts)
    red = tn.reduce(ts, operator.add, eps=1e-6 if f32 else 1e-10)
    assert rel_diff(red.torch().cpu(), want) <= (5e-5 if f32 else 1e-9)
    ref_cores, _ = oracle.reduce_sum(parts, eps=1e-6 if f32 else 1e-10)
    assert sum(red.ranks_tt.tolist()) <= sum(ranks(ref_cores)) + 2
    red3 = tn.reduce(ts, operator.add, rmax=4)
    mat = tn.round_tt(ts[0] + ts[1], rmax=4)                                   # the generic path on the padded cores
    fus = tn.Tensor._round_of_sum(ts[0], ts[1], eps=1e-14, rmax=4)
    assert fus is not None and fus.ranks_tt.tolist() == mat.ranks_tt.tolist() == [1, 4, 4, 4, 4, 1]
    assert rel_diff(fus.torch().cpu(), mat.torch().cpu()) <= (2e-5 if f32 else 1e-10)
    assert red3.ranks_tt.tolist() == [1, 4, 4, 4, 4, 1]
    # eps AND rmax together: the Tucker stage of round() gets the same rmax on the fused path as on the generic one
    # (round-2 advisor finding: the fused path used to drop it)
    e3 = 1e-2
    gen = tn.round(ts[0] + ts[1], eps=e3, rmax=3)
    fus = tn.Tensor._round_of_sum(ts[0], ts[1], eps=e3, rmax=3)
    assert fus.ranks_tt.tolist() == gen.ranks_tt.tolist() and max(fus.ranks_tt.tolist()) <= 3
    assert [None if u is None else tuple(u.shape) for u in fus.Us] == [None if u is None else tuple(u.shape) for u in gen.Us]
    assert all(u is None or u.shape[1] <= 3 for u in fus.Us)
    assert rel_diff(fus.torch().cpu(), gen.torch().cpu()) <= (5e-5 if f32 else 1e-9)
    red4 = tn.reduce(ts, operator.add, eps=e3, rmax=3)
    assert max(red4.ranks_tt.tolist()) <= 3 and all(u is None or u.shape[1] <= 3 for u in red4.Us)
    # batch mode, two-stream sized batch, unequal ranks of the addends
    a = oracle.tt_randn([6, 7, 5, 6], 3, dtype=dt, batch_size=130)
    b = oracle.tt_randn([6, 7, 5, 6], 5, dtype=dt, batch_size=130)
    ta, tb = gpu_tensor(a, batch=True), gpu_tensor(b, batch=True)
    fus = tn.Tensor._round_of_sum(ta, tb, rmax=8)
    X = oracle.tt_to_dense([c.double() for c in a], batch=True) + oracle.tt_to_dense([c.double() for c in b], batch=True)
    assert fus.ranks_tt.tolist() == [1, 6, 8, 6, 1]
    assert rel_diff(fus.torch().cpu(), X) <= (5e-6 if f32 else 1e-12)            # rank 8 = 3 + 5: exact
    # ranks above the fused kernel's 64 columns fall back to the padded core
    a = oracle.tt_randn([4, 5, 4], 40, dtype=dt)
    fb = tn.Tensor._round_of_sum(gpu_tensor(a), gpu_tensor(a), eps=1e-6)
    assert rel_diff(fb.torch().cpu(), 2 * dense(a)) <= (5e-5 if f32 else 1e-9)


# ------------------------------------------------------------------ consumers: shift_mode, TTMatrix
SHIFT_SPECS = [(1, 2, 1e-3), (3, -2, 1e-6), (0, 4, "same"), (4, -4, 1e-2), (2, 1, 0.3)]


@pytest.mark.gpu
def test_golden_shift_mode_on_device():
    g = load_case("consumers_f64")
    src = oracle.tt_to_dense(g["g"])
    for k, (n, sh, eps) in enumerate(SHIFT_SPECS):
        t = gpu_tensor(g["g"])
        r = tn.shift_mode(t, n, sh, eps=eps)
        assert r is t and all(c.is_cuda for c in t.cores)
        want = g[f"shift{k}"]
        assert t.ranks_tt.tolist() == oracle.tt_ranks(want), (k, t.ranks_tt.tolist(), oracle.tt_ranks(want))
        assert rel_diff(t.torch().cpu(), oracle.tt_to_dense(want)) <= 1e-10, k
    # float32, larger modes: against the oracle on the same input, and the moved tensor against the permuted source
    torch.manual_seed(41)
    cores = oracle.tt_randn([12, 9, 14, 10], [5, 6, 4], dtype=torch.float32)
    t = gpu_tensor(cores)
    tn.shift_mode(t, 0, 3, eps=1e-4)
    want = oracle.shift_mode(cores, 0, 3, eps=1e-4)
    assert t.ranks_tt.tolist() == oracle.tt_ranks(want)
    assert rel_diff(t.torch().cpu(), oracle.tt_to_dense([c.double() for c in want])) <= 2e-5
    assert rel_diff(t.torch().cpu(), oracle.tt_to_dense([c.double() for c in cores]).permute(1, 2, 3, 0)) <= 2e-4
    assert src is not None


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_golden_ttmatrix_on_device(alg):
    g = load_case("consumers_f64")
    ttm = tn.TTMatrix(g["m"].cuda(), input_dims=[11, 3, 4], output_dims=[23, 2, 3], ranks=[20, 7])
    assert ttm.ranks.tolist() == [20, 7] and all(c.is_cuda for c in ttm.cores)
    assert [tuple(c.shape) for c in ttm.cores] == [tuple(c.shape) for c in g["ttm_cores"]]
    assert rel_diff(ttm.torch().cpu(), g["ttm_dense"]) <= 1e-10
    tsq = tn.TTMatrix(g["sq"].cuda(), input_dims=[6, 5], output_dims=[6, 5], ranks=[36])
    assert abs(tsq.trace().item() - g["tsq_trace"].item()) <= 1e-10
    # truncating ranks, float32, batch: against the oracle per item
    torch.manual_seed(43)
    mb = torch.rand(3, 16 * 12, 10 * 8, dtype=torch.float32)
    b = tn.TTMatrix(mb.cuda(), input_dims=[16, 12], output_dims=[10, 8], ranks=[9])
    assert b.batch and b.ranks.tolist() == [9]
    for k in range(3):
        want = oracle.ttmatrix_to_dense(oracle.ttmatrix_cores(mb[k], [9], [16, 12], [10, 8], algorithm=alg))
        got = b.torch()[k].cpu()
        assert abs(rel_diff(got, mb[k]) - rel_diff(want, mb[k])) <= 1e-5  # same approximation error
        assert rel_diff(got, want) <= 5e-4  # flat random spectrum: the rank-9 subspace itself is only defined to ~gap^-1 eps


@pytest.mark.gpu
def test_golden_cp_variants_on_device():
    """Batched CP-ALS (lock-step items, batch-mean convergence) and CP on a Tucker core on the device."""
    g = load_case("cp_variants_f64")
    tb = tn.Tensor(g["batch_inp"].cuda(), ranks_cp=4, batch=True, max_iter=6, tol=-1.0)
    assert tb.batch and all(c.is_cuda for c in tb.cores) and len(tb.cp_errors) == 6
    want = g["batch_dense"]
    got = tb.torch().cpu()
    # the same ALS sweeps from the same HOSVD start (eigenvector signs cancel in the CP reconstruction)
    assert rel_diff(got, want) <= 1e-6
    errs = [(got[i] - g["batch_inp"][i]).norm() / g["batch_inp"][i].norm() for i in range(3)]
    assert abs(torch.stack(errs).mean().item() - tb.cp_errors[-1]) <= 1e-6
    # CP on a Tucker core.  The ALS itself, deterministically: the reference's Tucker core and its recorded random start
    from tntorch_amd import _hipops
    X = g["tucker_inp"]
    core = torch.einsum("ijk,ia,jb,kc->abc", X, *g["tucker_Us"])
    fac, errs = _hipops.cp_als(core.cuda(), 3, 5, -1.0, init=[f.cuda() for f in g["tucker_init"]])
    assert len(errs) == 5 and all(b <= a + 1e-12 for a, b in zip(errs, errs[1:]))
    assert rel_diff(oracle.cp_to_dense([f.cpu() for f in fac]), oracle.cp_to_dense(g["tucker_cores"])) <= 1e-6
    # ... and through the constructor (the device draws its own start; ALS from a random start can stall in a swamp, on
    # the CPU just as well, so only structure and monotonicity are asserted)
    torch.manual_seed(0)
    tt = tn.Tensor(X.cuda(), ranks_cp=3, ranks_tucker=4, max_iter=25)
    assert [tuple(c.shape) for c in tt.cores] == [(4, 3)] * 3 and [tuple(U.shape) for U in tt.Us] == [(9, 4), (8, 4), (7, 4)]
    assert all(c.is_cuda for c in tt.cores) and all(U.is_cuda for U in tt.Us)
    assert all(b <= a + 1e-9 for a, b in zip(tt.cp_errors, tt.cp_errors[1:])) and tt.cp_errors[-1] < 0.6
    assert rel_diff(tt.torch().cpu(), X) < 0.6
    Xb = torch.stack([X, X.flip(0)])
    tbt = tn.Tensor(Xb.cuda(), ranks_cp=3, ranks_tucker=[4, 4, 4], batch=True, max_iter=8)
    assert tbt.batch and [tuple(c.shape) for c in tbt.cores] == [(2, 4, 3)] * 3 and rel_diff(tbt.torch().cpu(), Xb) < 0.7


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_dense_tt_svd_inplace_carry_rotation(dt, monkeypatch):
    """Config-scale carries (> 4 GiB) are rotated in place, chunk by chunk (no second tensor of the carry's size): forced
    here at a small size, the result must equal the out-of-place path and the oracle."""
    from tntorch_amd import _hipops
    torch.manual_seed(47)
    X = torch.randn(14, 12, 9, 11, dtype=dt)
    ref = tn.Tensor(X.cuda(), ranks_tt=8)
    monkeypatch.setattr(_hipops, "_INPLACE_ROTATE_BYTES", 0)
    monkeypatch.setattr(_hipops, "_INPLACE_CHUNK_BYTES", 4096)
    Xd = X.cuda()
    keep = Xd.clone()
    got = tn.Tensor(Xd, ranks_tt=8)
    assert torch.equal(Xd, keep)  # the user's tensor is never the scratch
    assert got.ranks_tt.tolist() == ref.ranks_tt.tolist() == [1, 8, 8, 8, 1]
    tol = 1e-5 if dt == torch.float32 else 1e-12
    assert rel_diff(got.torch().cpu(), ref.torch().cpu()) <= tol
    want = oracle.tt_to_dense([c.double() for c in oracle.dense_to_tt(X, 8)])
    assert abs(rel_diff(got.torch().cpu(), X) - rel_diff(want, X)) <= (1e-5 if dt == torch.float32 else 1e-10)


# ------------------------------------------------------------------ round 3: the parity holes the round-2 review named
@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_c1_proxy_dense_64_4_vs_oracle(alg):
    """BASELINE config C1 at the largest size the reference's algorithm can still run (SURVEY section 6 "C1 proxy"): dense
    64^4 fp32 -> ranks_tt = 16.  The middle bond is n = I r = 1024: the block-Jacobi driver at C1's own bond size (the
    largest bond tested before was 512).  Oracle = `_full_rank_tt` + `round_tt` (tensor.py:10-104, 401-408) in the same
    precision: ranks identical, approximation error agrees to 1e-5 absolute, cores 1.. right-orthonormal."""
    torch.manual_seed(11)
    shape, r = [64] * 4, 16
    low = oracle.tt_to_dense(oracle.tt_randn(shape, r, dtype=torch.float64))
    X = (low / low.norm() * math.sqrt(low.numel()) + 1e-3 * torch.randn(shape, dtype=torch.float64)).float()
    ref = oracle.dense_to_tt(X, r, algorithm=alg)
    t = tn.Tensor(X.cuda(), ranks_tt=r, algorithm=alg)
    assert list(t.ranks_tt) == ranks(ref) == [1, 16, 16, 16, 1]
    e_o = rel_diff(t.torch().cpu().double(), X.double())
    e_r = rel_diff(oracle.tt_to_dense([c.double() for c in ref]), X.double())
    assert abs(e_o - e_r) <= 1e-5, (e_o, e_r)
    assert tt_rel_err(to_list(t.cores), ref) <= 3e-5      # separated spectrum (rank 16 + 1e-3 noise): the trains agree
    for c in t.cores[1:]:
        M = c.reshape(c.shape[0], -1).double().cpu()
        assert (M @ M.T - torch.eye(M.shape[0], dtype=torch.float64)).abs().max() < 3e-5


@pytest.mark.parametrize("kind", ["lowrank", "randn"])
def test_c4_proxy_cp_als_r32_vs_oracle(kind):
    """BASELINE config C4's path at a size the oracle finishes in seconds: CP-ALS R = 32 on a dense 48^4 fp32 tensor, a
    fixed 5 sweeps (tol < 0 never stops, tensor.py:380-381).  R = 32 takes the 128x32-tile / split-K GEMMs and
    `ttr_krp_contract` with the shapes of the 256^4 config (R <= 5 in the golden cases does not).  Against
    `oracle.cp_als` (tensor.py:210-400 restated) run in FLOAT64 on the same fp32 data: the error after EVERY sweep agrees to
    3e-4 absolute on the low-rank + noise input, 1e-5 on `randn` (ours: the algebraic estimate of the sweep; oracle: dense
    reconstruction), and so does the true error of the final factors.  Why fp64 as the yardstick: the input's
    Hadamard-of-Grams matrices reach condition 2e6 while the error still falls by ~0.1 per sweep, and the reference's own
    fp32 trajectory (lstsq) then depends on the host -- 0.316288 in the build container, 0.316509 on the GPU box's CPU after
    five sweeps, against 0.316341 in fp64 (measured); the device solves the R x R systems in fp64 for that reason."""
    torch.manual_seed(17)
    shape, R = [48] * 4, 32
    if kind == "lowrank":
        fac = [torch.randn(i, R, dtype=torch.float64) for i in shape]
        X = oracle.cp_to_dense(fac)
        X = (X / X.norm() * math.sqrt(X.numel()) + 1e-2 * torch.randn(shape, dtype=torch.float64)).float()
    else:
        X = torch.randn(shape, dtype=torch.float32)
    ref_cores, ref_err = oracle.cp_als(X.double(), R, max_iter=5, tol=-1.0)
    t = tn.Tensor(X.cuda(), ranks_cp=R, max_iter=5, tol=-1.0)
    assert len(t.cp_errors) == len(ref_err) == 5
    tol = 3e-4 if kind == "lowrank" else 1e-5
    for k, (a, b) in enumerate(zip(t.cp_errors, ref_err)):
        assert abs(float(a) - float(b)) <= tol, (kind, k, float(a), float(b))
    e_o = rel_diff(oracle.cp_to_dense([c.cpu().double() for c in t.cores]), X.double())
    e_r = rel_diff(oracle.cp_to_dense([c.double() for c in ref_cores]), X.double())
    assert abs(e_o - e_r) <= tol and abs(e_o - float(t.cp_errors[-1])) <= 1e-4, (kind, e_o, e_r, float(t.cp_errors[-1]))


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("B", [3, 130])   # 130: the two-stream path with the result arena
def test_all_zero_batch_matches_reference(dt, B):
    """round.py:137-141 in batch mode: when the largest singular value of the WHOLE batch lies below 1e-13 the bond is
    replaced by rank-1 zeros -- and so is every further bond, the carry being zero.  The reference's answer (pinned through
    the oracle): all ranks 1, all cores zero, whatever `rmax`."""
    shape, r = [6, 5, 7, 4], 5
    inp = [torch.zeros(B, 1 if k == 0 else r, n, 1 if k == len(shape) - 1 else r, dtype=dt) for k, n in enumerate(shape)]
    ref = oracle.round_tt([c.clone() for c in inp], rmax=3, batch=True)
    assert [c.shape[-1] for c in ref] == [1, 1, 1, 1] and all((c == 0).all() for c in ref)
    t = gpu_tensor(inp, batch=True)
    t.round_tt(rmax=3)
    assert [tuple(c.shape) for c in t.cores] == [tuple(c.shape) for c in ref]
    assert all((c == 0).all().item() and c.is_cuda and c.dtype == dt for c in t.cores)
    # truncated_svd(batch=True) on an all-zero batch: rank-1 zeros as well (round.py:138-141)
    l, m2 = tn.truncated_svd(torch.zeros(B, 8, 20, dtype=dt).cuda(), eps=1e-3, rmax=4, batch=True)
    assert tuple(l.shape) == (B, 8, 1) and tuple(m2.shape) == (B, 1, 20) and (l == 0).all() and (m2 == 0).all()


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_zero_item_inside_batch(dt):
    """ONE all-zero item in a batch whose other items are not: the reference's guard looks at the batch maximum
    (round.py:138), so nothing is replaced and the zero item is divided by its sigma = 0 -- `(1/0) * 0` = NaN in every core
    right of core 0 (pinned below through the oracle).  Conscious deviation: the device kernels
    return 0 for a division by an exactly-zero sigma, i.e. the zero item comes back as a finite train that represents
    the zero tensor, with the batch's ranks; the other items are untouched by their neighbour."""
    torch.manual_seed(2)
    B, shape, r = 4, [6, 5, 7, 4], 4
    g = oracle.tt_randn(shape, r, dtype=dt, batch_size=B)
    inp = oracle.tt_add(g, g, batch=True)
    for c in inp:
        c[2] = 0
    ref = oracle.round_tt([c.clone() for c in inp], rmax=r, batch=True)
    assert oracle.tt_ranks([c[0] for c in ref]) == [1, r, r, r, 1]
    assert all(torch.isnan(c[2]).any() for c in ref[1:]) and (ref[0][2] == 0).all()     # the reference's answer for the zero item
    t = gpu_tensor(inp, batch=True)
    t.round_tt(rmax=r)
    ours = to_list(t.cores)
    assert [tuple(c.shape) for c in ours] == [tuple(c.shape) for c in ref]
    assert all(torch.isfinite(c).all() for c in ours)
    assert oracle.tt_to_dense([c[2].double() for c in ours]).abs().max() == 0           # zero item: the zero tensor, finite
    tol = 2e-5 if dt == torch.float32 else 1e-10
    for i in (0, 1, 3):  # (dense comparison: the inner-product formula of tt_rel_err resolves 1e-8 at best)
        assert rel_diff(dense([c[i] for c in ours]), dense([c[i] for c in ref])) <= tol


@pytest.mark.parametrize("dt,eps", [(torch.float32, 1e-3), (torch.float64, 1e-6), (torch.float64, 1e-14)])
@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_eps_mode_without_rank_readbacks(dt, eps, alg, monkeypatch):
    """at most ONE host synchronisation per round_tt in eps mode.  The deferred sweep (rank rule bound on the
    device, every bond computed at its rank cap and masked there, one readback of all ranks at the end) against the
    round-2 sweep that reads one rank back per bond: identical ranks and the same train, on a decaying-spectrum input
    (data-dependent ranks that differ from bond to bond), with and without an rmax cap, and on an all-zero train."""
    inp = _decaying_tt([12, 10, 9, 11, 8, 16], 14, 0.6, dt, seed=3)   # (every bond: rows <= columns, the fused kernels' side)
    from tntorch_amd import _hipops
    monkeypatch.setattr(_hipops, "SWEEP_C_ENABLED", False)   # (the host loop's own two eps-mode sweeps; ttr_round_tt: next test)
    calls = []
    orig = _hipops._eps_deferred_ok
    monkeypatch.setattr(_hipops, "_eps_deferred_ok", lambda *a: calls.append(orig(*a)) or calls[-1])
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TTR_EPS_DEFERRED", mode)
        res = []
        for rmax in (None, 6):
            t = gpu_tensor(inp)
            t.round_tt(eps=eps, rmax=rmax, algorithm=alg)
            res.append(to_list(t.cores))
        z = gpu_tensor([torch.zeros_like(c) for c in inp])
        z.round_tt(eps=eps, algorithm=alg)
        res.append(to_list(z.cores))
        out[mode] = res
    assert calls == [True] * 3 + [False] * 3                                             # both sweeps really ran
    for a, b in zip(out["1"], out["0"]):
        assert ranks(a) == ranks(b)
        assert all(x.is_contiguous() for x in a)
    assert len(set(ranks(out["1"][0])[1:-1])) > 1 and max(ranks(out["1"][1])) <= 6      # data-dependent ranks, capped ranks
    assert ranks(out["1"][2]) == [1] * 7 and all((c == 0).all() for c in out["1"][2])   # zero guard: rank-1 zeros
    tol = 2e-5 if dt == torch.float32 else 1e-10
    for a, b in zip(out["1"][:2], out["0"][:2]):
        assert rel_diff(dense(a), dense(b)) <= tol
    ref = oracle.round_tt([c.clone() for c in inp], eps=eps, algorithm=alg)
    if alg == "svd" or eps > 1e-10:   # ('eig' at eps = 1e-14 sits on the Gram noise floor: ranks not asserted, tests/test_round.py:52-59)
        assert all(abs(x - y) <= 1 for x, y in zip(ranks(out["1"][0]), ranks(ref)))
    assert rel_diff(dense(out["1"][0]), dense(ref)) <= max(3 * eps, tol)


def _sweep_both_ways(monkeypatch, make, call):
    """Run ``call(tensor)`` on ``make()`` through ttr_round_tt and through the host loop over the per-kernel entries; returns the
    two lists of cores and how many sweeps went through the one-call entry."""
    from tntorch_amd import _hipops
    out = {}
    n0 = _hipops.SWEEP_C_CALLS
    for on in (True, False):
        monkeypatch.setattr(_hipops, "SWEEP_C_ENABLED", on)
        t = make()
        call(t)
        out[on] = [c.clone() for c in t.cores]
        if on:
            used = _hipops.SWEEP_C_CALLS - n0
    return out[True], out[False], used


@pytest.mark.parametrize("alg", ["svd", "eig"])
@pytest.mark.parametrize("case", ["metric3", "metric130", "decay", "small_f64", "two_cores", "lead_rank", "c2_f64"])
def test_whole_sweep_entry_is_bit_identical_to_the_host_loop_batch(case, alg, monkeypatch):
    """ttr_round_tt (SURVEY 8b's `tt_round_sweep_batched_*`: tensor.py:1905-1906 + 2053-2083 behind ONE library call) enqueues
    the same kernels in the same order as the host loop of `_round_tt_sweep`: the rounded cores are BIT-identical -- on the
    metric's shape (one sub-batch, and two sub-batches on two streams writing the result arena), a decaying spectrum (no
    shortcut fires), fp64, a two-core train, a train with a leading rank > 1, and the shape of BASELINE config C2."""
    if case == "metric3":
        inp, rmax = _metric_input(3, seed=11), 32
    elif case == "metric130":
        inp, rmax = _metric_input(130, seed=12), 32
    elif case == "decay":
        inp, rmax = [torch.stack([a, b]) for a, b in zip(_decaying_tt([64] * 5, 64, 0.5, torch.float32, seed=5),
                                                           _decaying_tt([64] * 5, 64, 1.0, torch.float32, seed=6))], 32
    elif case == "small_f64":
        inp, rmax = [torch.stack([a, b]) for a, b in zip(_decaying_tt([12, 10, 9, 11, 8, 16], 14, 0.6, torch.float64, seed=3),
                                                           _decaying_tt([12, 10, 9, 11, 8, 16], 14, 0.3, torch.float64, seed=4))], 5
    elif case == "two_cores":
        torch.manual_seed(3)
        inp, rmax = [torch.randn(4, 1, 40, 24), torch.randn(4, 24, 48, 1)], 7
    elif case == "lead_rank":
        torch.manual_seed(4)
        inp, rmax = [torch.randn(3, 3, 20, 16), torch.randn(3, 16, 12, 16), torch.randn(3, 16, 9, 2)], 6
    else:
        torch.manual_seed(5)
        g = oracle.tt_randn([128] * 10, 32, dtype=torch.float64, batch_size=2)
        inp, rmax = oracle.tt_add(g, g, batch=True), 32
    a, b, used = _sweep_both_ways(monkeypatch, lambda: gpu_tensor(inp, batch=True), lambda t: t.round_tt(rmax=rmax, algorithm=alg))
    assert used == (2 if case == "metric130" else 1)
    assert [tuple(x.shape) for x in a] == [tuple(x.shape) for x in b]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert all(torch.isfinite(x).all() for x in a)


@pytest.mark.parametrize("case", ["metric3", "scaled", "two_cores"])
def test_r_factors_normalised_by_the_factor_kernel_give_the_same_bits(case, monkeypatch):
    """The fp32 sweep keeps every R factor at O(1) by an exact power of two per item.  ABI 11: the factor kernel leaves R at the
    exponent it factored at (ttr_qr_factor_expo / ttr_qr_factor_pushed_expo) instead of one ttr_pow2_normalize launch per core:
    the rounded cores are BIT-identical to the separate launches' (host loop, TTR_FUSE_QR_NORM=0), on the metric's shape, on a
    train whose cores are scaled by 1e-9 and 1e+7 (norms far outside fp32's squares without the normalisation) and on two cores."""
    from tntorch_amd import _hipops
    if case == "metric3":
        inp, rmax = _metric_input(3, seed=21), 32
    elif case == "scaled":
        inp = _metric_input(2, seed=22)
        inp = [c * (1e-9 if i % 2 == 0 else 1e7) for i, c in enumerate(inp)]
        rmax = 32
    else:
        torch.manual_seed(3)
        inp, rmax = [torch.randn(4, 1, 40, 24), torch.randn(4, 24, 48, 1)], 7
    monkeypatch.setattr(_hipops, "SWEEP_C_ENABLED", False)
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(_hipops, "FUSE_QR_NORM", fused)
        t = gpu_tensor(inp, batch=True)
        t.round_tt(rmax=rmax)
        out[fused] = [c.clone() for c in t.cores]
    monkeypatch.setattr(_hipops, "SWEEP_C_ENABLED", True)
    t = gpu_tensor(inp, batch=True)
    t.round_tt(rmax=rmax)
    assert all(torch.equal(x, y) for x, y in zip(out[True], out[False]))
    assert all(torch.equal(x, y) for x, y in zip(out[True], t.cores))
    assert all(torch.isfinite(x).all() for x in out[True])


@pytest.mark.parametrize("dt,eps", [(torch.float32, 1e-3), (torch.float64, 1e-6), (torch.float64, 1e-14)])
@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_whole_sweep_entry_eps_mode_vs_host_loop_and_oracle(dt, eps, alg, monkeypatch):
    """The reference's own call signature -- NON-batch `t.round_tt(eps=..., rmax=...)` (tensor.py:2008-2014) -- through
    ttr_round_tt's eps mode (delta formed on the device, ranks read back ONCE): bit-identical to the host loop's deferred sweep,
    same ranks as its bond-by-bond sweep, the oracle's ranks (+- 1 at the cut) and train; rank-1 zeros for an all-zero train."""
    from tntorch_amd import _hipops
    inp = _decaying_tt([12, 10, 9, 11, 8, 16], 14, 0.6, dt, seed=3)
    monkeypatch.setenv("TTR_EPS_DEFERRED", "1")
    for rmax in (None, 6):
        a, b, used = _sweep_both_ways(monkeypatch, lambda: gpu_tensor(inp), lambda t: t.round_tt(eps=eps, rmax=rmax, algorithm=alg))
        assert used == 1 and ranks(to_list(a)) == ranks(to_list(b)) and all(torch.equal(x, y) for x, y in zip(a, b))
        assert all(x.is_contiguous() for x in a)
        if rmax is None:
            ref = oracle.round_tt([c.clone() for c in inp], eps=eps, algorithm=alg)
            if alg == "svd" or eps > 1e-10:
                assert all(abs(x - y) <= 1 for x, y in zip(ranks(to_list(a)), ranks(ref)))
            assert rel_diff(dense(to_list(a)), dense(ref)) <= max(3 * eps, 2e-5 if dt == torch.float32 else 1e-10)
            assert len(set(ranks(to_list(a))[1:-1])) > 1     # data-dependent ranks
        else:
            assert max(ranks(to_list(a))) <= 6
    monkeypatch.setattr(_hipops, "SWEEP_C_ENABLED", True)
    n0 = _hipops.SWEEP_C_CALLS
    z = gpu_tensor([torch.zeros_like(c) for c in inp])
    z.round_tt(eps=eps, algorithm=alg)
    assert _hipops.SWEEP_C_CALLS == n0 + 1 and ranks(to_list(z.cores)) == [1] * 7 and all((c == 0).all() for c in z.cores)
    # the default policy (`auto`): a rank cap -> one call; no cap -> the bond-by-bond loop (computing every bond at full rank
    # costs more than the readbacks save, `_eps_deferred_ok`)
    monkeypatch.setenv("TTR_EPS_DEFERRED", "auto")
    n0 = _hipops.SWEEP_C_CALLS
    t = gpu_tensor(inp); t.round_tt(eps=eps, rmax=6, algorithm=alg)
    u = gpu_tensor(inp); u.round_tt(eps=eps, algorithm=alg)
    assert _hipops.SWEEP_C_CALLS == n0 + 1


@pytest.mark.parametrize("call", ["rmax32", "eps1e-4", "eps1e-4_rmax40"])
def test_reference_call_signature_on_the_metric_shape(call, monkeypatch):
    """The reference's own NON-batch calls on one 64^8 rank-64 train t = g + g (`t.round_tt(rmax=32)`, `t.round_tt(eps=1e-4)`):
    the deferred eps-mode sweep -- first pass through ttr_eigh_top with `need_all` (the packed bonds' 32 x 32 problems by the
    top-r path, complete spectra), certified flat test on the `rows32` items, ranks read back once -- through ttr_round_tt and
    through the host loop bit for bit, and against the oracle: ranks, bond singular values, train (SURVEY 8c bounds)."""
    it = [c[0] for c in _metric_input(1, seed=31)]
    kw = {"rmax32": dict(rmax=32), "eps1e-4": dict(eps=1e-4), "eps1e-4_rmax40": dict(eps=1e-4, rmax=40)}[call]
    monkeypatch.setenv("TTR_EPS_DEFERRED", "1")
    a, b, used = _sweep_both_ways(monkeypatch, lambda: gpu_tensor(it), lambda t: t.round_tt(**kw))
    assert used == 1 and all(torch.equal(x, y) for x, y in zip(a, b))
    monkeypatch.setenv("TTR_EPS_DEFERRED", "0")          # the bond-by-bond sweep (one rank readback per bond): same train
    t = gpu_tensor(it); t.round_tt(**kw)
    assert ranks(to_list(t.cores)) == ranks(to_list(a)) and tt_rel_err(to_list(t.cores), to_list(a)) <= 1e-5
    ref = oracle.round_tt([c.clone() for c in it], algorithm="svd", **kw)
    ours = to_list(a)
    assert ranks(ours) == ranks(ref) == [1] + [32] * 7 + [1]
    assert tt_rel_err(ours, ref) <= 1e-5 and tt_rel_err(ours, it) <= 1e-5 and _right_orth_err(ours) <= 5e-5
    so, sr = oracle.bond_singular_values(ours), oracle.bond_singular_values(ref)
    assert all(((x - y).abs().max() / y.max()).item() <= 1e-5 for x, y in zip(so, sr))


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_whole_sweep_entry_eps_mode_long_last_core(dt, monkeypatch):
    """eps mode forms delta from ||last core|| on the device; a carry of 2^20 elements takes the chunked two-stage norm
    (`_hip.norm` / norm_one in ttr_roundtt.hip): ttr_round_tt and the host loop agree bit for bit, ranks as the oracle's."""
    torch.manual_seed(17)
    it = [torch.randn(1, 64, 64, dtype=dt), (torch.randn(64, 6, dtype=dt) @ torch.randn(6, 16384, dtype=dt)).reshape(64, 16384, 1)]
    monkeypatch.setenv("TTR_EPS_DEFERRED", "1")
    a, b, used = _sweep_both_ways(monkeypatch, lambda: gpu_tensor(it), lambda t: t.round_tt(eps=1e-3, rmax=16))
    assert used == 1 and all(torch.equal(x, y) for x, y in zip(a, b))
    ref = oracle.round_tt([c.clone() for c in it], eps=1e-3, rmax=16, algorithm="svd")
    assert ranks(to_list(a)) == ranks(ref) == [1, 6, 1]
    assert rel_diff(dense(to_list(a)), dense(ref)) <= (2e-5 if dt == torch.float32 else 1e-10)


def test_whole_sweep_entry_declines_what_it_does_not_cover(monkeypatch):
    """Outside ttr_round_tt's envelope (TT ranks above a TSQR panel, bonds with more rows than columns) the planner returns
    TTR_E_UNSUPPORTED and the host loop runs -- same results as ever; zero batches keep the batch-mode zero guard."""
    from tntorch_amd import _hip, _hipops
    assert _hip.round_tt_plan(torch.float32, [(1, 16, 16), (16, 16, 256), (256, 16, 16), (16, 16, 1)], [4] * 3, 1, False) == -2
    assert _hip.round_tt_plan(torch.float32, [(1, 64, 64)] + [(64, 64, 64)] * 6 + [(64, 64, 1)], [32] * 7, 2048, False) > 0
    assert _hip.round_tt_plan(torch.float32, [(1, 64, 64), (64, 64, 1)], [32], 2, True) == -2       # eps mode rounds ONE train
    torch.manual_seed(0)
    X = torch.randn(16, 16, 16, 16)
    n0 = _hipops.SWEEP_C_CALLS
    t = tn.Tensor(X.cuda(), ranks_tt=4)
    assert ranks(to_list(t.cores)) == [1, 4, 4, 4, 1]
    z = gpu_tensor([torch.zeros(5, 1, 8, 6), torch.zeros(5, 6, 8, 6), torch.zeros(5, 6, 8, 1)], batch=True)
    z.round_tt(rmax=3)
    assert _hipops.SWEEP_C_CALLS == n0 + 1                       # (only the zero batch: the dense ctor has its own sweep)
    assert [tuple(c.shape) for c in z.cores] == [(5, 1, 8, 1), (5, 1, 8, 1), (5, 1, 8, 1)] and all((c == 0).all() for c in z.cores)


def test_flat_spectrum_items_skip_the_second_gram_pass(monkeypatch):
    """Batch-mode 'svd' truncation: an item whose KEPT singular values lie within a factor 8 (here: 4) of each other is decided by the
    first Gram pass alone (ttr_spectrum_flat; the second pass exists for kept singular values far below sigma_1).  On the
    metric's shape (flat bonds: sigma_32 / sigma_1 ~ 0.7 on five of seven) the shortcut must (a) really trigger, (b) agree
    with the full two-pass result to 1.2e-5 (measured 6.7e-6: the one-pass result -- first pass by ttr_eigh_top -- is 2.6e-6 from
    the oracle, the two-pass one 8.3e-6; two fp32 results inside the 2e-5 parity bound), (c) stay inside the parity
    bounds against the oracle's LAPACK 'svd', and (d) not
    touch items with a decaying spectrum (bit-identical results there)."""
    from tntorch_amd import _hip, _hipops
    sg = torch.tensor([[4.0, 3.0, 2.0, 1.0, 0.1], [4.0, 3.0, 2.0, 0.9, 0.1], [0.0, 0.0, 0.0, 0.0, 0.0]], dtype=torch.float32).cuda()
    assert _hip.spectrum_flat(sg, 4, 0.25).tolist() == [1, 0, 0] and _hip.spectrum_flat(sg, 5, 0.02).tolist() == [1, 1, 0]
    inp = _metric_input(4, seed=11)
    monkeypatch.setattr(_hipops, "SWEEP_C_ENABLED", False)   # (spies on the host loop's per-kernel calls)
    calls = []
    orig = _hip.eigh_top      # (batch mode: the first-pass launch's flags ARE the pass-through flags, ttr_eigh_top -- no ttr_spectrum_flat launch)

    def spy(*a, **k):
        out = orig(*a, **k)
        calls.append((out[3] != 0).to(torch.int32))
        return out

    monkeypatch.setattr(_hip, "eigh_top", spy)
    res = {}
    for thr in (0.25, 0.0):
        monkeypatch.setattr(_hipops, "FLAT_SPECTRUM_THR", thr)
        t = gpu_tensor(inp, batch=True)
        t.round_tt(rmax=32)
        res[thr] = to_list(t.cores)
    skipped = sum(int(c.sum().item()) for c in calls)
    assert len(calls) == 7 and skipped >= 4 * 4          # at least four of the seven bonds of every item
    for i in range(4):
        a, b = [c[i] for c in res[0.25]], [c[i] for c in res[0.0]]
        assert tt_rel_err(a, b) <= 1.2e-5
        ref = oracle.round_tt([c[i] for c in inp], rmax=32, algorithm="svd")
        assert ranks(a) == ranks(ref) and tt_rel_err(a, ref) <= 2e-5
        so, sr = oracle.bond_singular_values(a), oracle.bond_singular_values(ref)
        assert all(((x - y).abs().max() / y.max()).item() <= 2e-5 for x, y in zip(so, sr))
    assert _right_orth_err([c[0] for c in res[0.25]]) <= 5e-6
    # decaying spectrum: nothing is flat, the shortcut changes nothing
    dec = _decaying_tt([20, 18, 16, 18, 20], 24, 1.0, torch.float32, seed=5, batch=3)
    out = {}
    for thr in (0.25, 0.0):
        monkeypatch.setattr(_hipops, "FLAT_SPECTRUM_THR", thr)
        t = gpu_tensor(dec, batch=True)
        t.round_tt(rmax=12)
        out[thr] = to_list(t.cores)
    assert all(torch.equal(x, y) for x, y in zip(out[0.25], out[0.0]))


def test_top_r_first_pass_on_the_metric_shape(monkeypatch):
    """Batch-mode bonds with <= 64 rows and rmax <= 32: pass 1 through ttr_eigh_top (the r largest eigenpairs where the kept
    spectrum is flat and free of close pairs, QL inside the same launch otherwise) against the plain QL first pass: same ranks,
    results within 1.2e-5 of each other (measured 6.7e-6: 2.6e-6 and 8.3e-6 from the oracle) and BOTH inside the parity bound
    against the oracle's LAPACK 'svd'."""
    from tntorch_amd import _hip, _hipops
    inp = _metric_input(4, seed=23)
    monkeypatch.setattr(_hipops, "SWEEP_C_ENABLED", False)   # (spies on the host loop's per-kernel calls)
    calls = []
    orig = _hip.eigh_top
    monkeypatch.setattr(_hip, "eigh_top", lambda *a, **k: calls.append(orig(*a, **k)) or calls[-1])
    res = {}
    for on in (True, False):
        monkeypatch.setattr(_hipops, "EIGH_TOP_ENABLED", on)
        t = gpu_tensor(inp, batch=True)
        t.round_tt(rmax=32)
        res[on] = to_list(t.cores)
    assert len(calls) == 7 and sum(int(c[3].sum().item()) for c in calls) >= 4 * 4   # (only from the run with the switch on)
    for i in range(4):
        a, b = [c[i] for c in res[True]], [c[i] for c in res[False]]
        assert ranks(a) == ranks(b) and tt_rel_err(a, b) <= 1.2e-5
        ref = oracle.round_tt([c[i] for c in inp], rmax=32, algorithm="svd")
        assert ranks(a) == ranks(ref) and tt_rel_err(a, ref) <= 2e-5 and tt_rel_err(b, ref) <= 2e-5
    assert _right_orth_err([c[0] for c in res[True]]) <= 5e-6


def test_flat_spectrum_shortcut_in_eps_mode(monkeypatch):
    """eps mode (non-batch): the rank comes from the tail energies, so an item only takes the one-pass shortcut when the rank rule's
    decision on pass 1's sigma is robust against pass 1's absolute error E = 64 n eps sigma_1^2 (ttr_spectrum_flat with use_delta).
    Flag logic on hand-made spectra, then an fp64 rank-deficient train (config C2's structure): the shortcut must trigger on every
    interior bond and give the two-pass result (ranks identical, cores to 1e-11), i.e. the oracle's."""
    from tntorch_amd import _hip, _hipops
    d2 = 1e-8
    sg = torch.tensor([[1.0, 0.8, 0.6, 0.5, 1e-9, 1e-9, 1e-9, 1e-9],      # clear gap: rank 4, kept values within a factor 2
                       [1.0, 0.8, 0.6, 0.5, 1e-4, 1e-9, 1e-9, 1e-9],      # sigma_5^2 = delta^2: the decision is not robust
                       [1.0, 0.8, 0.6, 0.05, 1e-9, 1e-9, 1e-9, 1e-9],     # robust rank 4, but sigma_4 < sigma_1 / 8
                       [1.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3]], dtype=torch.float64).cuda()  # nothing to cut, all kept values flat
    assert _hip.spectrum_flat(sg, 8, 0.125, True, d2).tolist() == [1, 0, 0, 1]
    assert _hip.spectrum_flat(sg, 2, 0.125, True, d2).tolist() == [1, 1, 1, 1]      # the cap decides: tail(1) > delta^2 + E everywhere
    # eps = 1e-14 (the default of round_tt(rmax=r), tensor.py:2008-2014) on a spectrum with a noise tail: the cap binds robustly
    assert _hip.spectrum_flat(sg, 4, 0.125, True, 1e-29).tolist() == [1, 1, 0, 1]
    d2_dev = torch.tensor([d2], dtype=torch.float64).cuda()
    assert _hip.spectrum_flat(sg, 8, 0.125, True, 0.0, d2_dev).tolist() == [1, 0, 0, 1]
    assert _hip.spectrum_flat(sg, 8, 0.125, True, 0.0).tolist() == [0, 0, 0, 1]     # delta = 0: the batch criterion at keep = 8
    # fp32: E = 64 * 8 * 1.2e-7 = 6e-5 > delta^2 -- what is cut cannot be certified, only the nothing-to-cut item qualifies
    assert _hip.spectrum_flat(sg.float(), 8, 0.125, True, d2).tolist() == [0, 0, 0, 1]
    # `rows32` items: sigma[32..] are structural zeros (exact in either pass) -- cutting them is certain, only the 32 computed values
    # are tested.  fp32, eps = 1e-14 (what `t.round_tt(rmax=32)` passes on): a flat 32-value spectrum + 32 structural zeros qualifies
    # with the flag, not without (without it the zeros are "values within E of delta^2")
    s64 = torch.zeros(2, 64, dtype=torch.float32)
    s64[:, :32] = torch.linspace(1.0, 0.6, 32)
    s64[1, 31] = 1e-3                                                        # kept value far below sigma_1: needs the second pass
    fl = torch.tensor([1, 1], dtype=torch.int32).cuda()
    # (the structural-zero reading belongs to the rank rule WITHOUT the noise floor, TTR_STRICT_RANKS=0: exact zeros are cut)
    _hip.set_knob(_hip.KNOB_RANK_NOISE_FLOOR, 0)
    try:
        assert _hip.spectrum_flat(s64.cuda(), 32, 0.125, True, 1e-29, rows32=fl).tolist() == [1, 0]
        assert _hip.spectrum_flat(s64.cuda(), 32, 0.125, True, 1e-29).tolist() == [0, 0]
        assert _hip.spectrum_flat(s64.cuda(), 48, 0.125, True, 1e-29, rows32=fl).tolist() == [1, 0]     # the cap above the live block
    finally:
        _hip.set_knob(_hip.KNOB_RANK_NOISE_FLOOR, 1)
    # the default (round 6): the rule sees the 32 zeros at eps sigma_1 like LAPACK's noise -- 32 x (1.2e-7)^2 >> delta^2, they are
    # KEPT by the rule and the cap decides, robustly where the keep-th value carries more than E (item 0), with or without the flag;
    # a cap of 48 keeps 16 of the noise-level values: far below sigma_1 / 8, the second pass (and the completion) runs
    assert _hip.spectrum_flat(s64.cuda(), 32, 0.125, True, 1e-29, rows32=fl).tolist() == [1, 0]
    assert _hip.spectrum_flat(s64.cuda(), 32, 0.125, True, 1e-29).tolist() == [1, 0]
    assert _hip.spectrum_flat(s64.cuda(), 48, 0.125, True, 1e-29, rows32=fl).tolist() == [0, 0]
    monkeypatch.setattr(_hipops, "SWEEP_C_ENABLED", False)   # (below: spies on the host loop's per-kernel calls)

    torch.manual_seed(3)
    g = oracle.tt_randn([12] * 6, 6, dtype=torch.float64)
    inp = oracle.tt_add(g, g)   # TT ranks 12, numerical ranks 6 (3e6 entries: compared densely -- tt_rel_err resolves 1e-8)
    calls = []
    orig = _hip.spectrum_flat
    monkeypatch.setattr(_hip, "spectrum_flat", lambda *a, **k: calls.append(orig(*a, **k)) or calls[-1])
    res = {}
    for thr in (0.125, 0.0):
        monkeypatch.setattr(_hipops, "FLAT_SPECTRUM_THR", thr)
        t = gpu_tensor(inp)
        t.round_tt(eps=1e-6)
        res[thr] = [c.cpu() for c in t.cores]
    assert len(calls) == 5 and sum(int(c.sum().item()) for c in calls) >= 2   # (sigma_12 / sigma_1 of a random train is not above 1/8 on every bond)
    assert ranks(res[0.125]) == ranks(res[0.0]) == [1, 6, 6, 6, 6, 6, 1]
    assert rel_diff(dense(res[0.125]), dense(res[0.0])) <= 1e-12
    ref = oracle.round_tt([c.clone() for c in inp], eps=1e-6, algorithm="svd")
    assert ranks(res[0.125]) == ranks(ref) and rel_diff(dense(res[0.125]), dense(ref)) <= 1e-12


@pytest.mark.parametrize("scale", [1.0, 1e18, 1e-18])
def test_dense_single_norm_from_the_first_gram_matrix(scale, monkeypatch):
    """A single config-scale dense tensor (forced here on a small one): ||X|| -- delta and the fp32 range guard -- comes from the trace
    of the first bond's Gram matrix, which is handed on to the first truncation; out of range it falls back to the norm pass and
    the scaled input.  Same result as the eager path (in range: delta differs by fp32 rounding of the trace only)."""
    from tntorch_amd import _hipops
    torch.manual_seed(8)
    low = oracle.tt_to_dense(oracle.tt_randn([12, 11, 10, 9], 3, dtype=torch.float64))
    X = ((low / low.norm() + 1e-4 * torch.randn(low.shape, dtype=torch.float64) / math.sqrt(low.numel())) * scale).float()
    out = {}
    for mode, limit in (("lazy", 0), ("eager", 1 << 60)):
        monkeypatch.setattr(_hipops, "_LAZY_GUARD_BYTES", limit)
        for kw in ({"ranks_tt": 3}, {"eps": 1e-3}):
            t = tn.Tensor(X.cuda(), **kw)
            out[mode, tuple(kw)] = [c.cpu() for c in t.cores]
            assert all(torch.isfinite(c).all() for c in out[mode, tuple(kw)])
    for kw in (("ranks_tt",), ("eps",)):
        a, b = out["lazy", kw], out["eager", kw]
        assert ranks(a) == ranks(b)
        assert rel_diff(dense([c.double() for c in a]) / scale, dense([c.double() for c in b]) / scale) <= 1e-5


@pytest.mark.parametrize("scale", [1.0, 1e18, 1e-18])
def test_dense_batch_range_guard_from_the_first_gram_matrix(scale, monkeypatch):
    """Batch-mode dense -> TT in fp32: from 256 MB on the range guard is read off the trace of the first bond's Gram matrix instead of
    a norm pass over the input (forced here on a small input).  In range: the same kernels on the same data, bit-identical; out of
    range (Gram entries overflow / fall into the denormals): the sweep restarts on the scaled input and agrees with the eager guard."""
    from tntorch_amd import _hipops
    torch.manual_seed(6)
    X = (torch.randn(3, 12, 11, 10, 9, dtype=torch.float64) * scale).float().cuda()
    out = {}
    for lazy, limit in (("lazy", 0), ("eager", 1 << 60)):
        monkeypatch.setattr(_hipops, "_LAZY_GUARD_BYTES", limit)
        t = tn.Tensor(X, ranks_tt=4, batch=True)
        out[lazy] = [c.cpu() for c in t.cores]
        assert all(torch.isfinite(c).all() for c in out[lazy])
    if scale == 1.0:
        assert all(torch.equal(a, b) for a, b in zip(out["lazy"], out["eager"]))
    for i in range(3):
        a = dense([c[i].double() for c in out["lazy"]]) / scale
        b = dense([c[i].double() for c in out["eager"]]) / scale
        assert rel_diff(a, b) <= 1e-5
        ref = oracle.dense_to_tt((X[i].cpu().double() / scale).float(), 4)
        e_o, e_r = rel_diff(a, X[i].cpu().double() / scale), rel_diff(dense(ref), (X[i].cpu().double() / scale))
        assert abs(e_o - e_r) <= 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("batch", [False, True])
def test_big_bond_truncation_from_selected_eigenpairs(batch, dt, monkeypatch):
    """Bonds whose Gram matrix is larger than one workgroup (here 144 x 144) with a rank cap far below: a flat spectrum (dense
    random data) is decided by the r largest eigenpairs alone (ttr_tridiag / ttr_tri_eigsel / ttr_tridiag_back) -- in batch mode
    and, when the cap provably binds, for a single tensor in eps mode; a decaying spectrum (low rank + noise) is not flat and
    takes the block-Jacobi path as before (bit-identical to the solver switched off).  Against the oracle either way."""
    from tntorch_amd import _hip, _hipops
    torch.manual_seed(9)
    shape, r = [12, 12, 40, 9], 4
    Xr = torch.randn(shape, dtype=torch.float64).to(dt)
    low = oracle.tt_to_dense(oracle.tt_randn(shape, 3, dtype=torch.float64))
    Xl = (low / low.norm() + 1e-3 * torch.randn(shape, dtype=torch.float64) / math.sqrt(low.numel())).to(dt)
    calls = []
    orig = _hip.eigh_topk
    monkeypatch.setattr(_hip, "eigh_topk", lambda G, k: calls.append(tuple(G.shape)) or orig(G, k))
    monkeypatch.setattr(_hipops, "SUBSPACE_ENABLED", False)   # (round 4: the range finder would take the low-rank input first)

    def run(X):
        if batch:
            t = tn.Tensor(torch.stack([X, 2 * X]).cuda(), ranks_tt=r, batch=True)
            return [c[0].cpu() for c in t.cores]
        return [c.cpu() for c in tn.Tensor(X.cuda(), ranks_tt=r).cores]

    for X, flat in ((Xr, True), (Xl, False)):
        calls.clear()
        monkeypatch.setattr(_hipops, "EIGH_TOPK_ENABLED", True)
        ours = run(X)
        assert any(s[-1] == 144 for s in calls)          # the solver was tried on the 144 x 144 bond ...
        monkeypatch.setattr(_hipops, "EIGH_TOPK_ENABLED", False)
        base = run(X)
        if not flat:                                      # ... and its answer declined: nothing changes
            assert all(torch.equal(a, b) for a, b in zip(ours, base))
        ref = oracle.dense_to_tt(X, r)
        assert ranks(ours) == ranks(ref)
        e_o, e_b, e_r = rel_diff(dense(ours), X), rel_diff(dense(base), X), rel_diff(dense(ref), X)
        bound = 1e-5 if dt == torch.float32 else 1e-10
        assert abs(e_o - e_r) <= bound and abs(e_b - e_r) <= bound


# ------------------------------------------------------------------ big bonds with concentrated spectra: the certified range finder
def _lowrank_noise(m, n, r, noise, dt, seed, decay=0.0):
    g = torch.Generator().manual_seed(seed)
    U = torch.linalg.qr(torch.randn(m, r, generator=g, dtype=torch.float64))[0]
    V = torch.linalg.qr(torch.randn(n, r, generator=g, dtype=torch.float64))[0]
    s = 2.0 ** (-decay * torch.arange(r, dtype=torch.float64))
    M = (U * s) @ V.T
    M = M / M.norm() + noise * torch.randn(m, n, generator=g, dtype=torch.float64) / math.sqrt(m * n)
    return M.to(dt)


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("shape", [(3000, 256), (200, 1100), (1024, 1024)])
@pytest.mark.parametrize("batch", [False, True])
def test_big_bond_range_finder_vs_lapack(dt, shape, batch):
    """Both dimensions above 64, rank cap 8, energy concentrated in 8 directions (+ 1e-3 noise; second item: sigma_j ~ 2^(-j/2)):
    `truncate` takes the subspace path (Gram matrix -> certified range finder -> fused small truncation) and returns what
    LAPACK's SVD of the same matrix gives: kept singular values to 2e-5 sigma_1 (fp32) / 1e-11 (fp64), the same
    approximation error, orthonormal rows of the right factor.  Tall (right Gram), wide (left Gram) and square."""
    from tntorch_amd import _hipops
    m, n = shape
    r = 8
    f32 = dt == torch.float32
    Ms = [_lowrank_noise(m, n, r, 1e-3, dt, 1), _lowrank_noise(m, n, r, 1e-3, dt, 2, decay=0.5)]
    _hipops.PATH_TRACE = []
    try:
        if batch:
            left, right = tn.truncated_svd(torch.stack(Ms).cuda(), rmax=r, left_ortho=False, batch=True)
            outs = [(left[i].cpu(), right[i].cpu()) for i in range(2)]
        else:
            outs = []
            for M in Ms:
                left, right = tn.truncated_svd(M.cuda(), rmax=r, left_ortho=False)
                outs.append((left.cpu(), right.cpu()))
        paths = [p[0] for p in _hipops.PATH_TRACE]
    finally:
        _hipops.PATH_TRACE = None
    assert paths and all(p == "subspace" for p in paths), paths
    for M, (left, right) in zip(Ms, outs):
        assert left.shape == (m, r) and right.shape == (r, n)
        Md = M.double()
        sv = torch.linalg.svdvals(Md)
        ours_sv = torch.linalg.svdvals(left.double())          # right has orthonormal rows: sigma(left) = kept sigma
        assert ((ours_sv - sv[:r]).abs().max() / sv[0]).item() <= (2e-5 if f32 else 1e-11)
        e_o = rel_diff(left.double() @ right.double(), Md)
        e_r = (sv[r:].square().sum().sqrt() / sv.square().sum().sqrt()).item()
        assert abs(e_o - e_r) <= (1e-5 if f32 else 1e-11), (e_o, e_r)
        Rr = right.double()
        assert (Rr @ Rr.T - torch.eye(r, dtype=torch.float64)).abs().max().item() <= (3e-5 if f32 else 1e-11)


@pytest.mark.parametrize("left_ortho", [True, False])
def test_big_bond_range_finder_declines_flat_and_heavy_tails(left_ortho):
    """`randn` (flat: the participation ratio exceeds the basis) and a slowly decaying spectrum (sigma_j ~ 1/j: the energy
    outside any 32-dimensional basis is not small against sigma_8^2) fail the certificate and take the full paths; results
    against LAPACK either way."""
    from tntorch_amd import _hipops
    g = torch.Generator().manual_seed(5)
    m, n, r = 2048, 256, 8
    U = torch.linalg.qr(torch.randn(m, n, generator=g, dtype=torch.float64))[0]
    V = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))[0]
    heavy = ((U / torch.arange(1, n + 1, dtype=torch.float64)) @ V.T).float()
    flat = torch.randn(m, n, generator=g, dtype=torch.float32)
    for M in (flat, heavy):
        _hipops.PATH_TRACE = []
        try:
            left, right = tn.truncated_svd(M.cuda(), rmax=r, left_ortho=left_ortho)
            paths = [p[0] for p in _hipops.PATH_TRACE]
        finally:
            _hipops.PATH_TRACE = None
        assert paths and "subspace" not in paths, paths
        sv = torch.linalg.svdvals(M.double())
        e_o = rel_diff(left.cpu().double() @ right.cpu().double(), M.double())
        e_r = (sv[r:].square().sum().sqrt() / sv.square().sum().sqrt()).item()
        assert abs(e_o - e_r) <= 1e-5, (e_o, e_r)


def test_c1_proxy_lowrank_takes_range_finder():
    """The C1 proxy of `test_c1_proxy_dense_64_4_vs_oracle` (dense 64^4, TT rank 16 + 1e-3 noise, eps mode with the cap): its
    n = 1024 bond is decided by the certified range finder, not by the n x n eigen-decomposition."""
    from tntorch_amd import _hipops
    torch.manual_seed(11)
    shape, r = [64] * 4, 16
    low = oracle.tt_to_dense(oracle.tt_randn(shape, r, dtype=torch.float64))
    X = (low / low.norm() * math.sqrt(low.numel()) + 1e-3 * torch.randn(shape, dtype=torch.float64)).float()
    _hipops.PATH_TRACE = []
    try:
        t = tn.Tensor(X.cuda(), ranks_tt=r)
        trace = list(_hipops.PATH_TRACE)
    finally:
        _hipops.PATH_TRACE = None
    assert [p for p in trace if p[2] == 1024] == [("subspace", 4096, 1024, 16)], trace
    assert list(t.ranks_tt) == [1, 16, 16, 16, 1]
    e_o = rel_diff(t.torch().cpu().double(), X.double())
    assert abs(e_o - 1e-3) <= 2e-5, e_o


def test_cp_hosvd_init_selected_eigenpairs():
    """tensor.py:228-277 at a mode size above one workgroup (I = 160, R = 8: the shape class of BASELINE C4's 256-mode init): the R
    leading eigenvectors of every mode Gram matrix come from the selected-eigenpair solver and equal LAPACK's up to sign."""
    from tntorch_amd import _hip, _hipops
    torch.manual_seed(23)
    shape, R = [160, 24, 160], 8
    fac = [torch.randn(i, R, dtype=torch.float64) * (2.0 ** (-torch.arange(R, dtype=torch.float64) / 3)) for i in shape]   # component weights 2^-j
    X = oracle.cp_to_dense(fac)
    X = (X / X.norm() + 1e-3 * torch.randn(shape, dtype=torch.float64) / math.sqrt(X.numel())).float()
    calls = []
    orig = _hip.eigh_topk
    _hip.eigh_topk = lambda G, k: calls.append(tuple(G.shape)) or orig(G, k)
    try:
        ours = _hipops.cp_hosvd_init(X.cuda(), R)
    finally:
        _hip.eigh_topk = orig
    assert [c[-1] for c in calls] == [160, 160]          # modes 0 and 2 (mode 1: I = 24, one workgroup)
    ref = oracle.cp_hosvd_init(X.double(), R)
    for a, b in zip(ours, ref):
        a = a.cpu().double()
        assert a.shape == b.shape
        d = (a.T @ b).abs()                               # |cosines|: the identity up to sign for a separated spectrum
        assert (d - torch.eye(R, dtype=torch.float64)).abs().max() < 1e-2, d.diagonal()   # (fp32: eps lambda_1 / gap ~ 2e-3 for the smallest pair)
        assert (a.T @ a - torch.eye(R, dtype=torch.float64)).abs().max() < 1e-5


@pytest.mark.parametrize("kind", ["randn", "lowrank"])
def test_from_dense_consuming_matches_constructor(kind, monkeypatch):
    """`Tensor.from_dense_consuming` (the first carry written in place over the consumed front of the input: BASELINE C1 at
    64^6 has no room for a carry next to it) returns exactly what the constructor returns; the row ranges of the in-place
    projection are forced small so that five launches chain (each writes into rows the previous ones consumed)."""
    from tntorch_amd import _hipops
    monkeypatch.setattr(_hipops, "_INPLACE_FIRST_ROWS", 1024)
    torch.manual_seed(31)
    shape, r = [64, 64, 32, 64], 16
    if kind == "randn":
        X = torch.randn(shape, dtype=torch.float32)
    else:
        low = oracle.tt_to_dense(oracle.tt_randn(shape, r, dtype=torch.float64))
        X = (low / low.norm() * math.sqrt(low.numel()) + 1e-3 * torch.randn(shape, dtype=torch.float64)).float()
    ref = tn.Tensor(X.cuda(), ranks_tt=r)
    Xd = X.cuda()
    t = tn.Tensor.from_dense_consuming(Xd, r)
    assert list(t.ranks_tt) == list(ref.ranks_tt) == [1, 16, 16, 16, 1]
    for a, b in zip(t.cores, ref.cores):
        assert torch.equal(a, b)
    assert not torch.equal(Xd.cpu(), X)      # (the input really was consumed)
    with pytest.raises(ValueError):
        tn.Tensor.from_dense_consuming(X, r)  # CPU tensors: the constructor is the way


def test_from_dense_consuming_with_a_long_last_mode_completes_with_a_warning():
    """A last mode above the 64 columns of the fused column-sweep kernels has no in-place carry: the call completes through the
    constructor's path (as in rounds 1 - 4) and says so, instead of raising."""
    torch.manual_seed(32)
    X = torch.randn(16, 16, 96, dtype=torch.float32)
    ref = tn.Tensor(X.cuda(), ranks_tt=8)
    with pytest.warns(RuntimeWarning, match="last mode > 64"):
        t = tn.Tensor.from_dense_consuming(X.cuda(), 8)
    assert list(t.ranks_tt) == list(ref.ranks_tt)
    for a, b in zip(t.cores, ref.cores):
        assert torch.equal(a, b)


def test_free_function_round_tt_does_not_copy_the_train_but_never_aliases_it():
    """`tn.round_tt(t)` (round.py:7-19: clone, then round in place): on the device the clone is shallow -- the rounding rebinds the
    cores, it never writes into them -- so the input is bit-for-bit unchanged, the result equals the method's, and no core of the
    result shares storage with the input (a one-core tensor is copied after all)."""
    g = oracle.tt_randn([12] * 5, 6, dtype=torch.float32)
    inp = oracle.tt_add(g, g)
    t = gpu_tensor(inp)
    before = [c.clone() for c in t.cores]
    r = tn.round_tt(t, rmax=6)
    assert all(torch.equal(a, b) for a, b in zip(t.cores, before)) and list(t.ranks_tt) == [1, 12, 12, 12, 12, 1]
    m = gpu_tensor(inp)
    m.round_tt(rmax=6)
    assert list(r.ranks_tt) == list(m.ranks_tt) and all(torch.equal(a, b) for a, b in zip(r.cores, m.cores))
    held = {c.untyped_storage().data_ptr() for c in t.cores}
    assert all(c.untyped_storage().data_ptr() not in held for c in r.cores)
    one = tn.Tensor([torch.randn(1, 7, 1).cuda()])
    r1 = tn.round_tt(one)
    assert torch.equal(r1.cores[0], one.cores[0]) and r1.cores[0].untyped_storage().data_ptr() != one.cores[0].untyped_storage().data_ptr()


def test_verbose_prints_the_reference_stage_lines_on_device(capsys):
    t = gpu_tensor(oracle.tt_randn([16, 16, 16, 16], 8, dtype=torch.float32))
    t.round_tt(rmax=3, verbose=True)
    out = capsys.readouterr().out.splitlines()
    assert [ln.split(":")[0] for ln in out] == ["Orthogonalization time"] + ["Time (SVD)", "Time (product)"] * 3
    t = gpu_tensor(oracle.tt_randn([16, 16, 16], 8, dtype=torch.float64))
    t.round_tt(eps=1e-3, algorithm="eig", verbose=True)
    out = capsys.readouterr().out.splitlines()
    assert [ln.split(":")[0] for ln in out] == ["Orthogonalization time"] + ["Time (gram)", "Time (symmetric EIG)", "Time (product)"] * 2
