"""Pin the oracle (oracle/tt_oracle.py) against the golden vectors recorded from the
unmodified reference (oracle/gen_golden.py) and the notebook known answers."""
import pytest
import torch

import oracle
from parity import analytic_128, load_case, load_meta


def _max_abs(a, b):
    return max((x - y).abs().max().item() for x, y in zip(a, b))


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_round_eps_f64(alg):
    g = load_case("round_eps_f64")
    out = oracle.round_tt(g["inp"], eps=1e-8, algorithm=alg)
    assert oracle.tt_ranks(out) == oracle.tt_ranks(g[alg])
    assert _max_abs(out, g[alg]) < 1e-9


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_round_rmax_f32(alg):
    g = load_case("round_rmax_f32")
    out = oracle.round_tt(g["inp"], rmax=3, algorithm=alg)
    assert oracle.tt_ranks(out) == [1, 3, 3, 3, 3, 1]
    assert _max_abs(out, g[alg]) < 2e-3  # float32 LAPACK, flat randn spectrum
    assert (oracle.tt_to_dense(out) - oracle.tt_to_dense(g[alg])).norm() / oracle.tt_to_dense(g[alg]).norm() < 1e-4


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_round_batch_f64(alg):
    g = load_case("round_batch_f64")
    out = oracle.round_tt(g["inp"], rmax=2, algorithm=alg, batch=True)
    assert _max_abs(out, g[alg]) < 1e-10


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_dense_f64(alg):
    g = load_case("dense_f64")
    out = oracle.dense_to_tt(g["X"], 4, algorithm=alg)
    assert oracle.tt_ranks(out) == [1, 4, 4, 4, 4, 1]
    assert _max_abs(out, g[alg]) < 1e-9


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_dense_batch_f32(alg):
    g = load_case("dense_batch_f32")
    out = oracle.dense_to_tt(g["X"], 3, algorithm=alg, batch=True)
    d_out = oracle.tt_to_dense(out, batch=True)
    d_ref = oracle.tt_to_dense(g[alg], batch=True)
    assert (d_out - d_ref).norm() / d_ref.norm() < 1e-4


def test_c0_input_checksum_and_result():
    meta = load_meta()["cases"]["c0_16x4_rmax4_f32"]
    g = load_case("c0_16x4_rmax4_f32")
    torch.manual_seed(0)
    X = torch.randn(16, 16, 16, 16, dtype=torch.float32)
    assert abs(float(X.double().sum()) - meta["x_sum"]) < 1e-6
    assert abs(float((X.double() ** 2).sum()) - meta["x_sumsq"]) < 1e-6
    for alg in ("svd", "eig"):
        out = oracle.round_tt(oracle.full_rank_tt(X), rmax=4, algorithm=alg)
        assert oracle.tt_ranks(out) == [1, 4, 4, 4, 1]
        d_out, d_ref = oracle.tt_to_dense(out), oracle.tt_to_dense(g[alg])
        assert (d_out - d_ref).norm() / d_ref.norm() < 1e-4
        # approximation error (gauge invariant) agrees with the reference's
        e_out = (d_out - X).norm() / X.norm()
        e_ref = (d_ref - X).norm() / X.norm()
        assert abs(e_out - e_ref) < 1e-5


def test_truncated_svd_calls():
    g = load_case("truncated_svd_f64")
    calls = load_meta()["cases"]["truncated_svd_f64"]["calls"]
    for c in calls:
        kw = {k: c[k] for k in ("eps", "rmax", "delta") if k in c}
        u, v = oracle.truncated_svd(g["M_" + c["M"]], left_ortho=c["left_ortho"], algorithm=c["algorithm"], **kw)
        assert u.shape == g[f"call{c['i']}_left"].shape, c
        assert (u - g[f"call{c['i']}_left"]).abs().max() < 1e-10, c
        assert (v - g[f"call{c['i']}_right"]).abs().max() < 1e-10, c
    for alg in ("svd", "eig"):
        u, v = oracle.truncated_svd(g["Mb"], batch=True, algorithm=alg)
        assert (u - g[f"batch_{alg}_left"]).abs().max() < 1e-9
        assert (v - g[f"batch_{alg}_right"]).abs().max() < 1e-9


def test_truncated_svd_errors():
    M = torch.rand(4, 5)
    with pytest.raises(ValueError):
        oracle.truncated_svd(M, delta=0.1, eps=0.1)
    with pytest.raises(AssertionError):
        oracle.truncated_svd(M, algorithm="qr")


def test_orthogonalize():
    g = load_case("orthogonalize_f64")
    a = [c.clone() for c in g["inp"]]; oracle.left_orthogonalize(a, 0)
    b = [c.clone() for c in g["inp"]]; oracle.right_orthogonalize(b, 4)
    c = [x.clone() for x in g["inp"]]; oracle.orthogonalize(c, 2)
    d = [x.clone() for x in g["inp"]]; oracle.orthogonalize(d, 4)
    assert _max_abs(a, g["left0"]) < 1e-12
    assert _max_abs(b, g["right4"]) < 1e-12
    assert _max_abs(c, g["orth2"]) < 1e-12
    assert _max_abs(d, g["orth4"]) < 1e-12


def test_known_answers_notebook():
    """decompositions.ipynb cell 3 (ranks_tt=3) and cell 18 (round_tt(eps=1e-5))."""
    ka = load_meta()["known_answers"]["measured_with_reference_here"]
    full = analytic_128()
    t3 = oracle.dense_to_tt(full, 3)
    assert oracle.tt_ranks(t3) == [1, 3, 3, 1] == ka["ranks_tt3"]
    e3 = ((oracle.tt_to_dense(t3) - full).norm() / full.norm()).item()
    assert abs(e3 - ka["relerr_tt3"]) < 1e-9 and abs(e3 - 0.0005) < 5e-5
    t = oracle.round_tt(oracle.full_rank_tt(full), eps=1e-5)
    assert oracle.tt_ranks(t) == [1, 4, 6, 1] == ka["ranks_eps1e-5_svd"]
    e = ((oracle.tt_to_dense(t) - full).norm() / full.norm()).item()
    assert abs(e - 8.3358e-06) < 1e-9


# ------------------------------------------------------------------ Tucker rounding / round()
def _tucker_ranks(cores):
    return [c.shape[-2] for c in cores]


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_round_tucker_eps_f64(alg):
    g = load_case("round_tucker_eps_f64")
    cores, Us = oracle.round_tucker(g["inp"], None, eps=1e-8, algorithm=alg)
    assert _tucker_ranks(cores) == _tucker_ranks(g[f"{alg}_cores"]) == [3, 5, 7, 3]  # bounded by R_k * R_{k+1}
    assert _max_abs(cores, g[f"{alg}_cores"]) < 1e-8 and _max_abs(Us, g[f"{alg}_Us"]) < 1e-8
    X = oracle.tt_to_dense(g["inp"])
    assert (oracle.tucker_to_dense(cores, Us) - X).norm() / X.norm() < 1e-8


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_round_tucker_rmax_f32(alg):
    g = load_case("round_tucker_rmax_f32")
    cores, Us = oracle.round_tucker(g["inp"], None, rmax=3, algorithm=alg)
    assert _tucker_ranks(cores) == [3, 3, 3, 3]
    d_out, d_ref = oracle.tucker_to_dense(cores, Us), oracle.tucker_to_dense(g[f"{alg}_cores"], g[f"{alg}_Us"])
    assert (d_out - d_ref).norm() / d_ref.norm() < 1e-4


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_round_tucker_batch_f64(alg):
    g = load_case("round_tucker_batch_f64")
    cores, Us = oracle.round_tucker(g["inp"], None, rmax=2, algorithm=alg, batch=True)
    assert _max_abs(cores, g[f"{alg}_cores"]) < 1e-9 and _max_abs(Us, g[f"{alg}_Us"]) < 1e-9


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_ctor_tucker_f64(alg):
    g = load_case("ctor_tucker_f64")
    cores, Us = oracle.dense_to_tucker_tt(g["inp"], ranks_tucker=4, ranks_tt=3, algorithm=alg)
    assert oracle.tt_ranks(cores) == [1, 3, 3, 3, 1] and _tucker_ranks(cores) == [4, 4, 4, 4]
    assert _max_abs(cores, g[f"{alg}_cores"]) < 1e-8 and _max_abs(Us, g[f"{alg}_Us"]) < 1e-8


@pytest.mark.parametrize("alg", ["svd", "eig"])
def test_round_general_f64(alg):
    g = load_case("round_general_f64")
    cores, Us = oracle.round_general(g["inp"], eps=1e-6, algorithm=alg)
    assert oracle.tt_ranks(cores) == oracle.tt_ranks(g[f"{alg}_cores"])
    assert _tucker_ranks(cores) == _tucker_ranks(g[f"{alg}_cores"])
    d_out, d_ref = oracle.tucker_to_dense(cores, Us), oracle.tucker_to_dense(g[f"{alg}_cores"], g[f"{alg}_Us"])
    assert (d_out - d_ref).norm() / d_ref.norm() < 1e-9


def test_known_answer_eps_ctor():
    """decompositions.ipynb cell 14: tn.Tensor(full, eps=1e-5) -> TT ranks [1,4,6,1], Tucker 4,5,6, 8.3402e-06."""
    full = analytic_128()
    cores, Us = oracle.dense_to_tucker_tt(full, eps=1e-5)
    assert oracle.tt_ranks(cores) == [1, 4, 6, 1] and _tucker_ranks(cores) == [4, 5, 6]
    err = ((oracle.tucker_to_dense(cores, Us) - full).norm() / full.norm()).item()
    assert abs(err - 8.340228167320888e-06) < 1e-10


# ------------------------------------------------------------------ CP-ALS (SURVEY 8f-1, config C4's algorithm)
def test_cp_als_golden():
    g = load_case("cp_als")
    runs = load_meta()["cases"]["cp_als"]["runs"]
    X = g["inp"]
    for name in ("r3_it1", "r3_it25", "r5_it4"):
        cores, errors = oracle.cp_als(X, runs[name]["R"], max_iter=runs[name]["max_iter"])
        assert max((a - b).abs().max().item() for a, b in zip(cores, g[name])) < 1e-7, name
        err = ((oracle.cp_to_dense(cores) - X).norm() / X.norm()).item()
        assert abs(err - runs[name]["relerr"]) < 1e-10 and abs(errors[-1].item() - err) < 1e-12
    Y = g["f32_inp"]
    cores, _ = oracle.cp_als(Y, 4, max_iter=6)
    d_o, d_r = oracle.cp_to_dense(cores), oracle.cp_to_dense(g["f32_r4_it6"])
    assert (d_o - d_r).norm() / d_r.norm() < 1e-4


# ------------------------------------------------------------------ producers
def test_producers_golden():
    g = load_case("producers_f64")
    prod = oracle.tt_mul(g["a"], g["b"])
    assert oracle.tt_ranks(prod) == [1, 6, 6, 6, 1] and _max_abs(prod, g["prod"]) == 0
    ts = [g[f"t{i}"] for i in range(5)]
    cores, Us = oracle.reduce_sum(ts, eps=1e-6)
    assert oracle.tt_ranks(cores) == g["red_ranks_tt"].tolist() and [c.shape[1] for c in cores] == g["red_ranks_tucker"].tolist()
    ref = g["red_dense"]
    assert (oracle.tucker_to_dense(cores, Us) - ref).norm() / ref.norm() < 1e-9
    cores3, _ = oracle.reduce_sum(ts, eps=0, rmax=3)
    ref3 = oracle.tt_to_dense(g["red3_cores"])
    assert (oracle.tt_to_dense(cores3) - ref3).norm() / ref3.norm() < 1e-10


SHIFT_SPECS = [(1, 2, 1e-3), (3, -2, 1e-6), (0, 4, "same"), (4, -4, 1e-2), (2, 1, 0.3)]


def test_consumers_shift_mode_and_ttmatrix():
    """tools.shift_mode and matrix.TTMatrix against the reference's recorded outputs."""
    g = load_case("consumers_f64")
    for k, (n, sh, eps) in enumerate(SHIFT_SPECS):
        out = oracle.shift_mode(g["g"], n, sh, eps=eps)
        want = g[f"shift{k}"]
        assert oracle.tt_ranks(out) == oracle.tt_ranks(want), (k, oracle.tt_ranks(out), oracle.tt_ranks(want))
        assert _max_abs(out, want) < 1e-10, k
    cores = oracle.ttmatrix_cores(g["m"], [20, 7], [11, 3, 4], [23, 2, 3])
    assert [tuple(c.shape) for c in cores] == [tuple(c.shape) for c in g["ttm_cores"]]
    assert _max_abs(cores, g["ttm_cores"]) < 1e-10
    assert (oracle.ttmatrix_to_dense(cores) - g["ttm_dense"]).abs().max() < 1e-12
    tsq = oracle.ttmatrix_cores(g["sq"], [36], [6, 5], [6, 5])
    assert abs(oracle.ttmatrix_trace(tsq).item() - g["tsq_trace"].item()) < 1e-12
    assert abs(oracle.ttmatrix_trace(tsq).item() - torch.trace(g["sq"]).item()) < 1e-11


def test_cp_variants_golden():
    """Batched CP-ALS and CP on a Tucker core (tensor.py:214-300) against the reference's recorded factors."""
    g = load_case("cp_variants_f64")
    cores, errors = oracle.cp_als_batch(g["batch_inp"], 4, max_iter=6, tol=-1.0)
    assert len(errors) == 6
    assert max((a - b).abs().max().item() for a, b in zip(cores, g["batch_r4_it6"])) < 1e-7
    fac, Us, _ = oracle.cp_on_tucker_core(g["tucker_inp"], 3, 4, g["tucker_init"], max_iter=5, tol=-1.0)
    assert max((a - b).abs().max().item() for a, b in zip(fac, g["tucker_cores"])) < 1e-7
    assert max((a - b).abs().max().item() for a, b in zip(Us, g["tucker_Us"])) < 1e-9
    dense = torch.einsum("abc,ia,jb,kc->ijk", oracle.cp_to_dense(fac), *Us)
    assert (dense - g["tucker_dense"]).abs().max() < 1e-9
