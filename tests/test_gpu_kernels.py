"""Kernel-level numerics of the C-ABI entry points on a real MI355X, each against a plain
float64 torch reference of the same operation (asymmetric operands, ragged sizes)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.float64]


def tol(dt, f32, f64):
    return f32 if dt == torch.float32 else f64


def _hip():
    from tntorch_amd import _hip

    _hip.lib()
    return _hip


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize(
    "M,N,K,tA,tB,B",
    [
        (64, 64, 64, False, False, 1),
        (64, 4096, 64, False, False, 3),
        (4096, 32, 64, False, False, 2),
        (64, 64, 2048, False, True, 4),   # Gram, left side
        (33, 33, 1000, True, False, 2),   # Gram, right side
        (17, 129, 5, False, False, 2),
        (1, 1, 1, False, False, 1),
        (70, 45, 19, True, True, 3),
        (32, 2048, 64, True, False, 2),   # projection V^T M
        (48, 48, 70000, True, False, 1),  # split-K path
    ],
)
def test_gemm(dt, M, N, K, tA, tB, B):
    h = _hip()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((B, K, M) if tA else (B, M, K), generator=g, dtype=torch.float64)
    Bm = torch.randn((B, N, K) if tB else (B, K, N), generator=g, dtype=torch.float64)
    ref = (A.transpose(1, 2) if tA else A) @ (Bm.transpose(1, 2) if tB else Bm)
    out = h.gemm(A.to(dt).cuda(), Bm.to(dt).cuda(), transA=tA, transB=tB).cpu().double()
    err = (out - ref).abs().max() / ref.abs().max()
    assert err < tol(dt, 3e-5 if K > 10000 else 5e-6, 1e-12), err


@pytest.mark.parametrize("M,N,K,tA,tB,B", [
    (256, 384, 1000, True, False, 2),     # TN, both operands mn-contiguous (the Gram orientation of a tall unfolding)
    (1024, 128, 512, False, False, 1),    # NN: M V1 (k-contiguous A, mn-contiguous B)
    (128, 256, 260, False, True, 3),      # NT: both k-contiguous
    (384, 128, 96, True, True, 1),        # TT
    (132, 260, 68, False, False, 2),      # ragged tile edges (multiples of 4, not of 128)
    (256, 256, 40000, True, False, 1),    # split-K with the XCD map (8 splits x 4 tiles)
    (2048, 256, 256, False, False, 1),    # row panels on the XCD map (16 x 2 tiles)
])
def test_gemm_big_tiles(M, N, K, tA, tB, B):
    """The 128 x 128-tile kernel (fp32, both output dimensions >= 128): every operand orientation, ragged edges, split-K,
    both workgroup maps -- against float64 torch."""
    h = _hip()
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = torch.randn((B, K, M) if tA else (B, M, K), generator=g, dtype=torch.float64)
    Bm = torch.randn((B, N, K) if tB else (B, K, N), generator=g, dtype=torch.float64)
    ref = (A.transpose(1, 2) if tA else A) @ (Bm.transpose(1, 2) if tB else Bm)
    out = h.gemm(A.float().cuda(), Bm.float().cuda(), transA=tA, transB=tB).cpu().double()
    assert (out - ref).abs().max() / ref.abs().max() < (3e-5 if K > 10000 else 5e-6)
    # epilogue scales and axpby on the big path
    rs = torch.rand(B, M, generator=g, dtype=torch.float64) + 0.5
    cs = torch.rand(B, N, generator=g, dtype=torch.float64) + 0.5
    out = h.gemm(A.float().cuda(), Bm.float().cuda(), transA=tA, transB=tB, rowscale=rs.float().cuda(), rowscale_mode=h.SCALE_DIV,
                 colscale=cs.float().cuda(), colscale_mode=h.SCALE_MUL).cpu().double()
    want = ref / rs[:, :, None] * cs[:, None, :]
    assert (out - want).abs().max() / want.abs().max() < (3e-5 if K > 10000 else 5e-6)
    C = torch.randn(B, M, N, generator=g, dtype=torch.float64)
    Cd = C.float().cuda()
    h.gemm_axpby(A.float().cuda(), Bm.float().cuda(), Cd, -0.5, 2.0, transA=tA, transB=tB)
    want = 2.0 * C - 0.5 * ref
    assert (Cd.cpu().double() - want).abs().max() / want.abs().max() < (3e-5 if K > 10000 else 5e-6)


@pytest.mark.parametrize("rows,n,B,side", [(5000, 256, 2, "right"), (70000, 1024, 1, "right"), (300, 384, 3, "left"), (4096, 128, 4, "right")])
def test_gemm_symmetric_products(rows, n, B, side):
    """A^T A / A A^T with the SAME tensor on both sides run only the tiles on and above the diagonal (mirrored on output,
    also through the split-K partials): equal to the float64 Gram matrix, exactly symmetric."""
    h = _hip()
    g = torch.Generator().manual_seed(rows + n)
    A = torch.randn((B, rows, n) if side == "right" else (B, n, rows), generator=g, dtype=torch.float64)
    Ad = A.float().cuda()
    if side == "right":
        out, ref = h.gemm(Ad, Ad, transA=True), A.transpose(1, 2) @ A
    else:
        out, ref = h.gemm(Ad, Ad, transB=True), A @ A.transpose(1, 2)
    out = out.cpu().double()
    assert (out - ref).abs().max() / ref.abs().max() < (3e-5 if rows > 10000 else 5e-6)
    assert (out - out.transpose(1, 2)).abs().max() == 0


@pytest.mark.parametrize("dt", DT)
def test_gemm_scales_and_views(dt):
    h = _hip()
    g = torch.Generator().manual_seed(5)
    A = torch.randn(2, 40, 64, generator=g, dtype=torch.float64)   # used as A[:, :, :24]^T: 24 x 40
    Bm = torch.randn(2, 40, 300, generator=g, dtype=torch.float64)
    rs = torch.rand(2, 64, generator=g, dtype=torch.float64) + 0.5
    cs = torch.rand(2, 300, generator=g, dtype=torch.float64) + 0.5
    ref = (A[:, :, :24].transpose(1, 2) @ Bm) / rs[:, :24, None] * cs[:, None, :]
    Ad, Bd = A.to(dt).cuda(), Bm.to(dt).cuda()
    out = h.gemm(Ad[:, :, :24], Bd, transA=True, rowscale=rs.to(dt).cuda(), rowscale_mode=h.SCALE_DIV,
                 colscale=cs.to(dt).cuda(), colscale_mode=h.SCALE_MUL).cpu().double()
    assert (out - ref).abs().max() / ref.abs().max() < tol(dt, 5e-6, 1e-12)
    # division by an exactly zero scale yields 0 (guarded), not inf/nan
    rs0 = rs.clone(); rs0[:, 3] = 0
    out = h.gemm(Ad[:, :, :24], Bd, transA=True, rowscale=rs0.to(dt).cuda(), rowscale_mode=h.SCALE_DIV).cpu()
    assert torch.isfinite(out).all() and (out[:, 3] == 0).all()


QR_SHAPES = [(64, 64), (4096, 64), (300, 17), (16, 16), (10, 40), (256, 16), (257, 16), (1000, 33), (5000, 64),
             (1, 1), (1, 7), (7, 1), (513, 32), (70000, 8)]


@pytest.fixture(params=[1, 0], ids=["pair-steps", "single-steps"])
def qr_variant(request):
    """TTR_KNOB_QR_PANEL: every QR test runs under the default panel kernel (pair steps) and the round-1 variant."""
    h = _hip()
    h.set_knob(h.KNOB_QR_PANEL, request.param)
    yield request.param
    h.set_knob(h.KNOB_QR_PANEL, 1)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("m,n", QR_SHAPES)
def test_qr(dt, m, n, qr_variant):
    h = _hip()
    g = torch.Generator().manual_seed(m * 31 + n)
    B = 2 if m * n < 400000 else 1
    A = torch.randn(B, m, n, generator=g, dtype=torch.float64).to(dt)
    Q, R = h.qr(A.cuda())
    Q, R = Q.cpu().double(), R.cpu().double()
    k = min(m, n)
    assert Q.shape == (B, m, k) and R.shape == (B, k, n)
    eye = torch.eye(k, dtype=torch.float64)
    assert (Q.transpose(1, 2) @ Q - eye).abs().max() < tol(dt, 2e-5, 5e-13)
    assert (Q @ R - A.double()).abs().max() / A.abs().max() < tol(dt, 1e-5, 1e-13)
    assert R.tril(-1).abs().max() == 0
    # same factorisation as LAPACK up to the per-row sign of R
    Rref = torch.linalg.qr(A.double())[1]
    assert (R.abs() - Rref.abs()).abs().max() / Rref.abs().max() < tol(dt, 2e-4, 1e-11)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("k,Rin,I,n", [(64, 64, 64, 64), (10, 7, 9, 12), (64, 64, 5, 33), (33, 64, 130, 64), (1, 1, 40, 8)])
def test_qr_pushed(dt, k, Rin, I, n, qr_variant):
    """Fused push + QR: factor the left unfolding of Rm @ core without forming it; apply gives Q [C; 0]."""
    h = _hip()
    g = torch.Generator().manual_seed(k * 1000 + Rin * 100 + I * 10 + n)
    B = 2
    Rm = torch.randn(B, k, Rin, generator=g, dtype=torch.float64).to(dt)
    core = torch.randn(B, Rin, I, n, generator=g, dtype=torch.float64).to(dt)
    P = (Rm.double() @ core.double().reshape(B, Rin, I * n)).reshape(B, k * I, n)
    f = h.qr_factor_pushed(Rm.cuda(), core.cuda())
    Q = h.qr_apply(f).cpu().double()
    R = f.R.cpu().double()
    kq = min(k * I, n)
    assert Q.shape == (B, k * I, kq) and R.shape == (B, kq, n)
    assert (Q.transpose(1, 2) @ Q - torch.eye(kq, dtype=torch.float64)).abs().max() < tol(dt, 3e-5, 1e-12)
    assert (Q @ R - P).abs().max() / P.abs().max() < tol(dt, 2e-5, 1e-12)
    Rref = torch.linalg.qr(P)[1]
    assert (R.abs() - Rref.abs()).abs().max() / Rref.abs().max() < tol(dt, 3e-4, 1e-10)
    C = torch.randn(B, kq, min(kq, 5), generator=g, dtype=torch.float64).to(dt)
    out = h.qr_apply(f, C.cuda()).cpu().double()
    assert (out - Q @ C.double()).abs().max() < tol(dt, 3e-5, 1e-12)


@pytest.mark.parametrize("shape", [(4096, 64), (64, 64), (33, 5), (700, 20), (9000, 48)])
def test_qr_factor_expo_is_the_plain_factorisation_at_another_exponent(shape):
    """ttr_qr_factor_expo (ABI 11): R comes back as R 2^-e with the exponent added to expo_acc -- the SAME bits as ttr_qr_factor's R
    once scaled back (exact powers of two), the same reflectors (ttr_qr_apply unchanged), e such that what is returned is O(1);
    items at 1e-20 and 1e+15, a zero item."""
    h = _hip()
    m, n = shape
    g = torch.Generator().manual_seed(m + n)
    A = torch.randn(4, m, n, generator=g, dtype=torch.float32)
    A[1] *= 1e-20
    A[2] *= 1e15
    A[3] = 0
    Ad = A.cuda()
    plain = h.qr_factor(Ad)
    expo = torch.tensor([0, 5, -3, 7], dtype=torch.int32, device="cuda")
    f = h.qr_factor(Ad, expo_acc=expo)
    e = (expo.cpu() - torch.tensor([0, 5, -3, 7], dtype=torch.int32)).double()
    assert int(e[3]) == 0 and torch.equal(f.R[3], plain.R[3])
    back = torch.ldexp(f.R.cpu().double(), e[:, None, None].expand_as(f.R).to(torch.int32))
    assert torch.equal(back.to(torch.float32), plain.R.cpu())
    mx = f.R[:3].abs().amax(dim=(1, 2)).cpu()
    assert (mx > 2.0 ** -12).all() and (mx < 2.0 ** 12).all()
    assert torch.equal(h.qr_apply(f), h.qr_apply(plain))


@pytest.mark.parametrize("k,Rin,I,n", [(64, 64, 64, 64), (64, 64, 8, 64), (64, 64, 4, 33), (10, 7, 9, 12), (33, 64, 130, 64)])
def test_qr_factor_pushed_expo_is_the_plain_factorisation_at_another_exponent(k, Rin, I, n):
    """ttr_qr_factor_pushed_expo: as above for the fused push -- TSQR trees with a top level of their own (the kernel keeps the
    exponent) and single-level ones (I <= 8: a normalisation launch inside the entry)."""
    h = _hip()
    g = torch.Generator().manual_seed(k * 1000 + Rin * 100 + I * 10 + n)
    B = 3
    Rm = torch.randn(B, k, Rin, generator=g, dtype=torch.float32)
    core = torch.randn(B, Rin, I, n, generator=g, dtype=torch.float32)
    Rm[1] *= 1e-18
    core[2] *= 1e12
    plain = h.qr_factor_pushed(Rm.cuda(), core.cuda())
    expo = torch.tensor([1, 0, -2], dtype=torch.int32, device="cuda")
    f = h.qr_factor_pushed(Rm.cuda(), core.cuda(), expo_acc=expo)
    e = (expo.cpu() - torch.tensor([1, 0, -2], dtype=torch.int32)).double()
    back = torch.ldexp(f.R.cpu().double(), e[:, None, None].expand_as(f.R).to(torch.int32))
    assert torch.equal(back.to(torch.float32), plain.R.cpu())
    mx = f.R.abs().amax(dim=(1, 2)).cpu()
    assert (mx > 2.0 ** -12).all() and (mx < 2.0 ** 12).all()
    assert torch.equal(h.qr_apply(f), h.qr_apply(plain))
    if plain.rows32 is not None:
        assert torch.equal(f.rows32, plain.rows32)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("pack", [0, 1, 2, 3])
@pytest.mark.parametrize("I,mixed", [(64, False), (128, False), (64, True), (24, False)])
def test_qr_pushed_rank_deficient_R_packs_rows(dt, I, mixed, pack):
    """Fused push with an R factor of numerical rank 32 of 64 (rows 32 .. 63 at the rounding level: the L2R sweep of a
    rank-inflated train, t = g + g): the level-0 blocks pack two mode indices per wave and drop the pushed rows (kk >= 32, i)
    as zeros (TTR_KNOB_QR_PACK), the trailing panels are H = I (TTR_KNOB_QR_RANK_SKIP).  Q R still reproduces the product to
    the rounding level, Q stays orthonormal on its range, and a batch that mixes a rank-deficient and a full-rank item (decided
    per item on the device) handles both; I = 24 (3 blocks: odd) never packs."""
    h = _hip()
    h.set_knob(h.KNOB_QR_PACK, pack)   # (3 = default; 1 / 2 = the item-major block pairings; the conftest fixture restores the default)
    g = torch.Generator().manual_seed(I + 7 * int(mixed))
    B, k, Rin, n = 3, 64, 64, 64
    top = torch.triu(torch.randn(B, 32, 64, generator=g, dtype=torch.float64))
    Rm = torch.cat([top, 1e-9 * torch.triu(torch.randn(B, 32, 64, generator=g, dtype=torch.float64), diagonal=32)], dim=1)
    if mixed:
        Rm[1] = torch.triu(torch.randn(64, 64, generator=g, dtype=torch.float64))
    Rm = Rm.to(dt)
    # a core with the block structure of g + g: rows 32.. only reach columns 32.. (rank-32 unfolding for the deficient items)
    gcore = torch.randn(B, 32, I, 32, generator=g, dtype=torch.float64)
    z = torch.zeros_like(gcore)
    core = torch.cat([torch.cat([gcore, z], dim=-1), torch.cat([z, gcore], dim=-1)], dim=1).to(dt)
    P = (Rm.double() @ core.double().reshape(B, Rin, I * n)).reshape(B, k * I, n)
    f = h.qr_factor_pushed(Rm.cuda(), core.cuda())
    R = f.R.cpu().double()
    C = torch.eye(64, dtype=torch.float64)[None, :, :32].repeat(B, 1, 1).to(dt)
    C2 = torch.eye(64, dtype=torch.float64)[None, :, 32:].repeat(B, 1, 1).to(dt)
    Q = torch.cat([h.qr_apply(f, C.cuda()).cpu().double(), h.qr_apply(f, C2.cuda()).cpu().double()], dim=2)   # Q [I; 0], 32 columns at a time
    assert torch.isfinite(Q).all() and torch.isfinite(R).all()
    assert (Q @ R - P).abs().max() / P.abs().max() < tol(dt, 2e-5, 1e-8)        # (fp64: the 1e-9 rows are dropped by design)
    for bi in range(B):
        r = 64 if (mixed and bi == 1) else 32
        Qr = Q[bi][:, :r]
        assert (Qr.T @ Qr - torch.eye(r, dtype=torch.float64)).abs().max() < tol(dt, 3e-5, 1e-12)
    assert torch.equal(torch.tril(f.R, diagonal=-1), torch.zeros_like(f.R))


@pytest.mark.parametrize("dt", DT)
def test_qr_rank_deficient(dt, qr_variant):
    """Left unfolding of g+g: exactly rank-deficient; Q must still be orthonormal."""
    h = _hip()
    g = torch.Generator().manual_seed(3)
    G = torch.randn(1, 2048, 16, generator=g, dtype=torch.float64)
    A = torch.cat([G, G, torch.zeros(1, 2048, 4, dtype=torch.float64), 2 * G[:, :, :8]], dim=2).to(dt)  # 2048 x 44, rank 16
    Q, R = h.qr(A.cuda())
    Q, R = Q.cpu().double(), R.cpu().double()
    assert (Q.transpose(1, 2) @ Q - torch.eye(44, dtype=torch.float64)).abs().max() < tol(dt, 2e-5, 5e-13)
    assert (Q @ R - A.double()).abs().max() / A.abs().max() < tol(dt, 1e-5, 1e-13)
    A0 = torch.zeros(1, 500, 20, dtype=dt)
    Q, R = h.qr(A0.cuda())
    assert (R == 0).all() and (Q.cpu().double().transpose(1, 2) @ Q.cpu().double() - torch.eye(20, dtype=torch.float64)).abs().max() < 1e-6


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("solver", [1, 2])
@pytest.mark.parametrize("n", [1, 2, 5, 16, 17, 18, 33, 49, 63, 64, 100, 200])  # 17, 18, 33, 49: reflector-block boundaries of the two-wave kernel
def test_eigh(dt, n, solver):
    """solver 1 = Jacobi (LDS up to ~100, L2-resident workspace above), 2 = tridiagonal QL (n <= 64, else Jacobi)."""
    h = _hip()
    g = torch.Generator().manual_seed(n)
    B = 3
    Mx = torch.randn(B, n, 3 * n + 1, generator=g, dtype=torch.float64)
    Mx = Mx * (0.7 ** torch.arange(n, dtype=torch.float64))[None, :, None]  # graded rows
    G = (Mx @ Mx.transpose(1, 2)).to(dt)
    V, sig, info = h.eigh_trunc(G.cuda(), h.EIG_RAW, False, 0.0, n, abs_floor=solver)
    V, sig, info = V.cpu().double(), sig.cpu().double(), info.cpu()
    wref = torch.linalg.eigvalsh(G.double()).flip(-1).clamp_min(0)
    assert (info == n).all()
    assert (sig[:, :-1] >= sig[:, 1:]).all()
    assert ((sig**2 - wref).abs().max(dim=1).values / wref[:, 0]).max() < tol(dt, 2e-6, 1e-13)
    eye = torch.eye(n, dtype=torch.float64)
    assert (V.transpose(1, 2) @ V - eye).abs().max() < tol(dt, 3e-5, 1e-12)
    resid = (G.double() @ V - V * (sig**2)[:, None, :]).abs().max() / wref.max()
    assert resid < tol(dt, 2e-5, 1e-12)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("solver", [1, 2])
@pytest.mark.parametrize("n", [7, 40, 64])
def test_eigh_match_diag(dt, n, solver):
    """TTR_EIG_MATCH_DIAG: same eigenpairs, but column order follows G's diagonal, so a nearly diagonal G
    gives V ~ I (up to sign) -- the property the block-Jacobi driver needs from its pair solver."""
    h = _hip()
    g = torch.Generator().manual_seed(100 + n)
    d = (torch.stack([torch.randperm(n, generator=g) for _ in range(2)]) + 1).double()   # gaps >= 1
    E = torch.randn(2, n, n, generator=g, dtype=torch.float64) * 1e-3
    G = (torch.diag_embed(d) + E + E.transpose(1, 2)).to(dt)
    V, sig, info = h.eigh_trunc(G.cuda(), h.EIG_MATCH_DIAG, False, 0.0, n, abs_floor=solver)
    V, sig = V.cpu().double(), sig.cpu().double()
    eye = torch.eye(n, dtype=torch.float64)
    assert (V.abs() - eye).abs().max() < 2e-2                      # near-identity, no sorting swaps
    assert (sig**2 - d).abs().max() < 2e-2                         # sigma follows the column order
    resid = (G.double() @ V - V * (sig**2)[:, None, :]).abs().max() / d.max()
    assert resid < tol(dt, 2e-5, 1e-12)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("n,B,solver", [(128, 2, 2), (150, 1, 2), (288, 1, 2), (512, 1, 2), (128, 2, 1), (150, 1, 1),
                                        (127, 2, 2), (211, 1, 2), (122, 1, 1),   # these three: padded (no block width divides n)
                                        (2048, 1, 2)])
def test_eigh_block_jacobi(dt, n, B, solver):
    """Large-n driver (_hipops._eigh_any): block Jacobi over pair problems + sorted epilogue.
    solver 2: tridiagonal pair solver, absolute stop test; solver 1: Jacobi pair solver, relative stop test."""
    _block_jacobi_case(dt, n, B, solver, graded=True)


@pytest.mark.parametrize("n,B", [(1024, 1), (256, 4)])
def test_eigh_block_jacobi_flat_spectrum(n, B):
    """A flat spectrum (Gram matrix of an i.i.d. matrix: what `randn` data gives the dense TT-SVD) converges slowly at
    first -- the off-diagonal mass of the n = 1024 problem is still 8 % after four sweeps, and the round-2 stop rule ("no
    longer halving") ended the absolute mode there.  The driver has to go on to the rounding floor."""
    _block_jacobi_case(torch.float32, n, B, 2, graded=False)


def _block_jacobi_case(dt, n, B, solver, graded):
    from tntorch_amd import _hipops
    h = _hip()
    g = torch.Generator().manual_seed(n)
    Mx = torch.randn(B, n, 2 * n + 1, generator=g, dtype=torch.float64)
    if graded:
        Mx = Mx * torch.logspace(0, -3, n, dtype=torch.float64)[None, :, None]
    G = (Mx @ Mx.transpose(1, 2)).to(dt)
    V, sig, info = _hipops._eigh_any(G.cuda(), h.EIG_RAW, False, 0.0, n, solver)
    V, sig, info = V.cpu().double(), sig.cpu().double(), info.cpu()
    wref = torch.linalg.eigvalsh(G.double()).flip(-1).clamp_min(0)
    assert (info == n).all()
    assert (sig[:, :-1] >= sig[:, 1:]).all()
    grow = max(1.0, (n / 512) ** 0.5)   # rounding of the n-term Rayleigh quotients / residuals grows like sqrt(n)
    assert ((sig**2 - wref).abs().max(dim=1).values / wref[:, 0]).max() < grow * tol(dt, 4e-6, 1e-13)
    eye = torch.eye(n, dtype=torch.float64)
    assert (V.transpose(1, 2) @ V - eye).abs().max() < grow * tol(dt, 3e-5, 1e-12)
    resid = (G.double() @ V - V * (sig**2)[:, None, :]).abs().max() / wref.max()
    assert resid < grow * tol(dt, 3e-5, 1e-12)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("relative", [False, True])
@pytest.mark.parametrize("n,B", [(256, 3), (1024, 1), (48, 5)])
def test_block_jacobi_device_loop_matches_host_loop(dt, relative, n, B):
    """The device-resident sweep loop (ttr_bj_solve / ttr_bj_apply / ttr_bj_control: pair table instead of moved blocks,
    convergence word on the device, nothing read back) against the round-2 host-driven loop on the same matrices: same
    eigenvalues, orthogonal V, small residual; C3's bond size (256, batched), C1's (1024) and a small block width (b = 24)."""
    from tntorch_amd import _hipops
    g = torch.Generator().manual_seed(n + B)
    Mx = torch.randn(B, n, n + 7, generator=g, dtype=torch.float64) * torch.logspace(0, -2, n, dtype=torch.float64)[None, :, None]
    G = (Mx @ Mx.transpose(1, 2))
    if relative:   # pass 2 of 'svd': nearly diagonal, graded
        w, Q = torch.linalg.eigh(G)
        E = 1e-4 * torch.randn(B, n, n, generator=g, dtype=torch.float64)
        G = torch.diag_embed(w) + (E + E.transpose(1, 2)) * w.sqrt()[:, :, None] * w.sqrt()[:, None, :]
    G = G.to(dt).cuda()
    out = {}
    for on_dev in (True, False):
        _hipops.BLOCK_JACOBI_ON_DEVICE = on_dev
        try:
            V, d = _hipops.eigh_block_jacobi(G, relative=relative)
        finally:
            _hipops.BLOCK_JACOBI_ON_DEVICE = True
        out[on_dev] = (V.cpu().double(), d.cpu().double())
    wref = torch.linalg.eigvalsh(G.cpu().double())
    grow = max(1.0, (n / 512) ** 0.5)
    for on_dev, (V, d) in out.items():
        ds = d.sort(dim=1).values
        # (relative mode returns diag(G) after all in-place updates, without the Rayleigh refinement: 3e-13 at n = 1024)
        assert ((ds - wref).abs().max(dim=1).values / wref[:, -1]).max() < grow * tol(dt, 4e-6, 3e-13 if relative else 1e-13), on_dev
        assert (V.transpose(1, 2) @ V - torch.eye(n, dtype=torch.float64)).abs().max() < grow * tol(dt, 3e-5, 1e-12), on_dev
        resid = (G.cpu().double() @ V - V * d[:, None, :]).abs().max() / wref.max()
        assert resid < grow * tol(dt, 3e-5, 1e-12), on_dev


@pytest.mark.parametrize("dt", DT)
def test_eigh_rank_rule(dt):
    """Rank rule of round.py:147-158 on a diagonal matrix with known sigma."""
    h = _hip()
    sig = torch.tensor([8.0, 4.0, 2.0, 1.0, 0.5, 0.25], dtype=torch.float64)
    perm = torch.tensor([3, 0, 5, 1, 4, 2])
    G = torch.diag(sig[perm] ** 2).to(dt)[None]
    cases = [
        (0.0, 100, 6), (0.25, 100, 5), (0.2499, 100, 6), (0.56, 100, 4),
        (100.0, 100, 1), (0.0, 3, 3), (0.6, 2, 2),
    ]
    for delta, rmax, expect in cases:
        V, s, info = h.eigh_trunc(G.cuda(), h.EIG_RAW, True, float(delta) ** 2, rmax)
        assert int(info[0]) == expect, (delta, rmax, int(info[0]), expect)
        assert torch.allclose(s.cpu().double()[0], sig, rtol=1e-6)
        # column j of V is the (signed) unit vector of the j-th largest eigenvalue
        expect_V = torch.zeros(6, 6, dtype=torch.float64)
        expect_V[torch.arange(6), perm] = 1.0
        assert torch.allclose(V.cpu().double()[0].abs(), expect_V, atol=1e-6)
    # batch mode: delta ignored
    V, s, info = h.eigh_trunc(G.cuda(), h.EIG_RAW, False, 1e9, 4)
    assert int(info[0]) == 4
    # zero guard
    V, s, info = h.eigh_trunc(torch.zeros(1, 5, 5, dtype=dt).cuda(), h.EIG_RAW, True, 0.0, 9)
    assert int(info[0]) == 0
    # reference clamp: a negative eigenvalue becomes sigma = 1e-4 (round.py:118-119)
    Gn = torch.diag(torch.tensor([4.0, -1e-9, 1.0], dtype=torch.float64)).to(dt)[None]
    V, s, info = h.eigh_trunc(Gn.cuda(), h.EIG_REF, True, 0.0, 9)
    assert torch.allclose(s.cpu().double()[0], torch.tensor([2.0, 1.0, 1e-4], dtype=torch.float64), rtol=1e-5)
    V, s, info = h.eigh_trunc(Gn.cuda(), h.EIG_RAW, True, 0.0, 9)
    assert s.cpu()[0, 2] == 0


@pytest.mark.parametrize("dt", DT)
def test_norm_and_scale(dt):
    h = _hip()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(5, 3, 777, generator=g, dtype=torch.float64)
    out = h.norm(x.to(dt).cuda()).cpu().double()
    assert torch.allclose(out, x.reshape(5, -1).norm(dim=1), rtol=tol(dt, 1e-6, 1e-14))
    s = torch.rand(5, 777, generator=g, dtype=torch.float64) + 0.1
    o = h.scale_cols(x.to(dt).cuda(), s.to(dt).cuda(), h.SCALE_MUL).cpu().double()
    assert torch.allclose(o, x * s[:, None, :], rtol=tol(dt, 1e-6, 1e-14))
    o = h.scale_cols(x.to(dt).cuda(), s.to(dt).cuda(), h.SCALE_DIV).cpu().double()
    assert torch.allclose(o, x / s[:, None, :], rtol=tol(dt, 1e-6, 1e-14))


@pytest.mark.parametrize("dt", DT)
def test_gemm_axpby(dt):
    h = _hip()
    g = torch.Generator().manual_seed(3)
    for (M, N, K, B, tA, tB, al, be) in [(70, 33, 129, 2, False, False, -1.0, 1.0), (64, 64, 5000, 1, True, False, 0.5, 0.0),
                                          (17, 90, 40, 3, False, True, 2.0, -0.25)]:
        A = torch.randn(B, *((K, M) if tA else (M, K)), generator=g, dtype=torch.float64)
        Bm = torch.randn(B, *((N, K) if tB else (K, N)), generator=g, dtype=torch.float64)
        C = torch.randn(B, M, N, generator=g, dtype=torch.float64)
        ref = be * C + al * ((A.transpose(1, 2) if tA else A) @ (Bm.transpose(1, 2) if tB else Bm))
        Cd = C.to(dt).cuda()
        out = h.gemm_axpby(A.to(dt).cuda(), Bm.to(dt).cuda(), Cd, al, be, transA=tA, transB=tB)
        assert out.data_ptr() == Cd.data_ptr()
        assert (out.cpu().double() - ref).abs().max() / ref.abs().max() < tol(dt, 2e-6, 1e-14)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("m,n,B", [(1000, 100, 2), (300, 200, 1), (128, 128, 1), (40, 150, 2), (100, 260, 1)])
def test_qr_blocked(dt, m, n, B):
    """QR above the 64-column TSQR panel: block Gram-Schmidt with re-orthogonalisation around the kernel."""
    from tntorch_amd import _hipops
    g = torch.Generator().manual_seed(m + n)
    A = torch.randn(B, m, n, generator=g, dtype=torch.float64).to(dt)
    Q, R = _hipops.qr(A.cuda())
    Q, R = Q.cpu().double(), R.cpu().double()
    k = min(m, n)
    assert Q.shape == (B, m, k) and R.shape == (B, k, n)
    assert (Q.transpose(1, 2) @ Q - torch.eye(k, dtype=torch.float64)).abs().max() < tol(dt, 2e-5, 1e-12)
    assert (Q @ R - A.double()).abs().max() / A.abs().max() < tol(dt, 2e-5, 1e-12)
    assert torch.tril(R[:, :, :k], -1).abs().max() < tol(dt, 1e-5, 1e-12)


@pytest.mark.parametrize("dt", DT)
def test_qr_blocked_rank_deficient(dt):
    """Rank-96 matrix with 160 columns: Q must stay orthonormal (second Gram-Schmidt pass on the normalised panel)."""
    from tntorch_amd import _hipops
    g = torch.Generator().manual_seed(11)
    A = (torch.randn(1, 700, 96, generator=g, dtype=torch.float64) @ torch.randn(1, 96, 160, generator=g, dtype=torch.float64)).to(dt)
    Q, R = _hipops.qr(A.cuda())
    Q, R = Q.cpu().double(), R.cpu().double()
    assert (Q.transpose(1, 2) @ Q - torch.eye(160, dtype=torch.float64)).abs().max() < tol(dt, 5e-5, 1e-11)
    assert (Q @ R - A.double()).abs().max() / A.abs().max() < tol(dt, 2e-5, 1e-12)


def test_qr_blocked_perturbs_only_collapsing_items(monkeypatch):
    """Full-rank inputs are factored as they are (rounds 1-2 added seeded 8-eps noise to every panel of everything); the noise is
    only drawn when some panel's projected remainder collapses, and only the collapsing items of the batch receive it."""
    from tntorch_amd import _hipops
    g = torch.Generator().manual_seed(21)
    full = torch.randn(2, 500, 150, generator=g, dtype=torch.float64)
    deficient = full.clone()
    deficient[:, :, 100:] = 0.0
    mixed = torch.stack([full[0], deficient[1]])  # item 0 keeps its rank in every panel, item 1 collapses in panels 2 and 3
    drawn = []
    orig = torch.randn
    monkeypatch.setattr(torch, "randn", lambda *a, **k: (drawn.append(a), orig(*a, **k))[1])
    Qf, Rf = _hipops.qr(full.cuda())
    assert not drawn
    Qm, Rm = _hipops.qr(mixed.cuda())
    assert drawn
    monkeypatch.undo()
    assert torch.equal(Qm[0], Qf[0]) and torch.equal(Rm[0], Rf[0])  # the full-rank item of the mixed batch: bit-identical
    for Q, R, X in ((Qf, Rf, full), (Qm, Rm, mixed)):
        Q, R = Q.cpu(), R.cpu()
        assert (Q.transpose(1, 2) @ Q - torch.eye(150, dtype=torch.float64)).abs().max() < 1e-11
        assert (Q @ R - X).abs().max() / X.abs().max() < 1e-12


@pytest.mark.parametrize("dt", DT)
def test_krp_contract_and_hadamard(dt):
    """ttr_krp_contract: out[p,q,r] = sum_j T[p,j,q,r] B[j,r] (trailing mode Q = 1, leading mode P = 1, general)."""
    h = _hip()
    g = torch.Generator().manual_seed(5)
    for (P, J, Q, R) in [(37, 19, 1, 32), (1, 23, 301, 32), (5, 7, 3, 5), (70000, 3, 1, 4), (1, 256, 1000, 32), (3, 1, 2, 64), (2, 300, 9, 33)]:
        T = torch.randn(P, J, Q, R, generator=g, dtype=torch.float64)
        B = torch.randn(J, R, generator=g, dtype=torch.float64)
        ref = torch.einsum("pjqr,jr->pqr", T, B)
        out = h.krp_contract(T.to(dt).cuda(), B.to(dt).cuda()).cpu().double()
        assert out.shape == (P, Q, R)
        assert (out - ref).abs().max() / ref.abs().max() < tol(dt, 5e-6, 1e-13), (P, J, Q, R)
    a = torch.randn(33, 65, generator=g, dtype=torch.float64)
    b = torch.randn(33, 65, generator=g, dtype=torch.float64)
    assert torch.allclose(h.hadamard(a.to(dt).cuda(), b.to(dt).cuda()).cpu().double(), (a.to(dt) * b.to(dt)).double(), rtol=0, atol=0)


@pytest.mark.parametrize("dt", DT)
def test_core_kron(dt):
    """ttr_core_kron vs the reference's broadcasting formula (tensor.py:2309-2320)."""
    h = _hip()
    g = torch.Generator().manual_seed(8)
    for (B, R1, S1, I, R2, S2) in [(1, 3, 2, 7, 4, 5), (3, 1, 1, 9, 6, 2), (2, 8, 8, 33, 8, 8), (1, 5, 4, 1, 1, 1)]:
        a = torch.randn(B, R1, I, R2, generator=g, dtype=torch.float64).to(dt)
        c = torch.randn(B, S1, I, S2, generator=g, dtype=torch.float64).to(dt)
        ref = (a[:, :, None, :, :, None] * c[:, None, :, :, None, :]).reshape(B, R1 * S1, I, R2 * S2)
        out = h.core_kron(a.cuda(), c.cuda()).cpu()
        assert torch.equal(out, ref)


@pytest.mark.parametrize("dt", DT)
def test_qr_blocked_exact_dependence(dt):
    """Exactly zero and exactly repeated column blocks (what block-diagonal TT sums produce): the panels' zero
    remainders must not be completed with colliding unit vectors.  Square, wide and all-zero inputs."""
    from tntorch_amd import _hipops
    g = torch.Generator().manual_seed(12)
    base = torch.randn(1, 136, 40, generator=g, dtype=torch.float64)
    A = torch.cat([base, torch.zeros(1, 136, 30, dtype=torch.float64), 0.5 * base, torch.zeros(1, 136, 26, dtype=torch.float64)], dim=2)
    for X in (A, torch.cat([A, base[:, :, :14]], dim=2), torch.zeros(1, 136, 136, dtype=torch.float64), A[:, :100]):
        X = X.to(dt)
        Q, R = _hipops.qr(X.cuda())
        Q, R = Q.cpu().double(), R.cpu().double()
        k = min(X.shape[1], X.shape[2])
        assert (Q.transpose(1, 2) @ Q - torch.eye(k, dtype=torch.float64)).abs().max() < tol(dt, 5e-5, 1e-11)
        # (an all-zero input comes back with R at the perturbation floor, ~1e-15 in fp32, instead of exact zeros)
        assert (Q @ R - X.double()).abs().max() <= tol(dt, 2e-5, 1e-12) * float(X.abs().max()) + 1e-13


def test_unsupported_shapes_raise():
    h = _hip()
    with pytest.raises(NotImplementedError):
        h.qr(torch.randn(1, 300, h.max_qr_cols(torch.float32) + 1).cuda())
    with pytest.raises(TypeError):
        h.qr(torch.randn(1, 8, 4).cuda().half())


# ------------------------------------------------------------------ round 2 additions
def test_qr_pair_steps_match_single_steps():
    """The pair-step panel kernel (v_permlane-swap reductions, two reflectors per step) against the one-reflector
    kernel: the same Householder factorisation up to summation order."""
    h = _hip()
    g = torch.Generator().manual_seed(5)
    A = torch.randn(4, 4096, 64, generator=g, dtype=torch.float32)
    A[1, :, 20:40] = A[1, :, :20]            # exactly dependent columns
    A[2, :, 7] = 0
    A = A.cuda()
    out = {}
    for v in (0, 1):
        h.set_knob(h.KNOB_QR_PANEL, v)
        try:
            f = h.qr_factor(A)
            out[v] = (f.R.clone(), h.qr_apply(f))
        finally:
            h.set_knob(h.KNOB_QR_PANEL, 1)
    assert (out[0][0][0] - out[1][0][0]).abs().max() / out[0][0][0].abs().max() < 1e-5   # generic item: same R
    for b in range(4):
        Q = out[1][1][b].double()
        assert (Q.T @ Q - torch.eye(64, dtype=torch.float64, device="cuda")).abs().max() < 3e-5
        assert (Q @ out[1][0][b].double() - A[b].double()).abs().max() / A[b].abs().max() < 1e-5


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("R,n,B", [(64, 2048, 3), (64, 4096, 1), (37, 1000, 2), (5, 33, 2), (1, 7, 1), (64, 70000, 1), (48, 16, 300)])
def test_sweep_gram_rotgram_project(dt, R, n, B):
    """ttr_rowgram / ttr_rotgram / ttr_project against float64 torch."""
    h = _hip()
    g = torch.Generator().manual_seed(R * 7 + n)
    M = torch.randn(B, R, n, generator=g, dtype=torch.float64).to(dt)
    V1 = torch.linalg.qr(torch.randn(B, R, R, generator=g, dtype=torch.float64))[0].to(dt)
    V2 = torch.linalg.qr(torch.randn(B, R, R, generator=g, dtype=torch.float64))[0].to(dt)
    sig = (torch.rand(B, R, generator=g, dtype=torch.float64) + 0.5).to(dt)
    Md, V1d, V2d = M.double(), V1.double(), V2.double()
    t = tol(dt, 2e-5, 1e-12)
    G = h.rowgram(M.cuda()).cpu().double().sum(dim=1)
    Gr = Md @ Md.transpose(1, 2)
    assert (G - Gr).abs().max() / Gr.abs().max() < t
    G2 = h.rowgram(M.cuda(), V1.cuda()).cpu().double().sum(dim=1)
    Mw = V1d.transpose(1, 2) @ Md
    G2r = Mw @ Mw.transpose(1, 2)
    assert (G2 - G2r).abs().max() / G2r.abs().max() < t
    assert (G2 - G2.transpose(1, 2)).abs().max() == 0                       # both triangles written from one value
    ro = max(1, R // 2)
    U = V1d @ V2d[:, :, :ro]
    right, left = h.project(M.cuda(), V1.cuda(), V2.cuda(), sig.cuda(), ro, scale_right=True)
    assert (right.cpu().double() - (U.transpose(1, 2) @ Md) / sig.double()[:, :ro, None]).abs().max() / Md.abs().max() < 4 * t
    assert (left.cpu().double() - U * sig.double()[:, None, :ro]).abs().max() < t
    right, left = h.project(M.cuda(), None, V2.cuda(), sig.cuda(), ro, scale_right=False)
    assert (right.cpu().double() - V2d[:, :, :ro].transpose(1, 2) @ Md).abs().max() / Md.abs().max() < 4 * t
    assert (left.cpu().double() - V2d[:, :, :ro]).abs().max() < t
    # a strided view of a larger tensor (leading dimension > n), and a zero sigma (TTR_SCALE_DIV: 0, not inf)
    big = torch.randn(B, R, n + 5, generator=g, dtype=torch.float64).to(dt).cuda()
    Gv = h.rowgram(big[:, :, 2:2 + n]).cpu().double().sum(dim=1)
    ref = big[:, :, 2:2 + n].cpu().double()
    assert (Gv - ref @ ref.transpose(1, 2)).abs().max() / Gr.abs().max() < t
    sig0 = sig.clone()
    sig0[:, 0] = 0
    right, _ = h.project(M.cuda(), None, V2.cuda(), sig0.cuda(), ro, scale_right=True)
    assert (right[:, 0] == 0).all() and torch.isfinite(right).all()


@pytest.mark.parametrize("dt", DT)
def test_pow2_normalize_and_scale_batch(dt):
    h = _hip()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(5, 7, 33, generator=g, dtype=torch.float64).to(dt)
    x[1] *= 1e20 if dt == torch.float64 else 1e15
    x[2] *= 1e-20 if dt == torch.float64 else 1e-15
    x[3] = 0
    acc = torch.full((5,), 3, dtype=torch.int32).cuda()
    y, e = h.pow2_normalize(x.cuda(), expo_acc=acc)
    y, e = y.cpu(), e.cpu()
    nrm = x.double().reshape(5, -1).norm(dim=1)
    want_e = torch.frexp(nrm)[1].to(torch.int32)
    want_e[3] = 0
    assert torch.equal(e, want_e) and torch.equal(acc.cpu(), want_e + 3)
    assert torch.equal(y, torch.ldexp(x, -e[:, None, None].to(torch.int32)))          # exact power-of-two scaling
    yn = y.double().reshape(5, -1).norm(dim=1)
    assert ((yn >= 0.5 - 1e-6) & (yn < 1.0 + 1e-6))[[0, 1, 2, 4]].all() and yn[3] == 0
    _, e2 = h.pow2_normalize(x.cuda(), exponent_only=True)
    assert torch.equal(e2.cpu(), want_e)
    back = h.scale_batch(y.cuda(), expo=e.cuda(), expo_sign=+1).cpu()
    assert torch.equal(back, x)
    s = torch.tensor([2.0, -1.0, 0.5, 3.0, 1.0], dtype=dt)
    z = h.scale_batch(x.cuda(), scale=s.cuda()).cpu()
    assert torch.allclose(z, x * s[:, None, None], rtol=1e-6 if dt == torch.float32 else 1e-14, atol=0)
    z = h.scale_batch(x.cuda(), scale=-2.5).cpu()
    assert torch.allclose(z, x * -2.5, rtol=1e-6 if dt == torch.float32 else 1e-14, atol=0)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("solver", [1, 2, 3])
def test_eigh_split_k_partials(dt, solver):
    """G given as split-K partial sums (what a fused Gram kernel leaves behind): summed on load."""
    h = _hip()
    g = torch.Generator().manual_seed(17)
    B, parts, n = 3, 5, 48
    Mx = torch.randn(B, parts, n, 40, generator=g, dtype=torch.float64)
    Gp = (Mx @ Mx.transpose(2, 3)).to(dt)
    G = Gp.double().sum(dim=1)
    V, sig, info = h.eigh_trunc(Gp.cuda(), h.EIG_RAW, False, 0.0, n, abs_floor=solver)
    V, sig = V.cpu().double(), sig.cpu().double()
    wref = torch.linalg.eigvalsh(G).flip(-1)
    assert ((sig**2 - wref).abs().max(dim=1).values / wref[:, 0]).max() < tol(dt, 2e-6, 1e-13)
    assert (G @ V - V * (sig**2)[:, None, :]).abs().max() / wref.max() < tol(dt, 2e-5, 1e-12)


@pytest.mark.parametrize("dt", DT)
def test_eigh_live_graded(dt):
    """TTR_SOLVER_JACOBI_LIVE on an accurately formed graded Gram matrix (pass 2 of 'svd'): every eigenvalue above the
    dead threshold (n eps)^2 lambda_max comes out with RELATIVE accuracy; the absolute-floor solver only resolves
    them to eps * lambda_max."""
    h = _hip()
    g = torch.Generator().manual_seed(23)
    n = 64
    eps = torch.finfo(dt).eps
    decay = 0.5 if dt == torch.float32 else 1.2          # sigma_63 / sigma_0 = 3e-10 (fp32) / 2e-23 (fp64)
    s = 2.0 ** (-decay * torch.arange(n, dtype=torch.float64))
    Q, _ = torch.linalg.qr(torch.randn(3, 512, n, generator=g, dtype=torch.float64))
    W = torch.linalg.qr(torch.randn(3, n, n, generator=g, dtype=torch.float64))[0]
    W = torch.eye(n, dtype=torch.float64) + 1e-3 * (W - torch.eye(n, dtype=torch.float64))   # nearly diagonal after "pass 1"
    rows = (W * s[None, None, :]).transpose(1, 2) @ Q.transpose(1, 2)                        # graded rows, mildly coupled
    G = (rows @ rows.transpose(1, 2))
    Gd = G.to(dt)
    V, sig, info = h.eigh_trunc(Gd.cuda(), h.EIG_RAW, False, 0.0, n, abs_floor=h.SOLVER_JACOBI_LIVE)
    V, sig = V.cpu().double(), sig.cpu().double()
    wref = torch.linalg.eigvalsh(Gd.double()).flip(-1).clamp_min(0)
    sref = wref.sqrt()
    live = sref[0] > 4 * n * eps * sref[0, 0]
    rel = ((sig[0] - sref[0]).abs() / sref[0])[live]
    assert rel.max() < tol(dt, 2e-4, 1e-10), rel.max()
    assert (V.transpose(1, 2) @ V - torch.eye(n, dtype=torch.float64)).abs().max() < tol(dt, 3e-5, 1e-12)
    # rotated rows of the LIVE directions are orthogonal relative to their own norms (what the projection needs)
    Rw = V[0].T @ rows[0]
    Rn = Rw[live] / Rw[live].norm(dim=1, keepdim=True)
    assert (Rn @ Rn.T - torch.eye(int(live.sum()), dtype=torch.float64)).abs().max() < tol(dt, 5e-4, 1e-9)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("parts", [1, 3])
def test_eigh_live_prefix(dt, parts):
    """TTR_SOLVER_JACOBI_LIVE, n = 64, restricts the solve to the leading block that holds every live index (G_ii above
    (n eps)^2 max G_ii): items whose frozen tail is exactly zero (a packed bond: blockdiag(G11, 0)), items with a noise-level
    tail coupled to the live block (frozen indices keep e_i and their own diagonal entry, sorted into place), an item with a live
    index at the very end (nothing to cut) and a pass-through item."""
    h = _hip()
    g = torch.Generator().manual_seed(29)
    B, n, m = 6, 64, 300
    eps = torch.finfo(dt).eps
    s = 2.0 ** (-0.4 * torch.arange(32, dtype=torch.float64))
    rows = torch.zeros(B, n, m, dtype=torch.float64)
    W = torch.linalg.qr(torch.randn(B, 32, 32, generator=g, dtype=torch.float64))[0]
    W = torch.eye(32, dtype=torch.float64) + 1e-2 * (W - torch.eye(32, dtype=torch.float64))
    Q = torch.linalg.qr(torch.randn(B, m, 32, generator=g, dtype=torch.float64))[0]
    rows[:, :32] = (W * s[None, None, :]).transpose(1, 2) @ Q.transpose(1, 2)
    noise = (0.01 * n * eps) * torch.randn(B, 32, m, generator=g, dtype=torch.float64) / m ** 0.5   # far below the frozen threshold
    rows[2, 32:] = noise[2]
    rows[3, 32:] = noise[3]
    rows[4, 32:] = noise[4]
    rows[4, 63] = 1e-2 * torch.randn(m, generator=g, dtype=torch.float64) / m ** 0.5                # item 4: live index 63
    G = rows @ rows.transpose(1, 2)
    Gp = (torch.stack([G * w for w in (0.5, 0.25, 0.25)], dim=1) if parts > 1 else G).to(dt).cuda()
    skip = torch.tensor([0, 0, 0, 0, 0, 1], dtype=torch.int32, device="cuda")
    sin = torch.rand(B, n, generator=g, dtype=torch.float64).sort(dim=1, descending=True).values.to(dt).cuda()
    delta2 = float((noise[2] ** 2).sum()) * 4
    V, sg, info = h.eigh_trunc(Gp, h.EIG_RAW, True, delta2, n, abs_floor=h.SOLVER_JACOBI_LIVE, skip_items=skip, sigma_in=sin)
    V, sg, info = V.cpu().double(), sg.cpu().double(), info.cpu()
    assert torch.equal(sg[5], sin[5].cpu().double()) and torch.equal(V[5], torch.eye(n, dtype=torch.float64))
    Gd = (Gp.double().sum(dim=1) if parts > 1 else Gp.double()).cpu()
    sref = torch.linalg.eigvalsh(Gd).flip(-1).clamp_min(0).sqrt()
    for b in range(5):
        assert (sg[b, :-1] >= sg[b, 1:]).all()                                           # sorted over the whole spectrum
        assert (V[b].T @ V[b] - torch.eye(n, dtype=torch.float64)).abs().max() < tol(dt, 3e-5, 1e-12)
        live = sref[b] > 4 * n * eps * sref[b, 0]
        assert (((sg[b] - sref[b]).abs() / sref[b])[live]).max() < tol(dt, 2e-4, 1e-10)
        assert (sg[b] - sref[b]).abs().max() < 4 * n * eps * sref[b, 0]
        Rw = V[b].T @ rows[b]
        Rn = Rw[live] / Rw[live].norm(dim=1, keepdim=True)
        assert (Rn @ Rn.T - torch.eye(int(live.sum()), dtype=torch.float64)).abs().max() < tol(dt, 5e-4, 1e-9)
        # the rank rule saw the whole spectrum: the longest tail with sum(sigma^2) <= delta2
        tails = torch.cumsum((sg[b] ** 2).flip(0), 0).flip(0)
        assert int(info[b]) == max(1, int((tails.to(dt) > torch.tensor(delta2, dtype=dt)).sum()))
    for b in (0, 1):   # exactly zero tail: unit vectors, zero sigma, no coupling
        assert (sg[b, 32:] == 0).all() and torch.equal(V[b, 32:, 32:], torch.eye(32, dtype=torch.float64))
        assert (V[b, :32, 32:] == 0).all() and (V[b, 32:, :32] == 0).all()


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("columns", [False, True])
def test_orth_fixup(dt, columns):
    h = _hip()
    g = torch.Generator().manual_seed(29)
    B, r, n = 4, 12, 300
    Q = torch.linalg.qr(torch.randn(B, n, r, generator=g, dtype=torch.float64))[0].transpose(1, 2).contiguous()  # [B, r, n]
    sig = torch.ones(B, r, dtype=torch.float64) * torch.linspace(1, 0.5, r, dtype=torch.float64)
    X = Q.clone()
    # item 0: untouched.  item 1: last 3 vectors dead and noisy.  item 2: a dead vector inside the span, a zero one, a NaN one
    sig[1, 9:] = 1e-12
    X[1, 9:] = X[1, 9:] * 3.0 + 0.3 * X[1, :3] + 0.2 * torch.randn(3, n, generator=g, dtype=torch.float64)
    sig[2, 8:] = 0.0
    X[2, 8] = X[2, 0] - 2 * X[2, 3]
    X[2, 9] = 0
    X[2, 10] = float("nan")
    X[2, 11] = 5 * X[2, 11]
    sig[3, 11] = 1e-9
    Xd = (X.transpose(1, 2).contiguous() if columns else X).to(dt).cuda()
    before = Xd.clone()
    h.orth_fixup(Xd, sig.to(dt).cuda(), r, 1e-6, columns=columns)
    out = Xd.cpu().double()
    out = out.transpose(1, 2) if columns else out
    eye = torch.eye(r, dtype=torch.float64)
    t = tol(dt, 2e-6, 1e-13)
    for b in range(B):
        assert (out[b] @ out[b].T - eye).abs().max() < t, b
    assert torch.equal(Xd[0], before[0])                                   # nothing dead: untouched, bit for bit
    live = out[1, :9] if True else None
    assert (live - X[1, :9]).abs().max() < t                               # live vectors are never modified


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("n", [1024, 1500])
def test_orth_fixup_split_rounds_special_vectors(dt, n):
    """The three-launch rounds of ttr_orth_fixup (large batches; forced here with TTR_KNOB_ORTH_SPLIT = 1): an untouched item, noisy
    dead vectors, a dead vector inside the span of the others, a zero one, one with NaN entries, a scaled one -- orthonormal
    results, live vectors bit-identical, the same vectors as the single-launch kernel where the remainders are genuine."""
    h = _hip()
    g = torch.Generator().manual_seed(31 + n)
    B, r = 5, 12
    Q = torch.linalg.qr(torch.randn(B, n, r, generator=g, dtype=torch.float64))[0].transpose(1, 2).contiguous()
    sig = torch.ones(B, r, dtype=torch.float64) * torch.linspace(1, 0.5, r, dtype=torch.float64)
    X = Q.clone()
    sig[1, 9:] = 1e-12
    X[1, 9:] = X[1, 9:] * 3.0 + 0.3 * X[1, :3] + 0.2 * torch.randn(3, n, generator=g, dtype=torch.float64)
    sig[2, 8:] = 0.0
    X[2, 8] = X[2, 0] - 2 * X[2, 3]
    X[2, 9] = 0
    X[2, 10] = float("nan")
    X[2, 11] = 5 * X[2, 11]
    sig[3, 11] = 1e-9
    sig[4, 4:] = 1e-10                                   # most vectors dead, mostly leakage of the live ones: needs the second round
    X[4, 4:] = 0.1 * X[4, 4:] + X[4, :4].sum(dim=0, keepdim=True)
    Xd = X.to(dt).cuda()
    before = Xd.clone()
    sg = sig.to(dt).cuda()
    assert h.lib().ttr_orth_fixup_workspace_bytes(h.dtype_code(dt), r, n, B, 1) == 0            # default: always the single launch
    mono = before.clone()
    h.orth_fixup(mono, sg, r, 1e-6)
    h.set_knob(h.KNOB_ORTH_SPLIT, 1)
    assert h.lib().ttr_orth_fixup_workspace_bytes(h.dtype_code(dt), r, n, B, 1) > 0
    h.orth_fixup(Xd, sg, r, 1e-6)
    out = Xd.cpu().double()
    eye = torch.eye(r, dtype=torch.float64)
    t = tol(dt, 2e-6, 1e-13)
    for b in range(B):
        assert torch.isfinite(out[b]).all() and (out[b] @ out[b].T - eye).abs().max() < t, b
    assert torch.equal(Xd[0], before[0]) and torch.equal(Xd[1, :9], before[1, :9]) and torch.equal(Xd[4, :4], before[4, :4])
    for b in (1, 3, 4):                                  # genuine remainders: the same vectors as the single-launch kernel
        assert (Xd[b] - mono[b]).abs().max().item() < tol(dt, 2e-5, 1e-11), b


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("r,n,first", [(32, 2048, 17), (64, 4096, 1), (30, 777, 29), (7, 50, 0), (70, 500, 40), (32, 1000, 5), (20, 300, 0),
                                       (16, 4096, 9)])
def test_orth_fixup_many_dead(dt, r, n, first, split):
    """Most of the kept vectors below the resolution (a rank-64 bond with sigma_j ~ 2^-j: SURVEY 8d's decaying variant of the
    metric): the block kernel (r <= 64; Gram matrix + Gram-Schmidt in coefficient space) and the sequential one (r = 70) leave
    every vector orthonormal, the live ones bit-identical, and the dead ones inside span(their own old value, earlier vectors)
    where that remainder was genuine."""
    h = _hip()
    if split:   # the three-launch rounds (large batches), forced on this small one; shapes it does not cover keep the single launch
        if r > 64 or n < 512:
            pytest.skip("outside the split path's envelope")
        h.set_knob(h.KNOB_ORTH_SPLIT, 1)
    g = torch.Generator().manual_seed(r * 1000 + n)
    B = 3
    Q = torch.linalg.qr(torch.randn(B, n, r, generator=g, dtype=torch.float64))[0].transpose(1, 2).contiguous()  # [B, r, n]
    sig = torch.linspace(1.0, 0.25, r, dtype=torch.float64).repeat(B, 1)
    sig[:, first:] = 1e-12 * torch.rand(B, r - first, generator=g, dtype=torch.float64) if first > 0 else 0.0   # (sigma_0 = 0: all dead)
    X = Q.clone()
    noise = torch.randn(B, r - first, n, generator=g, dtype=torch.float64)
    X[:, first:] = 0.5 * X[:, first:] + 0.1 * noise + (0.3 * X[:, :1] if first > 0 else 0.0)   # noisy, not orthogonal, not unit
    X[1, r - 1] = 0.0                                                                            # one exactly-zero dead vector
    Xd = X.to(dt).cuda()
    before = Xd.clone()
    h.orth_fixup(Xd, sig.to(dt).cuda(), r, 1e-6)
    out = Xd.cpu().double()
    eye = torch.eye(r, dtype=torch.float64)
    for b in range(B):
        assert (out[b] @ out[b].T - eye).abs().max() < tol(dt, 2e-6, 1e-13), b
    assert torch.equal(Xd[:, :first], before[:, :first])
    # genuine remainders are kept: dead vector `first` of item 0 is its old value minus the projection on the earlier vectors
    old = X[0, first].to(dt).double()
    prev = out[0, :first]
    rem = old - prev.T @ (prev @ old)
    rem = rem / rem.norm()
    assert (out[0, first] - rem).abs().max() < tol(dt, 5e-6, 1e-12)
    if r <= 32 and dt == torch.float32 and not split:
        # round 5's inner loops (TTR_KNOB_ORTH_V2, the default) against round 4's on the same input: another summation order of the
        # same sums -- the same vectors to rounding
        h.set_knob(h.KNOB_ORTH_V2, 0)
        try:
            X1 = before.clone()
            h.orth_fixup(X1, sig.to(dt).cuda(), r, 1e-6)
        finally:
            h.set_knob(h.KNOB_ORTH_V2, 2)
        assert torch.equal(X1[:, :first], Xd[:, :first]) and (X1 - Xd).abs().max().item() < 2e-5
    if split and r <= 32:
        # the second round's Gram matrix left by the first round's apply launch (TTR_KNOB_ORTH_V2 = 2, the default) against a
        # Gram launch of its own (1): the same sums in another order -- the same vectors to rounding
        h.set_knob(h.KNOB_ORTH_V2, 1)
        try:
            X1 = before.clone()
            h.orth_fixup(X1, sig.to(dt).cuda(), r, 1e-6)
        finally:
            h.set_knob(h.KNOB_ORTH_V2, 2)
        o1 = X1.cpu().double()
        for b in range(B):
            assert (o1[b] @ o1[b].T - eye).abs().max() < tol(dt, 2e-6, 1e-13), b
        assert torch.equal(X1[:, :first], Xd[:, :first]) and (X1 - Xd).abs().max().item() < tol(dt, 2e-5, 1e-11)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("rows,n,B", [(4096, 64, 2), (100000, 32, 1), (1000, 37, 3), (33, 5, 2), (7, 1, 1), (300000, 64, 1), (64, 64, 40)])
def test_colsweep_gram_project(dt, rows, n, B):
    """ttr_colgram / ttr_colproject (tall matrices, contraction over the rows) against float64 torch."""
    h = _hip()
    g = torch.Generator().manual_seed(rows + n)
    M = torch.randn(B, rows, n, generator=g, dtype=torch.float64).to(dt)
    V1 = torch.linalg.qr(torch.randn(B, n, n, generator=g, dtype=torch.float64))[0].to(dt)
    V2 = torch.linalg.qr(torch.randn(B, n, n, generator=g, dtype=torch.float64))[0].to(dt)
    sig = (torch.rand(B, n, generator=g, dtype=torch.float64) + 0.5).to(dt)
    Md, V1d, V2d = M.double(), V1.double(), V2.double()
    t = tol(dt, 2e-5, 1e-12)
    G = h.colgram(M.cuda()).cpu().double()
    Gr = Md.transpose(1, 2) @ Md
    assert (G - Gr).abs().max() / Gr.abs().max() < t
    G2 = h.colgram(M.cuda(), V1.cuda()).cpu().double()
    Mw = Md @ V1d
    G2r = Mw.transpose(1, 2) @ Mw
    assert (G2 - G2r).abs().max() / G2r.abs().max() < t
    ro = max(1, n // 2)
    U = V1d @ V2d[:, :, :ro]
    left, right = h.colproject(M.cuda(), V1.cuda(), V2.cuda(), sig.cuda(), ro, left_ortho=True)
    assert (left.cpu().double() - (Md @ U) / sig.double()[:, None, :ro]).abs().max() / Md.abs().max() < 4 * t
    assert (right.cpu().double() - U.transpose(1, 2) * sig.double()[:, :ro, None]).abs().max() < t
    left, right = h.colproject(M.cuda(), None, V2.cuda(), sig.cuda(), ro, left_ortho=False)
    assert (left.cpu().double() - Md @ V2d[:, :, :ro]).abs().max() / Md.abs().max() < 4 * t
    assert (right.cpu().double() - V2d[:, :, :ro].transpose(1, 2)).abs().max() < t


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("k,ra,rb,I,ca,cb", [(64, 32, 32, 64, 32, 32), (20, 7, 9, 11, 5, 12), (64, 40, 24, 16, 16, 48), (33, 10, 3, 70, 30, 20)])
def test_qr_pushed_sum(dt, k, ra, rb, I, ca, cb, qr_variant):
    """Fused push of the block-diagonal middle core of a TT sum: same factorisation as pushing the padded core."""
    h = _hip()
    g = torch.Generator().manual_seed(k + ra * 3 + I)
    B = 2
    Rm = torch.randn(B, k, ra + rb, generator=g, dtype=torch.float64).to(dt)
    a = torch.randn(B, ra, I, ca, generator=g, dtype=torch.float64).to(dt)
    b = torch.randn(B, rb, I, cb, generator=g, dtype=torch.float64).to(dt)
    za, zb = torch.zeros(B, ra, I, cb, dtype=dt), torch.zeros(B, rb, I, ca, dtype=dt)
    core = torch.cat([torch.cat([a, za], dim=-1), torch.cat([zb, b], dim=-1)], dim=1)
    n = ca + cb
    P = (Rm.double() @ core.double().reshape(B, ra + rb, I * n)).reshape(B, k * I, n)
    f = h.qr_factor_pushed_sum(Rm.cuda(), a.cuda(), b.cuda())
    Q = h.qr_apply(f).cpu().double()
    R = f.R.cpu().double()
    kq = min(k * I, n)
    assert Q.shape == (B, k * I, kq) and R.shape == (B, kq, n)
    assert (Q.transpose(1, 2) @ Q - torch.eye(kq, dtype=torch.float64)).abs().max() < tol(dt, 3e-5, 1e-12)
    assert (Q @ R - P).abs().max() / P.abs().max() < tol(dt, 2e-5, 1e-12)
    f2 = h.qr_factor_pushed(Rm.cuda(), core.cuda())
    assert (f2.R.cpu().double().abs() - R.abs()).abs().max() / R.abs().max() < tol(dt, 3e-4, 1e-10)


@pytest.mark.gpu
def test_batch_above_grid_limit():
    """More than 65535 batch items (the batch is a grid dimension): the entry points slice the batch themselves."""
    torch.manual_seed(3)
    B = 70001
    _hip = globals()["_hip"]()
    A = torch.randn(B, 5, 6, device="cuda")
    Bm = torch.randn(B, 6, 3, device="cuda")
    C = _hip.gemm(A, Bm)
    assert (C - torch.bmm(A.double(), Bm.double()).float()).abs().max().item() < 1e-5
    X = torch.randn(B, 12, 5, device="cuda")
    Q, R = _hip.qr(X)
    assert (torch.bmm(Q, R) - X).abs().max().item() < 1e-5
    eye = torch.eye(5, device="cuda").expand(B, 5, 5)
    assert (torch.bmm(Q.transpose(1, 2), Q) - eye).abs().max().item() < 1e-5
    sc = torch.rand(B, 5, device="cuda") + 0.5
    Y = _hip.scale_cols(X, sc, _hip.SCALE_MUL)
    assert (Y - X * sc[:, None, :]).abs().max().item() < 1e-6


@pytest.mark.parametrize("Rin,I,n", [(64, 64, 64), (64, 8, 64), (40, 24, 32), (64, 16, 48)])
def test_qr_apply_pushed_gram(Rin, I, n):
    """ttr_qr_apply_pushed_gram: the apply kernel's own row Gram matrix of its output (round.py:104-109 fused into the
    push-left of tensor.py:2081-2083) -- same Out bit for bit, partials sum to M M^T of the k x (I kcols) unfolding."""
    h = _hip()
    g = torch.Generator().manual_seed(Rin + 10 * I + n)
    B, k, kc = 3, 64, 32
    Rm = torch.randn(B, k, Rin, generator=g).cuda()
    core = torch.randn(B, Rin, I, n, generator=g).cuda()
    C = torch.randn(B, n, kc, generator=g).cuda()
    f = h.qr_factor_pushed(Rm, core)
    plain = h.qr_apply(f, C)
    # row packing (default knob) and the fused Gram epilogue exclude each other: refused, not a Gram matrix with unwritten partials
    with pytest.raises(NotImplementedError):
        h.qr_apply(f, C, want_gram=True)
    h.set_knob(h.KNOB_QR_PACK, 0)
    try:
        f = h.qr_factor_pushed(Rm, core)
        assert torch.equal(h.qr_apply(f, C), plain)   # (full-rank Rm: nothing packed either way)
        out, G = h.qr_apply(f, C, want_gram=True)
        out5, G5 = h.qr_apply(f, C[:, :, :5].contiguous(), want_gram=True)
        # rank-32 Rm (what packs with the knob on): unpacked map, every partial written
        Rlow = (torch.randn(B, k, 32, generator=g) @ torch.randn(B, 32, Rin, generator=g)).cuda()
        flow = h.qr_factor_pushed(Rlow, core)
        outl, Gl = h.qr_apply(flow, C, want_gram=True)
    finally:
        h.set_knob(h.KNOB_QR_PACK, 3)
    Ml = outl.double().reshape(B, k, I * kc)
    refl = Ml @ Ml.transpose(1, 2)
    assert Gl is not None and (Gl.double().sum(dim=1) - refl).abs().max() / refl.abs().max() < 2e-6
    assert G is not None and G.shape == (B, I // 8, k, k)
    assert torch.equal(out, plain)
    M = out.double().reshape(B, k, I * kc)
    ref = M @ M.transpose(1, 2)
    got = G.double().sum(dim=1)
    assert (got - ref).abs().max() / ref.abs().max() < 2e-6
    assert torch.equal(G, G.transpose(2, 3))  # both triangles written from the same accumulators
    # the eigensolver takes the partials as they are (gparts)
    V, sig, _ = h.eigh_trunc(G, h.EIG_RAW, False, 0.0, k, abs_floor=h.SOLVER_TRIDIAG)
    sv = torch.linalg.svdvals(M)
    assert (sig.double() ** 2 - sv ** 2).abs().max() / sv.max() ** 2 < 1e-5  # eigenvalues of a Gram matrix: absolute accuracy
    # shapes the fused epilogue does not cover fall back to (Out, None)
    assert G5 is None and out5.shape == (B, k * I, 5)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("n,k,B,graded", [(256, 8, 3, False), (128, 16, 2, False), (200, 5, 2, True), (512, 32, 1, False),
                                          (65, 64, 1, False), (256, 1, 2, True), (256, 40, 2, True), (1024, 16, 1, False)])
def test_eigh_topk(dt, n, k, B, graded):
    """Selected eigenpairs above one workgroup (ttr_tridiag -> ttr_tri_eigsel -> ttr_qr -> ttr_tridiag_back): the k largest
    eigenvalues vs LAPACK, orthonormality and residual of the vectors; flat (Marchenko-Pastur) and graded spectra."""
    h = _hip()
    g = torch.Generator().manual_seed(7 * n + k)
    Mx = torch.randn(B, n, 3 * n + 1, generator=g, dtype=torch.float64)
    if graded:
        Mx = Mx * (0.9 ** torch.arange(n, dtype=torch.float64))[None, :, None]
    G = (Mx @ Mx.transpose(1, 2)).to(dt)
    X, lam, rmin = h.eigh_topk(G.cuda(), k)
    X, lam, rmin = X.cpu().double(), lam.cpu().double(), rmin.cpu().double()
    wref = torch.linalg.eigvalsh(G.double()).flip(-1)
    assert X.shape == (B, n, k) and lam.shape == (B, k)
    assert ((lam - wref[:, :k]).abs().max(dim=1).values / wref[:, 0]).max() < tol(dt, 3e-6, 1e-13)
    assert (X.transpose(1, 2) @ X - torch.eye(k, dtype=torch.float64)).abs().max() < tol(dt, 5e-6, 1e-12)
    resid = (G.double() @ X - X * lam[:, None, :]).abs().max() / wref.max()
    assert resid < tol(dt, 2e-5, 1e-11)
    assert rmin.min() > 0.5


def _top_errors(G, V, sig, cols):
    """(orthogonality, residual / ||G||, sigma error / sigma_1) of the first `cols` eigenpairs, in fp64 on the host."""
    Gd, Vd, s = G.cpu().double(), V.cpu().double()[:, :, :cols], sig.cpu().double()[:, :cols]
    lam = torch.linalg.eigvalsh(Gd).flip(-1)[:, :cols]
    orth = (Vd.transpose(1, 2) @ Vd - torch.eye(cols, dtype=torch.float64)).abs().max().item()
    res = ((Gd @ Vd - Vd * (s * s)[:, None, :]).norm(dim=(1, 2)) / Gd.norm(dim=(1, 2))).max().item()
    serr = ((s - lam.clamp_min(0).sqrt()).abs().amax(dim=1) / s[:, 0]).max().item()
    return orth, res, serr


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("n,r", [(64, 32), (64, 16), (64, 3), (64, 1), (48, 24), (40, 32), (57, 20)])
def test_eigh_top(dt, n, r):
    """Pass 1 of a batch-mode bond (ttr_eigh_top): on flat spectra without close pairs the r largest eigenpairs come from
    multisection + twisted factorisations + Newton-Schulz (flag 1: zeros beyond r, info = r) and match LAPACK; items the path
    declines carry the full QL decomposition (flag 0)."""
    h = _hip()
    assert h.eigh_top_ok(n, r) and not h.eigh_top_ok(n, 33) and not h.eigh_top_ok(39, 8) and not h.eigh_top_ok(65, 8) and not h.eigh_top_ok(n, n)
    B = 24
    g = torch.Generator().manual_seed(11 * n + r)
    Mx = torch.randn(B, n, 3 * n + 1, generator=g, dtype=torch.float64)
    G = (Mx @ Mx.transpose(1, 2)).to(dt).cuda()
    V, sig, info, flat = h.eigh_top(G, r, 0.125)
    assert V.shape == (B, n, n) and sig.shape == (B, n) and flat.dtype == torch.int32
    it, iq = (flat == 1).nonzero()[:, 0], (flat != 1).nonzero()[:, 0]   # (2: declined, but flat by its sigma)
    assert len(it) >= B - 2                                  # (a pair closer than 512 eps lambda_1 is possible, not likely)
    orth, res, serr = _top_errors(G[it], V[it], sig[it], r)
    assert orth < tol(dt, 3e-6, 1e-14) and res < tol(dt, 3e-6, 1e-14) and serr < tol(dt, 3e-6, 1e-14)
    assert float(V[it][:, :, r:].abs().max()) == 0.0 and float(sig[it][:, r:].abs().max()) == 0.0
    assert info[it].unique().tolist() == [r]
    if len(iq):
        orth, res, serr = _top_errors(G[iq], V[iq], sig[iq], n)
        assert orth < tol(dt, 2e-5, 1e-13) and res < tol(dt, 2e-5, 1e-13)


@pytest.mark.parametrize("dt", DT)
def test_carry_rows32_flags(dt):
    """ttr_carry_rows32: the packing test of the fused push on a stand-alone carry (the sweep's last core) -- rows 32.. hold at
    most (8 eps)^2 of the squared norm; knob QR_PACK = 0 switches it off."""
    h = _hip()
    g = torch.Generator().manual_seed(3)
    eps = torch.finfo(dt).eps
    R = torch.randn(5, 64, 64, generator=g, dtype=torch.float64)
    R[0, 32:] = 0
    R[1, 32:] *= 2 * eps
    R[2, 32:] *= 64 * eps
    R[3, 40] *= 1e-3
    R[3, 32:40] = 0
    R[3, 41:] = 0
    view = torch.zeros(5, 64, 80, dtype=dt)
    view[:, :, :64] = R.to(dt)
    Rd = view.cuda()[:, :, :64]                      # (leading dimension 80)
    assert h.carry_rows32(Rd).tolist() == [1, 1, 0, 0, 0]
    h.set_knob(h.KNOB_QR_PACK, 0)
    try:
        assert h.carry_rows32(Rd).tolist() == [0] * 5
    finally:
        h.set_knob(h.KNOB_QR_PACK, 3)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("r", [32, 20])
@pytest.mark.parametrize("parts", [1, 5])
def test_eigh_top_zero_tail_is_solved_as_the_leading_block(dt, r, parts):
    """ttr_eigh_top on 64 x 64 Gram matrices whose rows / columns 32.. are exactly zero (the carry of a packed bond): solved
    as the leading 32 x 32 block.  Flat items: the r largest eigenpairs, zeros elsewhere; declined items (graded block): the
    full decomposition blockdiag(V11, I) with 32 zero eigenvalues; rank / sigma / residuals as for the full-size solve.  An item
    with one nonzero entry on the lower diagonal is not shrunk."""
    h = _hip()
    B, n = 12, 64
    g = torch.Generator().manual_seed(100 + r + parts)
    rows = torch.zeros(B, n, 200, dtype=torch.float64)
    rows[:, :32] = torch.randn(B, 32, 200, generator=g, dtype=torch.float64)
    rows[8:10, :32] *= (2.0 ** (-0.5 * torch.arange(32, dtype=torch.float64)))[None, :, None]     # graded: declined
    rows[11, 40] = 0.5 * torch.randn(200, generator=g, dtype=torch.float64)                       # item 11: a live row below 32
    G = rows @ rows.transpose(1, 2)
    Gp = G.to(dt)
    if parts > 1:
        w = torch.tensor([0.5, 0.125, 0.125, 0.125, 0.125], dtype=dt)
        Gp = Gp[:, None] * w[None, :, None, None]
    Gd = (Gp.double().sum(dim=1) if parts > 1 else Gp.double())
    V, sig, info, flat = h.eigh_top(Gp.cuda(), r, 0.125)
    flat = flat.cpu()
    assert flat[:8].tolist() == [1] * 8 and flat[8:10].tolist() == [0, 0]
    it = (flat == 1).nonzero()[:, 0]
    orth, res, serr = _top_errors(Gd[it].to(dt).cuda(), V[it.cuda()], sig[it.cuda()], r)
    assert orth < tol(dt, 3e-6, 1e-14) and res < tol(dt, 3e-6, 1e-14) and serr < tol(dt, 3e-6, 1e-14)
    assert float(V[it.cuda()][:, :, r:].abs().max()) == 0.0 and float(sig[it.cuda()][:, r:].abs().max()) == 0.0
    assert info.cpu()[it].unique().tolist() == [r]
    Vc, sc = V.cpu().double(), sig.cpu().double()
    for b in range(10):
        assert float(Vc[b, 32:, :32].abs().max()) == 0.0                                       # nothing leaks into the zero rows
    for b in (8, 9):                                                                              # declined: the QL decomposition
        assert torch.equal(Vc[b, 32:, 32:], torch.eye(32, dtype=torch.float64)) and float(sc[b, 32:].abs().max()) == 0.0
        assert float(Vc[b, :32, 32:].abs().max()) == 0.0 and int(info[b]) == n      # (declined items: the full-size rank, as ever)
        wref = torch.linalg.eigvalsh(Gd[b]).flip(-1).clamp_min(0)
        assert ((sc[b] ** 2 - wref).abs().max() / wref[0]) < tol(dt, 2e-6, 1e-13)
        assert (Gd[b] @ Vc[b] - Vc[b] * (sc[b] ** 2)[None, :]).abs().max() / wref[0] < tol(dt, 2e-5, 1e-12)
        assert (Vc[b].T @ Vc[b] - torch.eye(n, dtype=torch.float64)).abs().max() < tol(dt, 2e-5, 1e-13)
    # item 11 is a full-size problem: row 40 takes part
    b = 11
    k = r if int(flat[b]) == 1 else n
    wref = torch.linalg.eigvalsh(Gd[b]).flip(-1).clamp_min(0)
    assert ((sc[b, :k] ** 2 - wref[:k]).abs().max() / wref[0]) < tol(dt, 3e-6, 1e-13)
    assert float(Vc[b, 40, :k].abs().max()) > 1e-3


@pytest.mark.parametrize("r", [32, 20])
def test_eigh_top_large_launches_use_a_build_at_higher_occupancy_with_the_same_bits(r):
    """fp32 ttr_eigh_top launches of >= 1024 matrices run the 32-row instance from a build capped at 168 VGPRs (three waves per SIMD,
    spilled registers; TTR_KNOB_EIGH_SMALL = 2, the default; 3: 128 VGPRs): flat, declined (graded) and full-size items come out
    BIT-identical to the spill-free build (1) -- eigenvectors, singular values, ranks and flags."""
    h = _hip()
    B, n = 1100, 64
    g = torch.Generator().manual_seed(500 + r)
    rows = torch.zeros(B, n, 96, dtype=torch.float64)
    rows[:, :32] = torch.randn(B, 32, 96, generator=g, dtype=torch.float64)
    rows[::7, :32] *= (2.0 ** (-0.5 * torch.arange(32, dtype=torch.float64)))[None, :, None]     # graded: declined
    rows[5::11, 40] = 0.5 * torch.randn(len(range(5, B, 11)), 96, generator=g, dtype=torch.float64)   # full-size items
    G = (rows @ rows.transpose(1, 2)).to(torch.float32).cuda()
    out = {}
    try:
        for v in (1, 2, 3):
            h.set_knob(h.KNOB_EIGH_SMALL, v)
            out[v] = [x.clone() for x in h.eigh_top(G, r, 0.125)]
    finally:
        h.set_knob(h.KNOB_EIGH_SMALL, 2)
    assert 0 < int((out[1][3] == 1).sum()) < B
    for v in (2, 3):
        assert all(torch.equal(a, b) for a, b in zip(out[1], out[v])), v


@pytest.mark.parametrize("dt", DT)
def test_eigh_top_declines_what_it_cannot_certify(dt):
    """Graded kept spectra, exactly and nearly multiple eigenvalues, identity, zero: flag 0 and the QL phase of the same launch
    delivers what ttr_eigh_trunc's tridiagonal solver delivers.  Pairs / triples just ABOVE the admitted distance (512 eps lambda_1)
    are taken by the top-r path and must come out orthonormal (two Newton-Schulz steps in fp32, three in fp64)."""
    h = _hip()
    n, r, B = 64, 32, 6
    e = torch.finfo(dt).eps
    g = torch.Generator().manual_seed(5)
    Q = torch.linalg.qr(torch.randn(B, n, n, generator=g, dtype=torch.float64))[0]

    def gram(spectrum):
        lam = torch.tensor(spectrum, dtype=torch.float64) ** 2
        return ((Q * lam) @ Q.transpose(1, 2)).to(dt).cuda()

    # (flag 2: declined by the top-r path, but the kept sigma of the full decomposition are flat -- the bond's pass-through flag)
    declined = {
        "graded": (0, gram([0.5 ** i for i in range(n)])),
        "double": (2, gram([1.0, 1.0] + [0.9 - 0.01 * i for i in range(n - 2)])),
        "pairs at 100 eps": (2, gram([(1.0 - 0.02 * (i // 2)) * (1.0 + 100 * e * (i % 2)) for i in range(n)])),
        "identity": (2, torch.eye(n, dtype=dt).repeat(B, 1, 1).cuda()),
    }
    for name, (flag, G) in declined.items():
        V, sig, info, flat = h.eigh_top(G, r, 0.125)
        assert flat.tolist() == [flag] * B, name
        Vq, sq, iq = h.eigh_trunc(G, h.EIG_RAW, False, 0.0, n, abs_floor=h.SOLVER_TRIDIAG)
        assert torch.equal(info, iq), name
        assert ((sig - sq).abs().max() / sq.max()).item() < tol(dt, 1e-6, 1e-14), name
        orth, res, _ = _top_errors(G, V, sig, n)
        assert orth < tol(dt, 2e-5, 1e-13) and res < tol(dt, 2e-5, 1e-13), name
    Z = torch.zeros(B, n, n, dtype=dt).cuda()
    V, sig, info, flat = h.eigh_top(Z, r, 0.125)
    assert flat.tolist() == [0] * B and info.tolist() == [0] * B and float(sig.abs().max()) == 0.0
    taken = {
        "pairs at 2000 eps": gram([(1.0 - 0.02 * (i // 2)) * (1.0 + 2000 * e * (i % 2)) for i in range(n)]),
        "triples at 600 eps": gram([(1.0 - 0.03 * (i // 3)) * (1.0 + 600 * e * (i % 3)) for i in range(n)]),
    }
    for name, G in taken.items():
        V, sig, info, flat = h.eigh_top(G, r, 0.125)
        assert flat.tolist() == [1] * B, name
        orth, res, serr = _top_errors(G, V, sig, r)
        assert orth < tol(dt, 3e-6, 1e-14) and res < tol(dt, 3e-6, 1e-14) and serr < tol(dt, 3e-6, 1e-14), name


@pytest.mark.parametrize("dt", DT)
def test_eigh_topk_multiple_eigenvalues_are_declined_or_spanned(dt):
    """A threefold largest eigenvalue: the twisted factorisation returns the same vector three times, the TSQR's min |R_jj| reports the
    collapse and the truncation's shortcut (`_topk_one_pass`) declines -- the block-Jacobi driver takes the bond.  A well separated
    spectrum next to it is accepted.  Whatever is accepted must span an invariant subspace."""
    from tntorch_amd import _hipops
    h = _hip()
    g = torch.Generator().manual_seed(3)
    n, k = 128, 4
    Q = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))[0]
    tail = torch.linspace(1.0, 0.1, n - 3, dtype=torch.float64)
    Gc = (Q * torch.cat([torch.tensor([5.0, 5.0, 5.0], dtype=torch.float64), tail])) @ Q.T       # lambda = 5, 5, 5, 1, ...
    Gs = (Q * torch.cat([torch.tensor([5.0, 4.0, 3.0], dtype=torch.float64), tail])) @ Q.T       # lambda = 5, 4, 3, 1, ...
    G = torch.stack([Gc, Gs]).to(dt).cuda()
    X, lam, rmin = h.eigh_topk(G, k)
    assert float(rmin[0]) < 0.5 and float(rmin[1]) > 0.5
    Xs = X[1].cpu().double()
    assert (Xs.T @ Xs - torch.eye(k, dtype=torch.float64)).abs().max() < tol(dt, 5e-6, 1e-12)
    assert (Gs @ Xs - Xs * lam[1].cpu().double()).abs().max() / 5.0 < tol(dt, 2e-5, 1e-11)
    assert _hipops._topk_one_pass(G, k) is None            # one collapsed item declines the batch ...
    res = _hipops._topk_one_pass(G[1:2], 3)                # ... the separated one alone is flat within a factor 8 at r = 3: accepted
    assert res is not None and res[2].tolist() == [3]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("mode", ["raw", "ref"])
def test_eigh_tridiag_zero_tail_is_solved_as_the_leading_block(dt, mode):
    """ttr_eigh_trunc, tridiagonal solver, on 64 x 64 Gram matrices with exactly zero rows / columns 32..: the leading block is
    solved, V = blockdiag(V11, I), sigma[32:] = 0, the rank rule sees all 64 values (cap and tail energies) -- same results as
    for a matrix that is not shrunk (one tiny nonzero entry on the lower diagonal keeps it at full size)."""
    h = _hip()
    g = torch.Generator().manual_seed(77)
    B, n = 4, 64
    rows = torch.zeros(B, n, 150, dtype=torch.float64)
    rows[:, :32] = torch.randn(B, 32, 150, generator=g, dtype=torch.float64) * (2.0 ** (-0.3 * torch.arange(32, dtype=torch.float64)))[None, :, None]
    G = (rows @ rows.transpose(1, 2)).to(dt)
    Gfull = G.clone()
    Gfull[:, 50, 50] = torch.finfo(dt).tiny * 4       # not shrunk: the diagonal tail is not exactly zero
    em = h.EIG_RAW if mode == "raw" else h.EIG_REF
    delta2 = float(G[0].double().diagonal().sum()) * 1e-6
    out = [h.eigh_trunc(x.cuda(), em, True, delta2, 48, abs_floor=h.SOLVER_TRIDIAG) for x in (G, Gfull)]
    (V, sg, info), (Vf, sf, inf_f) = out
    assert torch.equal(info, inf_f)
    Vc, sc = V.cpu().double(), sg.cpu().double()
    assert float(sc[:, 32:].abs().max()) == 0.0 and float((sc - sf.cpu().double()).abs().max() / sc.max()) < tol(dt, 2e-6, 1e-14)
    eye = torch.eye(32, dtype=torch.float64)
    for b in range(B):
        assert torch.equal(Vc[b, 32:, 32:], eye) and float(Vc[b, :32, 32:].abs().max()) == 0.0 and float(Vc[b, 32:, :32].abs().max()) == 0.0
        wref = torch.linalg.eigvalsh(G[b].double()).flip(-1).clamp_min(0)
        assert ((sc[b] ** 2 - wref).abs().max() / wref[0]) < tol(dt, 2e-6, 1e-13)
        assert (G[b].double() @ Vc[b] - Vc[b] * (sc[b] ** 2)[None, :]).abs().max() / wref[0] < tol(dt, 2e-5, 1e-12)
        assert (Vc[b].T @ Vc[b] - torch.eye(n, dtype=torch.float64)).abs().max() < tol(dt, 2e-5, 1e-13)


def test_executed_work_census_follows_the_kernels_decisions():
    """ttr_prof_enable(2) / ttr_prof_collect_work: the flops / bytes the launches EXECUTED, read off the kernels' own per-item
    decisions.  On the metric's shape a full-rank train executes (nearly) the algorithmic work; the rank-inflated t = g + g
    executes well under half of the QR flops (packed rows, rank-skipped panels) and loads half the rows in the Gram /
    projection kernels; the census itself does not change any result."""
    import bench
    import tntorch_amd as tn

    h = _hip()
    dev = torch.device("cuda", 0)
    B = 4
    gg = bench.make_input(B, dev, seed=5)
    # bond sigma ~ 2^(-j/4): numerical rank 64 (sigma_63 / sigma_0 = 2e-5: no panel is skipped, nothing packs) and a kept
    # spectrum that is NOT flat (sigma_31 / sigma_0 = 5e-3: the second Gram pass runs)
    full = bench.make_decaying_input(B, dev, seed=5, decay=0.25)
    model = bench.kernel_model()
    res = {}
    for name, inp in (("gg", gg), ("full", full)):
        t0 = tn.Tensor(inp, batch=True); t0.round_tt(rmax=32)
        h.prof_enable(2)
        t = tn.Tensor(inp, batch=True); t.round_tt(rmax=32)
        torch.cuda.synchronize()
        prof, work = h.prof_collect(), h.prof_collect_work()
        h.prof_enable(False)
        assert all(torch.equal(a, b) for a, b in zip(t.cores, t0.cores))
        res[name] = (prof, work)
        pk = bench.per_kind_roofline(prof, work, B, 1)
        for k in bench.CENSUS_KINDS:
            if k in pk and pk[k].get("frac") is not None:
                assert 0.0 < pk[k]["frac"] <= 1.0, (name, k, pk[k])          # executed work over measured time: below either roof
        assert bench.headline_roofline(pk, max(pk, key=lambda k: pk[k]["ms_per_step"]), prof, 1)["input_aware"] in (True, False)
    wf, wg = res["full"][1], res["gg"][1]
    for k in ("qr_factor", "qr_apply"):
        share_full = wf[k]["flops"] / (model[k]["flops"] * B)
        share_gg = wg[k]["flops"] / (model[k]["flops"] * B)
        assert 0.6 <= share_full <= 1.05, (k, share_full)      # (triangular R: 10 of 16 tiles of the push)
        assert 0.15 <= share_gg <= 0.6 * share_full, (k, share_gg, share_full)
    for k in ("rowgram", "project"):
        assert wf[k]["bytes"] > 0 and 0.45 <= wg[k]["bytes"] / wf[k]["bytes"] <= 0.8, (k, wg[k], wf[k])
    assert wg["rotgram"]["flops"] < 0.5 * wf["rotgram"]["flops"]       # flat kept spectra: the second Gram pass is passed through
    h.prof_enable(True)                                               # times only: the counters stay untouched
    t = tn.Tensor(gg, batch=True); t.round_tt(rmax=32)
    torch.cuda.synchronize()
    h.prof_collect()
    h.prof_enable(False)
    assert all(v["flops"] == 0 and v["bytes"] == 0 for v in h.prof_collect_work().values())
